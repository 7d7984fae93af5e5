#!/bin/bash
# interleaved A/B of the library builds under csrc/variants on the bench workload (GPU box): REPS rounds
root=$(cd $(dirname $0)/.. && pwd)
for rep in $(seq 1 ${REPS:-3}); do
  for so in $root/bayesiandatafusion.jl_amd/csrc/variants/libbdf_*.so; do
    BDF_LIB_PATH=$so python3 $root/bench.py --steps ${STEPS:-400} --warmup 200 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref --k1-min-launches 0 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$(basename $so)', d['value'], 'sweeps/s', d['ms_per_step'], 'ms  K1', d['roofline']['avg_launch_us'], 'us')"
  done
done

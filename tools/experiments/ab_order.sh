#!/bin/bash
# every K1 variant under csrc/variants x every launch order (BDF_K1_ORDER) on the bench workload (GPU box)
root=$(cd $(dirname $0)/.. && pwd)
for so in $root/bayesiandatafusion.jl_amd/csrc/variants/libbdf_*.so; do
  for o in ${ORDERS:-0 1 2 3}; do
    BDF_K1_ORDER=$o BDF_LIB_PATH=$so python3 $root/bench.py --steps ${1:-300} --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$(basename $so) order $o:', d['value'], 'sweeps/s', d['ms_per_step'], 'ms  K1', d['roofline']['avg_launch_us'], 'us')"
  done
done

#!/bin/bash
# packed K1 launch (BDF_K1_SLOTS waves per SIMD, BDF_K1_FCOST) on the bench workload (GPU box)
root=$(cd $(dirname $0)/.. && pwd)
run() { # lib slots fcost
  BDF_K1_SLOTS=$2 BDF_K1_FCOST=$3 BDF_LIB_PATH=$root/bayesiandatafusion.jl_amd/csrc/variants/libbdf_$1.so python3 $root/bench.py --steps ${STEPS:-300} --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1 slots $2 fcost $3:', d['value'], 'sweeps/s', d['ms_per_step'], 'ms  K1', d['roofline']['avg_launch_us'], 'us rmse', d['test_rmse'])"
}
run p5 0 55
for f in 30 55 80 120; do run p5 5 $f; done
for f in 30 55 80; do run p5 4 $f; done
for f in 30 55 80; do run p6 6 $f; done
run p5 3 55

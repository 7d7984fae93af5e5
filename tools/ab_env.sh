#!/bin/bash
# interleaved A/B of one environment switch on the bench workload (GPU box): tools/ab_env.sh VAR [runs]
var=$1; n=${2:-5}
root=$(cd $(dirname $0)/.. && pwd)
for i in $(seq 1 $n); do for on in 0 1; do
  if [ $on = 1 ]; then export $var=1; else unset $var; fi
  python3 $root/bench.py --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref --k1-min-launches 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$var=$on', d['value'], 'sweeps/s', d['ms_per_step'], 'ms  K1', d['roofline']['avg_launch_us'], 'rmse', d['test_rmse'])"
done; done

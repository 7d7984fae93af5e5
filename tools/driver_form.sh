#!/bin/bash
# the driver's bench command (--steps 20 --warmup 5) beside longer regions / warm-ups and the schedule switches (GPU box)
root=$(cd $(dirname $0)/.. && pwd)
out=$root/gpurun_out/driver_form.txt
: > $out
run() { # label, env..., -- args
  label=$1; shift
  ( for kv in "$@"; do [ "$kv" = "--" ] && break; export "$kv"; done
    args=(); seen=0; for a in "$@"; do if [ $seen = 1 ]; then args+=("$a"); fi; [ "$a" = "--" ] && seen=1; done
    BDF_DEBUG=1 BDF_BENCH_DEBUG=1 python3 $root/bench.py --gpus 1 --no-c4 --no-c3 --no-c5 --no-mref --no-cpu-baseline "${args[@]}" 2> /tmp/err.txt | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$label', d['value'], 'sweeps/s', d['ms_per_step'], 'ms  K1', r['avg_launch_us'], 'alone', r['avg_launch_us_alone'], 'rmse', d['test_rmse'])" >> $out
    grep -E "^\[b" /tmp/err.txt | cut -c1-700 >> $out )
}
for i in 1 2 3; do run "driver-form#$i" -- --steps 20 --warmup 5; done
run "w500/s20" -- --steps 20 --warmup 500
run "w5/s200" -- --steps 200 --warmup 5
run "w500/s200" -- --steps 200 --warmup 500
run "driver-form NO_POLL" BDF_NO_POLL=1 -- --steps 20 --warmup 5
run "driver-form RESERVE0" BDF_RESERVE_CUS=0 -- --steps 20 --warmup 5
run "w500/s200 NO_POLL" BDF_NO_POLL=1 -- --steps 200 --warmup 500
run "w500/s200 RESERVE0" BDF_RESERVE_CUS=0 -- --steps 200 --warmup 500
cat $out

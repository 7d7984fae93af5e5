#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do
BDF_DEBUG=1 BDF_BENCH_DEBUG=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref 2>/tmp/e.txt | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('driver form', d['value'], d['ms_per_step'], 'K1 in region', d['roofline']['avg_launch_us'], 'core', d['config']['host_core'])"
grep -a "timed region" /tmp/e.txt | tail -1 | cut -c1-330
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5 6; do
for cfg in "BDF_BENCH_NO_PRESYNC=1" "BDF_X=1"; do
env $cfg BDF_BENCH_DEBUG=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref 2>/tmp/e.txt | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$cfg driver form', d['value'], end=' ')"; grep -a "timed region" /tmp/e.txt | tail -1 | sed 's/.*enqueue total/enqueue total/'
done; done

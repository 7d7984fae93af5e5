#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for cfg in "BDF_PRED_LAG=2" "BDF_PRED_LAG=3"; do
echo "== $cfg"
for i in 1 2; do env $cfg BDF_DEBUG=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref 2>/tmp/e.txt | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('driver form', d['value'], 'rmse', d['test_rmse'], end=' ')"; grep -a "host enqueue" /tmp/e.txt | tail -1; done
env $cfg BDF_DEBUG=1 python3 bench.py --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref 2>/tmp/e.txt | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('long form', d['value'], 'rmse', d['test_rmse'], end=' ')"; grep -a "host enqueue" /tmp/e.txt | tail -1
done; done

#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for cfg in "BDF_EV_TIMING=1" "BDF_X=1"; do
echo "== $cfg"
env $cfg python3 tools/sweep_pace_parts.py 2>&1 | grep -a "^iteration"
env $cfg python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('driver form', d['value'], d['test_rmse'])"
done; done

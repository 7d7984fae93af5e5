#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "BDF_X=1" "BDF_ITEM_SIZE=224 BDF_PIECE_SIZE=160" "BDF_ITEM_SIZE=256 BDF_PIECE_SIZE=176" "BDF_ITEM_SIZE=192 BDF_PIECE_SIZE=160" "BDF_ITEM_SIZE=160 BDF_PIECE_SIZE=112" "BDF_X=1"; do
echo "== $cfg"
env $cfg python3 tools/sweep_pace_parts.py 2>&1 | grep -a "^iteration"
env $cfg python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('driver form', d['value'], 'K1 in region', d['roofline']['avg_launch_us'], 'alone', d['roofline']['avg_launch_us_alone'], 'rmse', d['test_rmse'])"
done

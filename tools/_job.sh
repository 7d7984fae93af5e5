#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for so in a_head b_coopcode; do
export BDF_LIB_PATH=$GRAFT_REPO_ROOT/bayesiandatafusion.jl_amd/csrc/variants/libbdf_$so.so
echo "== $so"
python3 tools/sweep_pace_parts.py 2>&1 | grep -a "^iteration"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('driver form', d['value'], 'alone', d['roofline']['avg_launch_us_alone'])"
done; done

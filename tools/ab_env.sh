#!/bin/bash
# interleaved A/B of environment settings under the driver's command (GPU box): tools/ab_env.sh "A=1 B=2" "C=3" ...  ("" = default)
R=$GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref"
for rep in 1 2 3 4; do for cfg in "$@"; do
  if [ -z "$cfg" ]; then v=$(python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 $B 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['avg_launch_us'])")
  else v=$(env $cfg python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 $B 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['avg_launch_us'])"); fi
  echo "[${cfg:-default}] run $rep: $v"
done; done

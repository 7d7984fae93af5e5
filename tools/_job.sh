cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | grep -E "passed|failed|^E " | tail -5 > gpurun_out/suite.txt
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-c4 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140 >> gpurun_out/suite.txt; done
python3 bench.py --no-c4 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('long', d['value'], d['ms_per_step'], d['roofline'])" >> gpurun_out/suite.txt
cat gpurun_out/suite.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x 2>&1 | grep -a "passed\|failed\|rror" | tail -3
python3 tools/soak_determinism.py 3000 2>&1 | tail -1
for rep in $(seq 1 16); do
BDF_BENCH_DEBUG=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref 2>/tmp/e.txt | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('driver form', d['value'], d['ms_per_step'], d['test_rmse'], end=' ')"
grep -a "timed region" /tmp/e.txt | tail -1 | sed 's/.*enqueue total/enqueue total/'
done
python3 bench.py --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('long form', d['value'], d['ms_per_step'], d['test_rmse'])"

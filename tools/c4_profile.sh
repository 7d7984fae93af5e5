#!/bin/bash
# kernel statistics of a C4-shaped run (GPU box): ROWS COLS NNZ as arguments (default 2M x 200k x 20M)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c4prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-c3 --no-c5 --no-mref --k1-min-launches 0 --c4-rows ${1:-2000000} --c4-cols ${2:-200000} --c4-nnz ${3:-20000000} > /tmp/c4.log 2>&1
tail -1 /tmp/c4.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['c4'])"
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/c4prof/*/*kernel_stats.csv")[0])))
for r in rows[:12]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:10.1f} us  {r['Percentage']:>6s}%")
PY

#!/bin/bash
# rocprofv3 kernel statistics of the default bench workload (GPU box); prints the top kernels
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bs
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bs -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 300 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref > /tmp/bs.log 2>&1
tail -1 /tmp/bs.log | cut -c1-200
cp /tmp/bs/*/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/${1:-bench}_kernel_stats.csv
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/bs/*/*kernel_stats.csv")[0])))
for r in rows[:10]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']:>6s}%")
PY

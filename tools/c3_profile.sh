#!/bin/bash
# kernel statistics of the C3 probe (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3prof -- python3 $GRAFT_REPO_ROOT/tools/c3_probe.py > /tmp/c3.log 2>&1
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/c3prof/*/*kernel_stats.csv")[0])))
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']:>6s}%")
PY

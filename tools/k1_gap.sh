#!/bin/bash
# gap between back-to-back row kernels on ONE stream with nothing in between (GPU box): what a kernel boundary costs
cd /tmp && export TMPDIR=/tmp
cat > /tmp/k1gap.py <<PY
import os, sys
os.environ["BDF_NO_NATIVE"] = "1"; os.environ["BDF_NO_OVERLAP"] = "1"
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bdf_amd as B
from bdf_amd import datasets
rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
eng = B.GibbsEngine(rd, 32, seed=1, device=0)
for i in range(1, 4):
    eng.sweep(i)
eng.sync()
for i in range(40):
    eng.ctx.set_sweep(10 + i)
    eng.sample_entity(i % 2)
eng.sync()
eng.close()
PY
rm -rf /tmp/gap
rocprofv3 --kernel-trace --output-format csv -d /tmp/gap -- python3 /tmp/k1gap.py > /tmp/gap.log 2>&1
python3 - <<'PY'
import csv, glob
rows = [r for r in csv.DictReader(open(glob.glob('/tmp/gap/*/*kernel_trace.csv')[0])) if 'k_rows<' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-40:]
gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(rows, rows[1:])]
dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
print("durations us:", [round(d, 1) for d in dur[:8]])
print("gaps us:", [round(g, 1) for g in gaps[:12]], "mean", round(sum(gaps) / len(gaps), 2))
PY

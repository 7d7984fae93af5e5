#!/bin/bash
# kernel timeline of the last two D = 10 sweeps of tools/mref_probe.py under rocprofv3 --kernel-trace (GPU box)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/mr
rocprofv3 --kernel-trace --output-format csv -d /tmp/mr -- python3 $GRAFT_REPO_ROOT/tools/mref_probe.py > /tmp/mr.log 2>&1
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/mr/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = next(i for i, r in enumerate(rows) if "k_rows<32" in r["Kernel_Name"])
part = rows[:idx]
k = [i for i, r in enumerate(part) if "k_rows_small" in r["Kernel_Name"]]
s0 = k[-2]; t0 = int(part[s0]["Start_Timestamp"])
for r in part[s0:s0 + 26]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?')}  " + r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60])
PY

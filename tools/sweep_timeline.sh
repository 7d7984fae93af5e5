#!/bin/bash
# kernel timeline of steady-state sweeps of the bench workload (GPU box): rocprofv3 --kernel-trace, then per-kernel begin / end
# relative to the first row kernel of a late sweep
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 100 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref --k1-min-launches 0 > /tmp/tl.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/tl/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:34]
k1 = [i for i, r in enumerate(rows) if 'k_rows<' in r['Kernel_Name']]
# a sweep = two row kernels; take sweeps 100..103 of the run
i0 = k1[2 * 110]
t0 = int(rows[i0]['Start_Timestamp'])
i1 = k1[2 * 113]
for r in rows[i0 - 2:i1 + 1]:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:7.1f} us  q{r.get('Queue_Id', '?'):>3s}  {names(r)}")
PY

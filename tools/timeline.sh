#!/bin/bash
# Kernel timeline of the bench workload: per-kernel durations and the idle gaps between consecutive kernels (all streams
# merged), from a rocprofv3 --kernel-trace run.  Run on the GPU box through gpurun.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/tl
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 10 --no-cpu-baseline "$@" > /tmp/tl.log 2>&1)
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('/tmp/tl/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', '?'), r.get('Queue_Id', '?')))
rows.sort()
# last 5 steps: find the k_prior launches (2 per step)
idx = [i for i, r in enumerate(rows) if 'k_rows' in r[2]]
start = idx[-11]            # 5 full steps before the last prior pair
seg = rows[start:idx[-1]]
t0 = seg[0][0]
print("one step (two K1 launches), times in us relative to the first kernel of the step:")
step = rows[idx[-5] - 1:idx[-3] + 1]
for s, e, n, st, q in step:
    short = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:40]
    print(f"  {(s - step[0][0]) / 1e3:8.1f} -> {(e - step[0][0]) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f})  q{q} {short}")
span = (seg[-1][1] - seg[0][0]) / 1e3 / 5
# union of busy intervals
busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
for s, e, *_ in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"per step: span {span:.1f} us, GPU busy (any kernel) {busy / 1e3 / 5:.1f} us, idle {span - busy / 1e3 / 5:.1f} us")
PY

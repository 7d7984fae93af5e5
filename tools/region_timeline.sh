#!/bin/bash
# kernel timeline of the driver-form timed region (bench.py --steps 20 --warmup 5) under rocprofv3 --kernel-trace (GPU box):
# the region is the group of 40 row launches between two idle gaps of the row stream
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref --k1-min-launches 0 > /tmp/tl.log 2>&1
tail -2 /tmp/tl.log | cut -c1-300
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/tl/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:26]
k1 = [r for r in rows if 'k_rows<' in r['Kernel_Name']]
# groups of row launches separated by > 60 us of row-stream idleness
groups, cur = [], [k1[0]]
for a, b in zip(k1, k1[1:]):
    if int(b['Start_Timestamp']) - int(a['End_Timestamp']) > 60000:
        groups.append(cur); cur = []
    cur.append(b)
groups.append(cur)
print("row-launch groups:", [len(g) for g in groups])
g = [x for x in groups if len(x) == 40]
if g:
    g = g[-1]
    t0, t1 = int(g[0]['Start_Timestamp']), int(g[-1]['End_Timestamp'])
    last_end = max(int(r['End_Timestamp']) for r in rows if t0 <= int(r['Start_Timestamp']) <= t1 + 200000)
    print(f"region: first row kernel start -> last row kernel end {(t1 - t0) / 1e3:.1f} us; -> last kernel end {(last_end - t0) / 1e3:.1f} us")
    for r in rows:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if t0 - 30000 <= s <= t1 + 200000:
            print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} us  q{r.get('Queue_Id', '?'):>3s}  {name(r)}")
PY

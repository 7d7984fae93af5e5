"""A few steady-state iterations of a rocprofv3 --kernel-trace CSV as a timeline: start / end of every kernel relative to the
first row launch shown (us).  usage: timeline_dump.py <kernel_trace.csv> [iterations=3] [skip_from_end=40]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 3
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k1 = [i for i, r in enumerate(rows) if "k_rows" in r["Kernel_Name"]]
first = k1[-(2 * (n_it + skip))]
last = k1[-(2 * skip)]
t0 = int(rows[first]["Start_Timestamp"])
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:34]
for r in rows[first:last]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{short(r['Kernel_Name']):34s} start {s:9.2f}  end {e:9.2f}  dur {e - s:7.2f}  queue {r.get('Queue_Id', '?')}")

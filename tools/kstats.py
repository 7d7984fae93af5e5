"""prints a rocprofv3 kernel_stats.csv per sweep: python tools/kstats.py file.csv [sweeps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    print(f"{name:46s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us  per sweep {float(r['TotalDurationNs']) / n / 1e3:8.1f} us  {100 * float(r['TotalDurationNs']) / tot:5.1f} %")

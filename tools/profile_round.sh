#!/bin/bash
# Round profile of the bench workload (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats            -> kernel_stats.csv
#   2. separate --pmc passes FETCH_SIZE, WRITE_SIZE -> K1 HBM traffic per launch (gfx950: FETCH_SIZE counts half of a
#      wide coalesced read, MI355X_MICROARCH.md "HBM"; the uncorrected and the doubled figure are both recorded)
# Output under gpurun_out/profile_<tag>/ ; copy what should be judged into profiles/.
tag=${1:-r01}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/profile_$tag
mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 300 --no-cpu-baseline > $out/bench_under_rocprof.log 2>&1)
cp /tmp/prof_stats/*/*kernel_stats.csv $out/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $out/pmc_$c.log 2>&1)
done
python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob('/tmp/prof_%s/*/*counter_collection.csv' % c):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for k, d in agg.items():
    rows.append({"kernel": k[:90], "launches": len(d.get("FETCH_SIZE", [])),
                 "FETCH_SIZE_KB_per_launch": sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [])), 1),
                 "WRITE_SIZE_KB_per_launch": sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [])), 1)})
rows.sort(key=lambda r: -r["FETCH_SIZE_KB_per_launch"] - r["WRITE_SIZE_KB_per_launch"])
k1 = [r for r in rows if "k_rows" in r["kernel"]]
summary = {"rows": rows[:12]}
if k1:
    f, w = k1[0]["FETCH_SIZE_KB_per_launch"] * 1024, k1[0]["WRITE_SIZE_KB_per_launch"] * 1024
    summary["k1_traffic_bytes_per_launch"] = {"fetch_uncorrected": f, "write": w, "hbm_bytes_fetch_doubled": 2 * f + w,
                                               "hbm_bytes_fetch_as_counted": f + w}
json.dump(summary, open('$out/hbm_traffic.json', 'w'), indent=1)
print(json.dumps(summary.get("k1_traffic_bytes_per_launch")))
PY
head -12 $out/kernel_stats.csv | cut -c1-160

#!/bin/bash
# Round-4 profile of the bench workload (GPU box), the parts of tools/profile_round.sh that changed this round:
#   1. the driver's command x3 and the long form        -> bench_driver_cmd_*.json, bench_long.json
#   2. rocprofv3 --kernel-trace --stats                  -> kernel_stats.csv
#   3. separate --pmc passes: FETCH_SIZE, WRITE_SIZE (K1 HBM traffic per launch, with the sha1 of the kernel source), two SQ sets
#   4. the schedule switches under the driver's command; K1 alone
# Output under gpurun_out/profile_<tag>/ ; copy what should be judged into profiles/.
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/profile_$tag
mkdir -p $out
B="--no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref"
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd_$i.json 2> $out/bench_driver_cmd_$i.err; done
python3 bench.py $B > $out/bench_long.json 2> $out/bench_long.err
(cd /tmp && rm -rf /tmp/prof_stats && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 300 $B > $out/bench_under_rocprof.log 2>&1)
cp /tmp/prof_stats/*/*kernel_stats.csv $out/kernel_stats.csv
i=0
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_warm -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --k1-min-launches 0 $B > $out/pmc_pass0.log 2>&1)
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/prof_pmc$i && BDF_NO_POLL=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/prof_pmc$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --k1-min-launches 0 $B > $out/pmc_pass$i.log 2>&1)
done
python3 - <<PY
import csv, glob, json, collections, sys
sys.path.insert(0, '$GRAFT_REPO_ROOT')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/prof_pmc*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for k, d in agg.items():
    rows.append({"kernel": k[:90], "launches": len(d.get("FETCH_SIZE", [])), "launches_write_pass": len(d.get("WRITE_SIZE", [])),
                 "FETCH_SIZE_KB_min_max": [min(d.get("FETCH_SIZE", [0])), max(d.get("FETCH_SIZE", [0]))],
                 "FETCH_SIZE_KB_per_launch": sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [])), 1),
                 "WRITE_SIZE_KB_per_launch": sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [])), 1)})
rows.sort(key=lambda r: -r["FETCH_SIZE_KB_per_launch"] - r["WRITE_SIZE_KB_per_launch"])
k1 = [r for r in rows if "k_rows" in r["kernel"]]
summary = {"rows": rows[:12], "k1_source_sha1": __import__("bench").k1_source_sha1()}
if k1:
    f, w = k1[0]["FETCH_SIZE_KB_per_launch"] * 1024, k1[0]["WRITE_SIZE_KB_per_launch"] * 1024
    summary["k1_traffic_bytes_per_launch"] = {"fetch_uncorrected": f, "write": w, "hbm_bytes_fetch_doubled": 2 * f + w,
                                               "hbm_bytes_fetch_as_counted": f + w}
json.dump(summary, open('$out/hbm_traffic.json', 'w'), indent=1)
pmc = {}
for k, d in agg.items():
    if 'k_rows' in k or 'k_hyper_sample' in k or 'k_predict_runs' in k:
        pmc[k[:90]] = {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in sorted(d.items())}
json.dump(pmc, open('$out/pmc_k_rows.json', 'w'), indent=1)
print(json.dumps(summary.get("k1_traffic_bytes_per_launch")))
PY
for i in 1 2 3; do for cfg in "default" "BDF_NO_POLL=1" "BDF_RESERVE_CUS=0"; do
  if [ "$cfg" = default ]; then v=$(python3 bench.py --gpus 1 --steps 20 --warmup 5 $B 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['value'])")
  else v=$(env $cfg python3 bench.py --gpus 1 --steps 20 --warmup 5 $B 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['value'])"); fi
  echo "$cfg run $i: $v sweeps/s (driver form)"
done; done > $out/schedule_ab_driver_form.txt
python3 tools/k1_alone.py > $out/k1_alone.txt 2>&1
head -8 $out/kernel_stats.csv | cut -c1-160
cat $out/schedule_ab_driver_form.txt
tail -4 $out/k1_alone.txt

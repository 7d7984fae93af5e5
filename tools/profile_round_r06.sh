#!/bin/bash
# Round-6 profile (GPU box): everything the bench line's roofline blocks cite.
#   1. the driver's command x3 and the long form            -> bench_driver_cmd_*.json, bench_long.json
#   2. rocprofv3 --kernel-trace --stats of the headline     -> kernel_stats.csv (+ the driver's own command: kernel_stats_driver_cmd.csv)
#   3. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | two SQ sets), BDF_RESERVE_CUS=0 BDF_NO_POLL=1 (CU-masked streams crash
#      rocprofv3's teardown) -> hbm_traffic.json, pmc_k_rows.json (both carry the sha1 of the row kernels' sources)
#   4. kernel statistics of C3 (ff, cg), C4 (bench's own block), C5, M-ref -> *_kernel_stats.csv, config_kernels.json
#   5. K1 alone
# Output under gpurun_out/profile_<tag>/ ; copy what should be judged into profiles/.
tag=${1:-r06}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/profile_$tag
mkdir -p $out
B="--no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref"
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 $B > $out/bench_driver_cmd_$i.json 2> $out/bench_driver_cmd_$i.err; done
python3 bench.py $B > $out/bench_long.json 2> $out/bench_long.err
(cd /tmp && rm -rf /tmp/prof_stats && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --steps 200 --warmup 300 $B > $out/bench_under_rocprof.log 2>&1)
cp /tmp/prof_stats/*/*kernel_stats.csv $out/kernel_stats.csv
(cd /tmp && rm -rf /tmp/prof_drv && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_drv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 $B > $out/bench_driver_under_rocprof.log 2>&1)
cp /tmp/prof_drv/*/*kernel_stats.csv $out/kernel_stats_driver_cmd.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/prof_pmc$i && BDF_RESERVE_CUS=0 BDF_NO_POLL=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/prof_pmc$i -- python3 $R/bench.py --steps 8 --warmup 4 --k1-min-launches 0 $B > $out/pmc_pass$i.log 2>&1)
done
python3 - <<PY
import csv, glob, json, collections, sys
sys.path.insert(0, '$R')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/prof_pmc*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
sha = __import__("bench").k1_source_sha1()
rows = []
for k, d in agg.items():
    rows.append({"kernel": k[:90], "launches": len(d.get("FETCH_SIZE", [])), "launches_write_pass": len(d.get("WRITE_SIZE", [])),
                 "FETCH_SIZE_KB_per_launch": sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [])), 1),
                 "WRITE_SIZE_KB_per_launch": sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [])), 1)})
rows.sort(key=lambda r: -r["FETCH_SIZE_KB_per_launch"] - r["WRITE_SIZE_KB_per_launch"])
k1 = [r for r in rows if "k_rows_col" in r["kernel"]] or [r for r in rows if "k_rows" in r["kernel"]]
summary = {"rows": rows[:12], "k1_source_sha1": sha, "schedule": "BDF_RESERVE_CUS=0 BDF_NO_POLL=1: all 256 CUs, event hand-overs (rocprofv3 --pmc serialises the streams)",
           "note": "rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes (KB per launch); hbm_bytes_fetch_doubled applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 64 of every 128 bytes of a wide streaming read)"}
if k1:
    f, w = k1[0]["FETCH_SIZE_KB_per_launch"] * 1024, k1[0]["WRITE_SIZE_KB_per_launch"] * 1024
    summary["k1_traffic_bytes_per_launch"] = {"kernel": k1[0]["kernel"], "fetch_uncorrected": f, "write": w, "hbm_bytes_fetch_doubled": 2 * f + w,
                                               "hbm_bytes_fetch_as_counted": f + w}
json.dump(summary, open('$out/hbm_traffic.json', 'w'), indent=1)
pmc = {"k1_source_sha1": sha, "schedule": "BDF_RESERVE_CUS=0 BDF_NO_POLL=1: all 256 CUs (1,024 SIMDs), event hand-overs", "kernels": {}}
for k, d in agg.items():
    if 'k_rows' in k or 'k_hyper' in k or 'k_predict_runs' in k:
        pmc["kernels"][k[:90]] = {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in sorted(d.items())}
json.dump(pmc, open('$out/pmc_k_rows.json', 'w'), indent=1)
print(json.dumps(summary.get("k1_traffic_bytes_per_launch")))
PY
# ---- the other configurations
cfg=$out/config_kernels_parts
mkdir -p $cfg
for w in ff cg; do
  (cd /tmp && rm -rf /tmp/c3prof_$w && BDF_RESERVE_CUS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3prof_$w -- python3 $R/tools/c3_probe.py $w > $out/c3_prof_$w.txt 2>&1)
  cp $(find /tmp/c3prof_$w -name "*kernel_stats.csv" | head -1) $out/c3_${w}_kernel_stats.csv
done
(cd /tmp && rm -rf /tmp/c5prof && BDF_RESERVE_CUS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5prof -- python3 $R/tools/c5_probe.py > $out/c5_prof.txt 2>&1)
cp $(find /tmp/c5prof -name "*kernel_stats.csv" | head -1) $out/c5_kernel_stats.csv
for d in 10 30; do
  (cd /tmp && rm -rf /tmp/mr$d && MREF_D=$d BDF_RESERVE_CUS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mr$d -- python3 $R/tools/mref_probe.py > $out/mref_d${d}_prof.txt 2>&1)
  cp $(find /tmp/mr$d -name "*kernel_stats.csv" | head -1) $out/mref_d${d}_kernel_stats.csv
done
(cd /tmp && rm -rf /tmp/c4prof && BDF_RESERVE_CUS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4prof -- python3 $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-c3 --no-c5 --no-mref --no-c4-uniform --k1-min-launches 0 > $out/c4_prof.txt 2>&1)
cp $(find /tmp/c4prof -name "*kernel_stats.csv" | head -1) $out/c4_kernel_stats.csv
python3 - <<PY
import csv, json
def top(path, skip=()):
    rows = [r for r in csv.DictReader(open(path)) if not any(s in r["Name"] for s in skip)]
    r = rows[0]
    tot = sum(float(x["TotalDurationNs"]) for x in rows)
    return {"kernel": r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:100], "calls": int(r["Calls"]),
            "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "share_of_kernel_time": round(float(r["TotalDurationNs"]) / tot, 3)}
out = {}
for name, f in (("c3_ff", "c3_ff_kernel_stats.csv"), ("c3_cg", "c3_cg_kernel_stats.csv"), ("c5", "c5_kernel_stats.csv"), ("c4", "c4_kernel_stats.csv")):
    try: out[name] = top("$out/" + f, skip=("k_spin", "elementwise", "fillBuffer", "copyBuffer"))
    except Exception as e: out[name] = {"error": str(e)}
for d in (10, 30):
    try: out[f"mref_d{d}"] = top(f"$out/mref_d{d}_kernel_stats.csv", skip=("k_spin", "elementwise", "fillBuffer", "copyBuffer"))
    except Exception as e: out[f"mref_d{d}"] = {"error": str(e)}
json.dump(out, open("$out/config_kernels.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
python3 tools/k1_alone.py > $out/k1_alone.txt 2>&1
BDF_RESERVE_CUS=8 python3 tools/k1_alone.py 2>&1 | grep "K1 alone\|launches only" | sed 's/^/[248 CUs] /' >> $out/k1_alone.txt
# ---- a timeline of steady-state iterations (kernel trace of every launch; the profiler's teardown may crash after the output is written)
(cd /tmp && rm -rf /tmp/tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $R/bench.py --steps 100 --warmup 100 --k1-min-launches 0 $B > $out/timeline_bench.log 2>&1)
python3 tools/timeline_dump.py $(find /tmp/tl -name "*kernel_trace.csv" | head -1) 4 20 > $out/timeline.txt 2>&1
python3 tools/kstats.py $out/kernel_stats.csv 500 8
tail -4 $out/k1_alone.txt
# ---- the one-launch CG solve's iteration by stamps (diagnostic build: tools/ab_k1.sh build cgst "-DBDF_CG_STAMPS"), the peer exchange's two-process timeline
if [ -f $R/bayesiandatafusion.jl_amd/csrc/variants/libbdf_cgst.so ]; then
  BDF_LIB_PATH=$R/bayesiandatafusion.jl_amd/csrc/variants/libbdf_cgst.so python3 tools/c3_cg_stamps.py > $out/c3_cg_handover.txt 2>&1
fi
timeout 900 bash tools/peer_overlap_trace.sh profile_$tag > $out/peer_overlap.log 2>&1
# ---- the driver's own command, everything in it (the line the round's BENCH record will hold)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_full.json 2> $out/bench_driver_full.err
tail -c 600 $out/bench_driver_full.json

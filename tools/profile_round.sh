#!/bin/bash
# Round profile of the bench workload (run on the GPU box through gpurun):
#   1. the bench line itself                       -> bench.json
#   2. rocprofv3 --kernel-trace --stats            -> kernel_stats.csv
#   3. separate --pmc passes (never with sys / runtime traces): FETCH_SIZE, WRITE_SIZE -> K1 HBM traffic per launch (gfx950:
#      FETCH_SIZE counts half of a wide coalesced read, MI355X_MICROARCH.md "HBM"; the uncorrected and the doubled figure
#      are both recorded, with the sha1 of the row kernel's source the passes ran on); two SQ counter sets -> pmc_k_rows.json
#   4. micro-benchmarks behind DESIGN.md's bound analysis: fp64 pipe probe, gather and finish-phase probes, the row kernel and
#      the prediction update alone, CU mask probe, back-to-back row kernels
# Output under gpurun_out/profile_<tag>/ ; copy what should be judged into profiles/.
tag=${1:-r03}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/profile_$tag
mkdir -p $out
B="--no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref"
# the driver's command first (fresh process, 5 warm-up + 20 timed steps), three times; then the default long form
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd_$i.json 2> $out/bench_driver_cmd_$i.err; done
python3 bench.py --no-c4 --no-c3 --no-c5 --no-mref --no-cpu-baseline > $out/bench.json 2> $out/bench.err
(cd /tmp && rm -rf /tmp/prof_stats && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 300 $B > $out/bench_under_rocprof.log 2>&1)
cp /tmp/prof_stats/*/*kernel_stats.csv $out/kernel_stats.csv
i=0
# (the first --pmc invocation on a fresh box has returned a third of the dispatches with doubled values: one throw-away pass first)
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_warm -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --k1-min-launches 0 $B > $out/pmc_pass0.log 2>&1)
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
  i=$((i+1))
  # BDF_NO_POLL=1: under --pmc kernels of different queues are serialised; a row kernel that polls for the draw would spin
  (cd /tmp && rm -rf /tmp/prof_pmc$i && BDF_NO_POLL=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/prof_pmc$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --k1-min-launches 0 $B > $out/pmc_pass$i.log 2>&1)
done
python3 - <<PY
import csv, glob, json, collections, hashlib, sys
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
summary = {"rows": rows[:12],
           "k1_source_sha1": __import__("bench").k1_source_sha1()}      # the kernel source and every header it includes
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
# the D = 64 row kernel on a C4-shaped relation (2M x 200k, 20M observations): kernel statistics and SQ counters
bash tools/c4_profile.sh 2000000 200000 20000000 > $out/c4_shaped_kernel_stats.txt 2>&1
C4="--steps 2 --warmup 2 --no-cpu-baseline --no-c3 --no-c5 --no-mref --k1-min-launches 0 --c4-rows 2000000 --c4-cols 200000 --c4-nnz 20000000 --c4-sweeps 3"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/prof_c4pmc$i && BDF_NO_POLL=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/prof_c4pmc$i -- python3 $GRAFT_REPO_ROOT/bench.py $C4 > $out/c4_pmc_pass$i.log 2>&1)
done
python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/prof_c4pmc*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_rows<64' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: {"n": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for c, v in sorted(d.items())} for k, d in agg.items()}
json.dump(out, open('$out/pmc_k_rows64_c4_shaped.json', 'w'), indent=1)
PY
# schedule switches under the driver's command, interleaved (polling on reserved CUs | events on reserved CUs | events on all CUs)
for i in 1 2 3 4 5; do for cfg in "default" "BDF_NO_POLL=1" "BDF_RESERVE_CUS=0"; do
  if [ "$cfg" = default ]; then v=$(python3 bench.py --gpus 1 --steps 20 --warmup 5 $B 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['value'])")
  else v=$(env $cfg python3 bench.py --gpus 1 --steps 20 --warmup 5 $B 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['value'])"); fi
  echo "$cfg run $i: $v sweeps/s (driver form)"
done; done > $out/schedule_ab_driver_form.txt
# (tools/bin/* are built in the container before the call: tools/build_probes.sh)
timeout 120 tools/bin/valu_cost_probe > $out/valu_cost_probe.txt 2>&1
for p in factor_probe_b factor_probe64_b; do echo "== $p (blocked variant)"; timeout 200 tools/bin/$p; done > $out/factor_probe_blocked.txt 2>&1
python3 tools/region_pace.py 5 12 20 > $out/region_pace.txt 2>&1
python3 tools/c3_probe.py > $out/c3.txt 2>&1
python3 tools/setup_cost.py > $out/setup_cost.txt 2>&1
python3 tools/c5_probe.py > $out/c5.txt 2>&1
tools/bin/fp64_pipe_probe > $out/fp64_pipe_probe.txt 2>&1
timeout 200 tools/bin/gather_probe 3952 > $out/gather_probe.txt 2>&1
timeout 200 tools/bin/factor_probe > $out/factor_probe.txt 2>&1
timeout 200 tools/bin/factor_probe64 >> $out/factor_probe.txt 2>&1
python3 tools/k1_alone.py > $out/k1_alone.txt 2>&1
tools/bin/hip_call_cost > $out/hip_call_cost.txt 2>&1
timeout 120 tools/bin/cu_mask_probe 8 > $out/cu_mask_probe.txt 2>&1
tools/k1_gap.sh > $out/k1_back_to_back.txt 2>&1
tools/sweep_timeline.sh > $out/sweep_timeline.txt 2>&1
head -12 $out/kernel_stats.csv | cut -c1-160
tail -2 $out/bench.json | cut -c1-300

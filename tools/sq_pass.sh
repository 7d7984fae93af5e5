#!/bin/bash
# SQ counters of the row kernel, the hyperprior draw and the prediction update (GPU box), event hand-overs (BDF_NO_POLL=1: under
# --pmc kernels of different queues are serialised and a polling row kernel would spin) -> gpurun_out/sq/pmc_k_rows.json
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sq
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
  i=$((i+1)); rm -rf /tmp/sq$i
  BDF_NO_POLL=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sq$i -- python3 $R/bench.py --steps 8 --warmup 4 --k1-min-launches 0 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref > $R/gpurun_out/sq/pass$i.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/sq*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
pmc = {}
for k, d in agg.items():
    if 'k_rows' in k or 'k_hyper_sample' in k or 'k_predict_runs' in k:
        pmc[k[:90]] = {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in sorted(d.items())}
json.dump(pmc, open('$R/gpurun_out/sq/pmc_k_rows.json', 'w'), indent=1)
for k, v in pmc.items():
    print(k[:60]); print({c: (x['n'], round(x['mean'])) for c, x in v.items()})
PY

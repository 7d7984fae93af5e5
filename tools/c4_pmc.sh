#!/bin/bash
# SQ and memory counters of configuration C4's row kernels at full size (GPU box) -> gpurun_out/r04_pmc_c4.json
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export BDF_RESERVE_CUS=0
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1)); rm -rf /tmp/c4p$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/c4p$i -- python3 $R/bench.py --steps 2 --warmup 2 --k1-min-launches 0 --no-cpu-baseline --no-c3 --no-c5 --no-mref --c4-sweeps 2 > /tmp/c4p$i.log 2>&1
  tail -2 /tmp/c4p$i.log | cut -c1-300
done
python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/c4p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
pmc = {}
for k, d in agg.items():
    if ('k_rows' in k and '64' in k) or 'k_rowmat' in k or 'k_hyper_partial<64' in k:
        pmc[k[:100]] = {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in sorted(d.items())}
json.dump(pmc, open('$R/gpurun_out/r04_pmc_c4.json', 'w'), indent=1)
for k, v in pmc.items():
    print(k[:70]); print({c: (x['n'], round(x['mean'])) for c, x in v.items()})
PY

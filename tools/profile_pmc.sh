#!/bin/bash
# PMC passes for the bench workload (run on the GPU box via gpurun).  Counters are collected in separate passes,
# with --kernel-trace only (never combined with sys/runtime traces).  Output: gpurun_out/pmc_<tag>/pass*.
tag=${1:-r1}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $out/pass$i.log 2>&1)
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$out/pass*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    if not any(x in k for x in ('k_rows', 'k_hyper', 'k_predict')): continue
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s n=%-4d mean=%.4g' % (c, len(v), sum(v) / len(v)))
PY

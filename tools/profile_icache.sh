#!/bin/bash
# Instruction-cache counters of the bench workload's kernels (separate --pmc passes, --kernel-trace only).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/icache
mkdir -p $out
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQ_IFETCH SQ_IFETCH_LEVEL" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU"; do
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
    if not any(x in k for x in ('k_rows', 'k_predict<')): continue
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s n=%-4d mean=%.4g' % (c, len(v), sum(v) / len(v)))
PY

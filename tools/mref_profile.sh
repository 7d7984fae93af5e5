#!/bin/bash
# the reference's benchmark shape (tools/mref_probe.py) under rocprofv3 (GPU box): kernel statistics, then SQ counters of the row kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/mr && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mr -- python3 $R/tools/mref_probe.py > /tmp/mr.log 2>&1
grep -a "M-ref" /tmp/mr.log
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/mr/*/*kernel_stats.csv")[0])))
for r in rows[:10]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:10.1f} us  {r['Percentage']:>6s}%")
PY
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/mrp$i && BDF_NO_POLL=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/mrp$i -- python3 $R/tools/mref_probe.py > /tmp/mrp$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/mrp*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_rows' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:60], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}, 'launches', len(next(iter(d.values()))))
PY

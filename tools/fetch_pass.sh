#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the row kernel per launch under a few schedules (GPU box): how much of the fetch is the schedule's
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "default" "BDF_NO_POLL=1" "BDF_RESERVE_CUS=0" "BDF_NO_NATIVE=1 BDF_NO_OVERLAP=1"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/fp
    if [ "$cfg" = default ]; then rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/fp -- python3 $R/bench.py --steps 8 --warmup 4 --k1-min-launches 0 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref > /tmp/fp.log 2>&1
    else env $cfg $(which rocprofv3) --pmc $c --kernel-trace --output-format csv -d /tmp/fp -- python3 $R/bench.py --steps 8 --warmup 4 --k1-min-launches 0 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref > /tmp/fp.log 2>&1; fi
    python3 - "$cfg" $c <<'PY'
import csv, glob, sys, collections
v = collections.defaultdict(list)
for f in glob.glob('/tmp/fp/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == sys.argv[2]: v[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
for k, x in v.items():
    if 'k_rows' in k or 'k_predict_runs' in k: print(sys.argv[1], sys.argv[2], k, 'launches', len(x), 'KB per launch: mean %.0f min %.0f max %.0f' % (sum(x) / len(x), min(x), max(x)))
PY
  done
done

#!/bin/bash
# kernel statistics of configuration C4 at full size and of the reference's benchmark shape with the low-rank sampler (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export BDF_RESERVE_CUS=0
for what in c4 mref; do
  rm -rf /tmp/lrp_$what
  if [ $what = c4 ]; then extra="--no-mref"; else extra="--no-c4"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lrp_$what -- python3 $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-c3 --no-c5 $extra --k1-min-launches 0 > /tmp/lrp_$what.log 2>&1
  echo "== $what"
  tail -1 /tmp/lrp_$what.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print({k: v for k, v in d.get('$what', {}).items() if k in ('ms_per_sweep', 'd10', 'd30', 'test_rmse')})"
  python3 - $what <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(f"/tmp/lrp_{sys.argv[1]}/*/*kernel_stats.csv")[0])))
for r in rows[:14]:
    print(f"{r['Name'].replace('(anonymous namespace)::','')[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:10.1f} us  total {float(r['TotalDurationNs'])/1e6:9.2f} ms {r['Percentage']:>6s}%")
PY
  cp /tmp/lrp_$what/*/*kernel_stats.csv $R/gpurun_out/r04_lr_${what}_kernel_stats.csv
done

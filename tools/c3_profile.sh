#!/bin/bash
# kernel statistics of configuration C3 (GPU box): the FF path and the forced CG path, one rocprofv3 run each -> gpurun_out/c3/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/c3
python3 $R/tools/c3_probe.py > $R/gpurun_out/c3/plain.txt 2>&1
for w in ff cg; do
  rm -rf /tmp/c3prof_$w
  BDF_RESERVE_CUS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3prof_$w -- python3 $R/tools/c3_probe.py $w > $R/gpurun_out/c3/prof_$w.txt 2>&1
  cp $(find /tmp/c3prof_$w -name "*kernel_stats.csv" | head -1) $R/gpurun_out/c3/${w}_kernel_stats.csv
done
cat $R/gpurun_out/c3/plain.txt
for w in ff cg; do echo "== $w"; cut -d, -f1-4 $R/gpurun_out/c3/${w}_kernel_stats.csv | cut -c1-150 | head -22; done

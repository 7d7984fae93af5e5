#!/bin/bash
# kernel statistics of configuration C5 on one GPU (GPU box) -> gpurun_out/c5/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/c5
rm -rf /tmp/c5prof
BDF_RESERVE_CUS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5prof -- python3 $R/tools/c5_probe.py > $R/gpurun_out/c5/prof.txt 2>&1
cp $(find /tmp/c5prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/c5/kernel_stats.csv
tail -3 $R/gpurun_out/c5/prof.txt
python3 $R/tools/kstats.py $R/gpurun_out/c5/kernel_stats.csv 1 24

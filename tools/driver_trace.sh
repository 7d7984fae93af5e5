#!/bin/bash
# rocprofv3 --kernel-trace --stats (no counters) of the driver's command with the schedule the bench measures: the row kernels
# polling for the hyperprior draws on reserved CUs (GPU box).  rocprofv3 crashes in its own teardown on processes that used
# CU-masked streams, after its output is written.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/dt
BDF_DEBUG=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-c3 --no-c4 --no-c5 --no-mref > /tmp/dt.log 2>&1
grep -a "polling=" /tmp/dt.log | head -2
grep -a '"metric"' /tmp/dt.log | cut -c1-260
f=$(ls /tmp/dt/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/r04_kernel_stats_polling_driver_cmd.csv
python3 $R/tools/kstats.py $f 1 14

#!/bin/bash
# copy the judged part of gpurun_out/profile_<tag>/ (tools/profile_round_r05.sh) into profiles/r05_*
tag=${1:-r05}
S=$(cd $(dirname $0)/.. && pwd)/gpurun_out/profile_$tag
P=$(cd $(dirname $0)/.. && pwd)/profiles
for f in kernel_stats.csv kernel_stats_driver_cmd.csv hbm_traffic.json pmc_k_rows.json config_kernels.json bench_long.json c3_ff_kernel_stats.csv \
         c3_cg_kernel_stats.csv c4_kernel_stats.csv c5_kernel_stats.csv mref_kernel_stats.csv k1_alone.txt; do
  [ -f $S/$f ] && cp $S/$f $P/r05_$f
done
cat $S/bench_driver_cmd_1.json $S/bench_driver_cmd_2.json $S/bench_driver_cmd_3.json 2>/dev/null | grep '^{' > $P/r05_bench_driver_cmd_repeats.jsonl
[ -f $S/timeline.txt ] && cp $S/timeline.txt $P/r05_timeline_final.txt
ls -la $P | grep r05_ | wc -l

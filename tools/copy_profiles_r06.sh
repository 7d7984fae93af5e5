#!/bin/bash
# copy the judged part of gpurun_out/profile_<tag>/ (tools/profile_round_r06.sh) into profiles/r06_*
tag=${1:-r06}
S=$(cd $(dirname $0)/.. && pwd)/gpurun_out/profile_$tag
P=$(cd $(dirname $0)/.. && pwd)/profiles
for f in kernel_stats.csv kernel_stats_driver_cmd.csv hbm_traffic.json pmc_k_rows.json config_kernels.json bench_long.json c3_ff_kernel_stats.csv \
         c3_cg_kernel_stats.csv c4_kernel_stats.csv c5_kernel_stats.csv mref_d10_kernel_stats.csv mref_d30_kernel_stats.csv k1_alone.txt c3_cg_handover.txt peer_overlap_timeline.txt; do
  [ -f $S/$f ] && cp $S/$f $P/r06_$f
done
[ -f $S/bench_driver_full.json ] && grep '^{' $S/bench_driver_full.json | tail -1 > $P/r06_bench_driver_form.json
cat $S/bench_driver_cmd_1.json $S/bench_driver_cmd_2.json $S/bench_driver_cmd_3.json 2>/dev/null | grep '^{' > $P/r06_bench_driver_cmd_repeats.jsonl
[ -f $S/timeline.txt ] && cp $S/timeline.txt $P/r06_timeline_final.txt
ls -la $P | grep r06_ | wc -l

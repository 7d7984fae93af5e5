#!/bin/bash
# K1c round 6: the driver-form bench over library builds (r05 baseline, three-wave and two-wave builds of the blocked factorisation), and per-wave stamps
cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r06B; mkdir -p $o
V=$GRAFT_REPO_ROOT/bayesiandatafusion.jl_amd/csrc/variants
B="--no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref"
bench() { echo "== $*" >> $o/bench.txt; env "$@" python3 bench.py --gpus 1 --steps 20 --warmup 5 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r.get('frac'), r.get('avg_launch_us'), r.get('avg_launch_us_alone'), d.get('value_without_device_warmup'))" >> $o/bench.txt; }
for rep in 1 2 3; do
  bench BDF_LIB_PATH=$V/libbdf_r05.so
  bench X=1
  bench BDF_LIB_PATH=$V/libbdf_w2.so
done
cat $o/bench.txt
for cfg in "st3 128" "st3 96" "st2 128"; do set -- $cfg
  echo "== $1 T=$2" >> $o/stamps.txt
  BDF_K1_COL=$2 BDF_LIB_PATH=$V/libbdf_$1.so python3 tools/col_stamps.py >> $o/stamps.txt 2>&1
done

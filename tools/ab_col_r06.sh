#!/bin/bash
# K1c round 6: the row launches alone (tools/k1_alone.py) over piece sizes, tier orders, the ticket queue and the two-wave build (GPU box)
cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r06A; mkdir -p $o
run() { echo "== $*" >> $o/alone.txt; env "$@" python3 tools/k1_alone.py 2>&1 | grep -E "launches only|K1 alone" >> $o/alone.txt; }
timeout 900 python3 -m pytest tests/test_gpu_rows_col.py -x -q > $o/pytest_col.txt 2>&1
tail -3 $o/pytest_col.txt
for R in 0 8; do
  for T in 128 96 80 64 48; do run BDF_RESERVE_CUS=$R BDF_K1_COL=$T; done
  for TI in HLh HhL HHL HlH; do run BDF_RESERVE_CUS=$R BDF_K1_COL=80 BDF_COL_TIERS=$TI; run BDF_RESERVE_CUS=$R BDF_K1_COL=64 BDF_COL_TIERS=$TI; done
  run BDF_RESERVE_CUS=$R BDF_K1_COL=64 BDF_COL_QUEUE=0
  run BDF_RESERVE_CUS=$R BDF_K1_COL=80 BDF_COL_QUEUE=0
  run BDF_RESERVE_CUS=$R BDF_K1_COL=128 BDF_COL_TIERS=HL BDF_LIB_PATH=$GRAFT_REPO_ROOT/bayesiandatafusion.jl_amd/csrc/variants/libbdf_w2.so
  run BDF_RESERVE_CUS=$R BDF_K1_COL=96 BDF_COL_TIERS=HL BDF_LIB_PATH=$GRAFT_REPO_ROOT/bayesiandatafusion.jl_amd/csrc/variants/libbdf_w2.so
done
cat $o/alone.txt

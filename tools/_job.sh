#!/bin/bash
root=$GRAFT_REPO_ROOT
for rep in 1 2; do
for so in base noprior; do
BDF_LIB_PATH=$root/bayesiandatafusion.jl_amd/csrc/variants/libbdf_$so.so python3 $root/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-c3 --no-c5 --no-mref --k1-min-launches 0 --c4-rows 2000000 --c4-cols 200000 --c4-nnz 20000000 --c4-sweeps 5 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$so', d['c4'].get('ms_per_sweep'), 'ms/sweep', d['c4'].get('error'))"
done; done

#!/bin/bash
# interleaved A/B of the library builds under csrc/variants: the row kernel alone (tools/k1_alone.py), then the bench (GPU box)
root=$(cd $(dirname $0)/.. && pwd)
for rep in 1 2; do
for so in $root/bayesiandatafusion.jl_amd/csrc/variants/libbdf_*.so; do
  echo "== $(basename $so)"; BDF_LIB_PATH=$so python3 $root/tools/k1_alone.py 2>&1 | grep -v amdgpu.ids | head -3
done; done
REPS=2 bash $root/tools/ab_libs.sh

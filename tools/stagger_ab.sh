#!/bin/bash
# K1 alone and the driver's bench line with every other workgroup of the row kernel starting late (GPU box)
R=$GRAFT_REPO_ROOT
for st in ${@:-0 10000 20000 30000 40000}; do
  echo "== BDF_K1_STAGGER=$st"
  BDF_K1_STAGGER=$st python3 $R/tools/k1_alone.py 2>&1 | grep -a "K1 alone\|launches only"
done

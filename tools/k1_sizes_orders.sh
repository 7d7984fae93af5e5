#!/bin/bash
# K1 alone (tools/k1_alone.py) over item / piece sizes x launch orders (GPU box): the longest single chain against the slot-time
for cfg in "192 128 1" "192 128 2" "128 128 1" "128 128 2" "128 96 2" "112 112 2" "96 96 2" "160 112 2" "144 144 2"; do
  set -- $cfg
  echo "== item $1 piece $2 order $3: $(ITEM=$1 PIECE=$2 BDF_K1_ORDER=$3 python3 $GRAFT_REPO_ROOT/tools/k1_alone.py 2>&1 | grep -a 'K1 alone\|launches only' | tr '\n' ' ')"
done

#!/bin/bash
# per-rank kernel statistics of the C4-shaped relation on one rank and on two (two ranks on ONE GPU: the test rig, host transport):
# k_hyper_partial per call -- one rank adds every row, each of two ranks its own half (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export BDF_RESERVE_CUS=0 C4_SWEEPS=2
rm -rf /tmp/cr1 /tmp/cr2_0 /tmp/cr2_1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cr1 -- python3 $R/tools/c4_ranks.py > /tmp/cr1.log 2>&1
export BDF_DIST_BACKEND=gloo WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
RANK=0 LOCAL_RANK=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cr2_0 -- python3 $R/tools/c4_ranks.py > /tmp/cr2_0.log 2>&1 &
RANK=1 LOCAL_RANK=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cr2_1 -- python3 $R/tools/c4_ranks.py > /tmp/cr2_1.log 2>&1 &
wait
grep -h '"world"' /tmp/cr1.log /tmp/cr2_0.log
python3 - <<'PY'
import csv, glob
for name, d in (("one rank", "/tmp/cr1"), ("two ranks, rank 0", "/tmp/cr2_0"), ("two ranks, rank 1", "/tmp/cr2_1")):
    f = glob.glob(d + "/*/*kernel_stats.csv")
    if not f:
        print(name, "no stats"); continue
    for r in csv.DictReader(open(f[0])):
        if "k_hyper_partial" in r["Name"] or "k_hyper_final" in r["Name"] or "k_sum_blocks" in r["Name"]:
            print(f"{name:20s} {r['Name'].replace('(anonymous namespace)::','')[:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY

#!/bin/bash
# The all-pairs peer exchange on the device's clock: two ranks (two processes on ONE GPU: the test rig; gloo carries the control
# messages) run a C4-shaped relation with BDF_COMM_PEER=1, each under rocprofv3 --kernel-trace --memory-copy-trace.  The timeline of
# one rank (kernels of the row stream and the peer copies, us from the first kernel shown) must show chunk c + 1's row kernels
# running WHILE chunk c's copy is in flight -- the exchange is ordered by interprocess events, no stream is synchronised.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-r06_peer}; mkdir -p $out
export BDF_RESERVE_CUS=0 C4_SWEEPS=2 BDF_COMM_PEER=1 BDF_DIST_BACKEND=gloo WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
export C4_SIZE=${C4_SIZE:-2000000,200000,20000000}
rm -rf /tmp/po_0 /tmp/po_1
RANK=0 LOCAL_RANK=0 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/po_0 -- python3 $R/tools/c4_ranks.py > $out/rank0.log 2>&1 &
RANK=1 LOCAL_RANK=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/po_1 -- python3 $R/tools/c4_ranks.py > $out/rank1.log 2>&1 &
wait
grep -h '"world"' $out/rank0.log
python3 $R/tools/peer_overlap_dump.py /tmp/po_0 > $out/peer_overlap_timeline.txt 2>&1
head -70 $out/peer_overlap_timeline.txt

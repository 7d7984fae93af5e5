#!/bin/bash
# host enqueue cost of the native sweep by NUMA node of the process (GPU box): the whole process confined to node 0 / node 1 cores
echo "visible: HIP=$HIP_VISIBLE_DEVICES ROCR=$ROCR_VISIBLE_DEVICES"; nproc
python3 - <<'PY'
import torch, glob, os
p = torch.cuda.get_device_properties(0)
print("pci", getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None), getattr(p, "pci_domain_id", None))
PY
for n in 0 1 0 1 0 1; do
  if [ $n = 0 ]; then cpus="0-63,128-191"; else cpus="64-127,192-255"; fi
  BDF_BENCH_PIN=0 BDF_DEBUG=1 taskset -c $cpus python3 bench.py --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref --k1-min-launches 0 2>/tmp/e.txt | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('node $n:', d['value'], 'sweeps/s', d['ms_per_step'], 'ms')"; grep "sweeps:" /tmp/e.txt
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 40); do
BDF_DEBUG=1 BDF_BENCH_DEBUG=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-c3 --no-c5 --no-mref 2>/tmp/e.txt | tail -1 > /tmp/o.txt
python3 - <<'PY'
import json
d=json.loads(open('/tmp/o.txt').read().strip().splitlines()[-1])
v=d['value']
line=[l for l in open('/tmp/e.txt', errors='replace') if 'timed region' in l]
host=[l for l in open('/tmp/e.txt', errors='replace') if 'host enqueue' in l]
print(round(v), 'K1', d['roofline']['avg_launch_us'], 'core', d['config']['host_core'], (line[-1].strip()[:400] if v < 10300 and line else ''), (host[-1].strip()[-90:] if v < 10300 and host else ''))
PY
done

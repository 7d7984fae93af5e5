# interleaved A/B of env settings on the bench workload: every config run in rotation; prints sweeps/s and K1 launch time
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for cfg in "X=1" "BDF_NO_PAIR_SORT=1"; do
  echo "$cfg: $(env $cfg python3 bench.py --no-cpu-baseline --warmup 300 --steps 300 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['avg_launch_us'])")"
done; done

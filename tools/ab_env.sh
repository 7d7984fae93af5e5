# interleaved A/B of env settings on the bench workload: every config run in rotation; prints sweeps/s and K1 launch time
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "X=1" "BDF_PIECE_SIZE=128" "BDF_PIECE_SIZE=96" "BDF_PIECE_SIZE=80" "BDF_ITEM_SIZE=256 BDF_PIECE_SIZE=96" "BDF_ITEM_SIZE=160 BDF_PIECE_SIZE=80" "BDF_ITEM_SIZE=224 BDF_PIECE_SIZE=112"; do
  echo "$cfg: $(env $cfg python3 bench.py --no-cpu-baseline --warmup 150 --steps 300 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['avg_launch_us'])")"
done; done

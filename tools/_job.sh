cd $GRAFT_REPO_ROOT
SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -1 gpurun_out/bench_full.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value', d['value'], d['ms_per_step'], 'rmse', d['test_rmse'])
print('roofline', d['roofline'])
print('cpu', d.get('cpu_baseline'))
print('c4', d.get('c4'))
print('c3', d.get('c3'))
print('mref', d.get('mref'))
"
echo "wall ${SECONDS}s"; tail -3 gpurun_out/bench_full.err | cut -c1-300

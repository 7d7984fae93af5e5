cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rows.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^E " | tail -4 > gpurun_out/look2.txt
for p in factor_probe_b factor_probe64_b; do echo "== $p"; timeout 120 tools/bin/$p | grep -E "factor\+backward .*waves/SIMD (1|2|7):"; done >> gpurun_out/look2.txt 2>&1
bash tools/ab_k1_alone.sh 2>&1 | grep -E "==|K1 alone|sweeps" >> gpurun_out/look2.txt
for v in base look2; do BDF_LIB_PATH=$PWD/bayesiandatafusion.jl_amd/csrc/variants/libbdf_$v.so python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c3 --no-mref --k1-min-launches 0 --c4-rows 2000000 --c4-cols 200000 --c4-nnz 20000000 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['c4']; print('$v', d['value'], 'sweeps/s | c4', c.get('ms_per_sweep'), 'ms/sweep rmse', c.get('test_rmse'), c.get('error'))" >> gpurun_out/look2.txt; done
cat gpurun_out/look2.txt

cd $GRAFT_REPO_ROOT
python3 tools/c4_quality_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/c4_quality.txt
python3 tools/c3_probe.py ff 2>&1 | grep -v amdgpu.ids >> gpurun_out/c4_quality.txt
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_features.py -x -q -m gpu 2>&1 | tail -2 >> gpurun_out/c4_quality.txt
cat gpurun_out/c4_quality.txt

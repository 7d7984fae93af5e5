#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_rows.py -q -m gpu -x -k "four_rows" 2>&1 | grep -a "passed\|failed\|rror\|assert\|Mismatch\|Max " | tail -12
BDF_K1_SMALL_MIN_ROWS=1 python -m pytest tests/test_gpu_rows.py tests/test_gpu_macau.py tests/test_gpu_golden.py -q -m gpu -x 2>&1 | grep -a "passed\|failed\|rror" | tail -4
python3 tools/mref_probe.py 2>&1 | grep -a M-ref

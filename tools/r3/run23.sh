#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py tests/test_gemm_variants_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 300 python tools/bench_gemm_threshold.py child 2>&1 | grep min_tiles
FI_GEMM_RAND_SCALES=1 timeout -k 10 300 python tools/bench_gemm_threshold.py child 2>&1 | grep min_tiles
FI_GEMM_SHAPES=small timeout -k 10 300 python tools/bench_gemm_threshold.py child 2>&1 | grep min_tiles
FI_GEMM_SHAPES=small FI_GEMM_RAND_SCALES=1 timeout -k 10 300 python tools/bench_gemm_threshold.py child 2>&1 | grep min_tiles

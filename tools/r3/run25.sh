#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > /tmp/tall2.py <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import bench_gemm
q = os.environ.get("FI_GEMM_RAND_SCALES") != "1"
for g, m, n, k in [(16, 128, 8192, 8192), (8, 128, 4096, 4096), (16, 128, 4096, 8192), (256, 64, 4096, 7168), (256, 32, 4096, 7168), (64, 96, 4096, 7168), (16, 512, 8192, 8192)]:
    bench_gemm.run(g=g, m=m, n=n, k=k, tag=os.environ.get("TAG", ""), quantised=q)
PY
TAG=default timeout -k 10 300 python /tmp/tall2.py 2>&1 | grep TFLOP
TAG=128x128-only FI_GEMM_WS_MIN_TILES=100000000 FI_GEMM_BIG=0 timeout -k 10 300 python /tmp/tall2.py 2>&1 | grep TFLOP
TAG=default FI_GEMM_RAND_SCALES=1 timeout -k 10 300 python /tmp/tall2.py 2>&1 | grep TFLOP
TAG=128x128-only FI_GEMM_RAND_SCALES=1 FI_GEMM_WS_MIN_TILES=100000000 FI_GEMM_BIG=0 timeout -k 10 300 python /tmp/tall2.py 2>&1 | grep TFLOP

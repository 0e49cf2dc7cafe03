#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > /tmp/tall.py <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import bench_gemm
q = os.environ.get("FI_GEMM_RAND_SCALES") != "1"
for g, m, n, k in [(256, 128, 4096, 7168), (64, 128, 4096, 7168), (256, 64, 4096, 7168), (16, 128, 8192, 8192), (256, 160, 2048, 7168), (32, 96, 7168, 2048)]:
    bench_gemm.run(g=g, m=m, n=n, k=k, tag=os.environ.get("TAG", ""), quantised=q)
PY
TAG=default timeout -k 10 300 python /tmp/tall.py 2>&1 | grep TFLOP

TAG=default FI_GEMM_RAND_SCALES=1 timeout -k 10 300 python /tmp/tall.py 2>&1 | grep TFLOP

timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_gemm_variants_gpu.py -x -q -m gpu 2>&1 | tail -3

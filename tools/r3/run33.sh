#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_prefill_gpu.py -x -q -m gpu -k "256 or head_dim" 2>&1 | tail -2 || exit 1
timeout -k 10 600 python tools/bench_prefill_dims.py 2>&1 | grep -v amdgpu | tail -14

#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/r3/sgrid.py 2>&1 | grep TFLOP
FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_before.so timeout -k 10 300 python tools/r3/sgrid.py 2>&1 | grep TFLOP | sed "s/^/before: /"
timeout -k 10 1000 python -m pytest tests/test_prefill_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -2

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_prefill_gpu.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -k "fp8 or c3" > gpurun_out/r3_run9_tests.log 2>&1
rc=$?; tail -8 gpurun_out/r3_run9_tests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python tools/bench_prefill_dims.py 2>&1 | grep -v amdgpu.ids
FI_PREFILL_FP8_NATIVE_D256=0 timeout -k 10 300 python tools/bench_prefill_dims.py 2>&1 | grep "fp8"

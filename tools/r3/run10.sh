#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_prefill_gpu.py tests/test_fuzz_gpu.py tests/test_custom_mask_gpu.py tests/test_graph_replan_gpu.py tests/test_ref_golden_gpu.py -m gpu -q -x > gpurun_out/r3_run10_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r3_run10_tests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
FI_FUZZ_SEEDS=150 timeout -k 10 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x -k prefill > gpurun_out/r3_run10_fuzz.log 2>&1; tail -2 gpurun_out/r3_run10_fuzz.log
timeout -k 10 300 python tools/bench_prefill_dims.py 2>&1 | grep -v amdgpu.ids

#!/bin/bash
# r3 GPU call 5: GEMM parity (hardware-scale path) + C4 timing with / without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py tests/test_gemm_variants_gpu.py tests/test_full_size_gpu.py tests/test_ref_golden_gpu.py -m gpu -q -x -k "gemm or c4" > gpurun_out/r3_run5_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r3_run5_tests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
for hw in 1 0 1 0; do
  echo "HW_SCALES=$hw"
  FI_GEMM_HW_SCALES=$hw timeout -k 10 300 python tools/bench_c4.py more > gpurun_out/r3_run5_bench_hw$hw.log 2>&1 || exit 1
  grep -v amdgpu.ids gpurun_out/r3_run5_bench_hw$hw.log
done

#!/bin/bash
# r3 GPU call 1: fp8 prefill parity (lagged P.V pipeline) + C3 A/B timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_prefill_gpu.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -k "fp8 or c3" -s > gpurun_out/r3_run1_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r3_run1_tests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
for lag in 1 0 1 0; do
  FI_PREFILL_FP8_LAG=$lag timeout -k 10 300 python tools/bench_c3.py 3 quick > gpurun_out/r3_run1_bench_lag$lag.log 2>&1 || exit 1
  echo "LAG=$lag"; tail -3 gpurun_out/r3_run1_bench_lag$lag.log
done

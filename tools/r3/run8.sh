#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_full_size_gpu.py tests/test_gemm_gpu.py -m gpu -q -x -k "c4 or gemm" > gpurun_out/r3_run8_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r3_run8_tests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
for rep in 1 2; do
for v in new old; do
  if [ $v = new ]; then unset FI_MI355_LIB; else export FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_gemm_big_0.so; fi
  echo "variant=$v"
  timeout -k 10 300 python tools/bench_c4.py more 2>&1 | grep -v amdgpu.ids || exit 1
done
done
cd /tmp && export TMPDIR=/tmp
for v in new old; do
  if [ $v = new ]; then unset FI_MI355_LIB; else export FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_gemm_big_0.so; fi
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_run8_pmc_$v -- python3 $GRAFT_REPO_ROOT/tools/prof_secondary_once.py c4 > /dev/null 2>&1
  echo "pmc variant=$v"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py group_gemm_fp8_big_kernel $GRAFT_REPO_ROOT/gpurun_out/r3_run8_pmc_$v
done

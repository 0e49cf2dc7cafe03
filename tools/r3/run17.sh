#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_gemm_variants_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for rep in 1 2; do
for v in new old; do
  if [ $v = new ]; then unset FI_MI355_LIB; else export FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_gemm_big_0.so; fi
  echo "variant=$v (new = range-checked output stores, counted in the tile prologue wait)"
  timeout -k 10 300 python tools/bench_c4.py 2>&1 | grep -v amdgpu.ids || exit 1
done
done
unset FI_MI355_LIB; timeout -k 10 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tail -12

#!/bin/bash
cd $GRAFT_REPO_ROOT
KO=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_gemm_big_1.so
FI_MI355_LIB=$KO timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then unset FI_MI355_LIB; else export FI_MI355_LIB=$KO; fi
  echo "variant=$v (new = DMA pieces two per n block, branch-free)"
  timeout -k 10 300 python tools/bench_c4.py 2>&1 | grep -v amdgpu.ids || exit 1
done
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in new old; do
  if [ $v = new ]; then unset FI_MI355_LIB; else export FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_gemm_big_0.so; fi
  echo "variant=$v (new = staggered DMA + compile-time out dtype)"
  timeout -k 10 300 python tools/bench_c4.py 2>&1 | grep -v amdgpu.ids || exit 1
done
done

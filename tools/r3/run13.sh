#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in default 8 16 24 4; do
  if [ $v = default ]; then unset FI_MI355_LIB; else export FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_gemm_big_$v.so; fi
  echo "KO=$v (8 no DMA wait, 16 DMA of k block 0 every step, 4 no DMA)"
  timeout -k 10 300 python tools/bench_c4.py 2>&1 | grep -v amdgpu.ids || exit 1
done

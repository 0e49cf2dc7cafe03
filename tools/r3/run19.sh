#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu 2>&1 | tail -2 || exit 1
FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_gemm_big_2.so timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu 2>&1 | tail -2 || exit 1
for rep in 1 2; do
for v in 1 0 2; do
  if [ $v = 1 ]; then unset FI_MI355_LIB; else export FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_gemm_big_$v.so; fi
  echo "carry=$v (0: none  1: two MFMAs held back behind the barrier  2: and the DMA issue behind them)"
  timeout -k 10 300 python tools/bench_c4.py 2>&1 | grep "ref-quantised" || exit 1
done
done

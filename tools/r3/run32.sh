#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 1024 512 2048; do
  if [ $v = 1024 ]; then unset FI_MI355_LIB; else export FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_gemm_big_$v.so; fi
  echo "band rows=$v"
  timeout -k 10 300 python tools/bench_c4.py 2>&1 | grep "ref-quantised" || exit 1
done
done

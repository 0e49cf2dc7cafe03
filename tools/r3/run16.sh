#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 4 5 6 7 8; do
  if [ $v = 0 ]; then unset FI_MI355_LIB; else export FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_gemm_big_$v.so; fi
  echo "spread=$v (4: 4,4,0,0  5: 8,0,0,0  6: 6,2,0,0  7: 4 top + 4 nb0  8: 8 top)"
  timeout -k 10 300 python tools/bench_c4.py 2>&1 | grep "ref-quantised" || exit 1
done
done

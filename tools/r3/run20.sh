#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in new old pk new old pk; do
  if [ $v = new ]; then unset FI_MI355_LIB; elif [ $v = old ]; then export FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_prefill_fp8_inst_0.so; else export FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_prefill_fp8_inst_1.so; fi
  echo "variant=$v (new = log-domain P codes, pk = with v_pk_fma_f32)"
  timeout -k 10 300 python tools/bench_c3.py 2 quick 2>&1 | grep -v amdgpu.ids || exit 1
done
unset FI_MI355_LIB
timeout -k 10 900 python -m pytest tests -q -m gpu -k "fp8" 2>&1 | grep -E "^FAILED|passed|failed|Greatest" | head -40

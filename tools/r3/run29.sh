#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in new old new old; do
  if [ $v = new ]; then unset FI_MI355_LIB; else export FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_noskip.so; fi
  echo "variant=$v (new = waves without query rows skip the MFMAs and the softmax)"
  timeout -k 10 300 python tools/bench_prefill.py 2>&1 | grep -E "bf16|fp16" || exit 1
done
unset FI_MI355_LIB
timeout -k 10 600 python tools/bench_ref_grids.py mixed 2>&1 | grep -E " 128 +[0-9.]+ +[0-9.]+ +[0-9.]+$"
timeout -k 10 1000 python -m pytest tests/test_prefill_gpu.py tests/test_fuzz_gpu.py tests/test_page_cascade_gpu.py -x -q -m gpu 2>&1 | tail -2

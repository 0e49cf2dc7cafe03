#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > /tmp/d256.py <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import torch
from bench_prefill import run
run(torch.bfloat16, b=16, qo=2048, kv=8192, hq=16, hkv=8, d=256, tag="bf16 d256 G=2 " + os.environ.get("TAG", ""))
PY
for rep in 1 2; do
for v in default 6_4 8_6 8_8; do
  if [ $v = default ]; then unset FI_MI355_LIB; else export FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_pf256_$v.so; fi
  TAG="prefetch(qk_pv)=$v" timeout -k 10 300 python /tmp/d256.py 2>&1 | grep TFLOP || exit 1
done
done

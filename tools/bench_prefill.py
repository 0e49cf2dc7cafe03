"""Prefill benchmark: BASELINE config C3 (fp8 e4m3 causal, qo 2048, kv 8192, bs 16, 32/8 heads, d128) and
its bf16 twin.  FLOPs = B*(2*kv - qo)*qo*Hq*(d_qk+d_vo) for causal (ref: flashinfer/testing/utils.py:280-297)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd"))
import torch
import flashinfer
from bench_decode_sweep import bench

DEV = torch.device("cuda:0")

def run(dtype, b=16, qo=2048, kv=8192, hq=32, hkv=8, d=128, ps=16, causal=True, tag=""):
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = b * kv // ps
    cache = torch.randn(npages, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16, generator=g).to(dtype)
    q = torch.randn(b * qo, hq, d, device=DEV, dtype=torch.bfloat16, generator=g).to(dtype)
    qo_indptr = (torch.arange(b + 1, dtype=torch.int32) * qo).to(DEV)
    indptr = (torch.arange(b + 1, dtype=torch.int32) * (kv // ps)).to(DEV)
    indices = torch.randperm(npages, device=DEV, generator=g).to(torch.int32)
    last = torch.full((b,), ps, dtype=torch.int32, device=DEV)
    ws = torch.zeros(128 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, "NHD")
    w.plan(qo_indptr, indptr, indices, last, hq, hkv, d, ps, causal=causal, q_data_type=dtype, kv_data_type=dtype,
           o_data_type=torch.bfloat16 if dtype != torch.float16 else torch.float16)
    out = torch.empty(b * qo, hq, d, device=DEV, dtype=torch.bfloat16 if dtype != torch.float16 else torch.float16)
    med, mn = bench(lambda: w.run(q, cache, out=out), iters=10, warm=3)
    flops = b * (2 * kv - qo) * qo * hq * 2 * d if causal else 2 * b * qo * kv * hq * 2 * d
    print(f"{tag:28s} {str(dtype):22s} causal={int(causal)} med={med:8.3f} ms min={mn:8.3f} ms  {flops/med/1e9:8.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    run(torch.bfloat16, tag="C3-shape bf16")
    run(torch.float8_e4m3fn, tag="C3 fp8 e4m3")
    run(torch.bfloat16, causal=False, tag="C3-shape bf16 non-causal")
    run(torch.bfloat16, b=32, qo=1024, kv=1024, hq=64, hkv=8, tag="ref sample shape (bs32 1k/1k 64/8)")
    run(torch.float16, b=4, qo=8192, kv=8192, tag="bs4 8k/8k fp16")

"""Prefill with 16-bit queries over an fp8 KV cache (the common 'kv_cache_dtype=fp8' deployment)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")
def run(kvdt, b=16, qo=2048, kv=8192, hq=32, hkv=8, d=128, ps=16):
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = b * kv // ps
    cache = torch.randn(npages, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16, generator=g).to(kvdt)
    q = torch.randn(b * qo, hq, d, device=DEV, dtype=torch.bfloat16, generator=g)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(torch.zeros(256 << 20, dtype=torch.uint8, device=DEV), "NHD")
    w.plan((torch.arange(b + 1, dtype=torch.int32) * qo).to(DEV), (torch.arange(b + 1, dtype=torch.int32) * (kv // ps)).to(DEV),
           torch.randperm(npages, device=DEV, generator=g).to(torch.int32), torch.full((b,), ps, dtype=torch.int32, device=DEV),
           hq, hkv, d, ps, causal=True, q_data_type=torch.bfloat16, kv_data_type=kvdt)
    med, _ = bench(lambda: w.run(q, cache), iters=10, warm=3)
    fl = b * (2 * kv - qo) * qo * hq * 2 * d
    print(f"bf16 q, kv {str(kvdt):22s}: {med:7.3f} ms  {fl/med/1e9:7.1f} TFLOP/s", flush=True)
run(torch.bfloat16); run(torch.float8_e4m3fn); run(torch.float8_e5m2)

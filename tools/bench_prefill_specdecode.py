"""Speculative-decoding-sized prefill: many requests, a few query tokens each, long contexts (HBM-bound like decode)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")
def run(b, qo, kv, hq=32, hkv=8, d=128, ps=16, dt=torch.bfloat16):
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = b * kv // ps
    cache = torch.randn(npages, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16, generator=g).to(dt)
    q = torch.randn(b * qo, hq, d, device=DEV, dtype=torch.bfloat16, generator=g)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(torch.zeros(256 << 20, dtype=torch.uint8, device=DEV), "NHD")
    w.plan((torch.arange(b + 1, dtype=torch.int32) * qo).to(DEV), (torch.arange(b + 1, dtype=torch.int32) * (kv // ps)).to(DEV),
           torch.randperm(npages, device=DEV, generator=g).to(torch.int32), torch.full((b,), ps, dtype=torch.int32, device=DEV),
           hq, hkv, d, ps, causal=True, q_data_type=torch.bfloat16, kv_data_type=dt)
    med, _ = bench(lambda: w.run(q, cache), iters=20, warm=5)
    nbytes = 2 * b * kv * hkv * d * cache.element_size()
    print(f"bs={b:3d} qo={qo:3d} kv={kv:6d}: {med*1e3:8.1f} us  {nbytes/med/1e6:7.1f} GB/s of KV  (split={w._plan_info[14]})", flush=True)
for qo in (1, 4, 8, 16, 32):
    run(64, qo, 8192)
run(8, 8, 32768); run(256, 4, 2048)

"""Cost of the logits variants (soft cap, ALiBi, custom mask, sliding window) against the plain kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")
hq, hkv, d, ps = 32, 8, 128, 16
def prefill(b, qo, kv, tag, **kw):
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = b * kv // ps
    cache = torch.randn(npages, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16, generator=g)
    q = torch.randn(b * qo, hq, d, device=DEV, dtype=torch.bfloat16, generator=g)
    if kw.pop("mask", False):
        kw["custom_mask"] = torch.tril(torch.ones(qo, kv, dtype=torch.bool, device=DEV), diagonal=kv - qo).repeat(b, 1, 1).view(-1)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(torch.zeros(256 << 20, dtype=torch.uint8, device=DEV), "NHD")
    w.plan((torch.arange(b + 1, dtype=torch.int32) * qo).to(DEV), (torch.arange(b + 1, dtype=torch.int32) * (kv // ps)).to(DEV),
           torch.randperm(npages, device=DEV, generator=g).to(torch.int32), torch.full((b,), ps, dtype=torch.int32, device=DEV),
           hq, hkv, d, ps, causal=True, q_data_type=torch.bfloat16, **kw)
    med, _ = bench(lambda: w.run(q, cache), iters=10, warm=3)
    print(f"prefill bs={b} qo={qo} kv={kv} {tag:12s}: {med:8.3f} ms", flush=True)
def decode(b, L, tag, **kw):
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = b * L // ps
    cache = torch.randn(npages, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16, generator=g)
    q = torch.randn(b, hq, d, device=DEV, dtype=torch.bfloat16, generator=g)
    w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(torch.zeros(128 << 20, dtype=torch.uint8, device=DEV), "NHD")
    w.plan((torch.arange(b + 1, dtype=torch.int32) * (L // ps)).to(DEV), torch.randperm(npages, device=DEV, generator=g).to(torch.int32),
           torch.full((b,), ps, dtype=torch.int32, device=DEV), hq, hkv, d, ps, q_data_type=torch.bfloat16, **kw)
    med, _ = bench(lambda: w.run(q, cache))
    print(f"decode bs={b} kv={L} {tag:12s}: {med*1e3:8.1f} us  {2*b*L*hkv*d*2/med/1e6:7.1f} GB/s", flush=True)
prefill(16, 2048, 8192, "plain"); prefill(16, 2048, 8192, "soft cap", logits_soft_cap=30.0)
prefill(16, 2048, 8192, "alibi", pos_encoding_mode="ALIBI"); prefill(4, 2048, 8192, "plain bs4"); prefill(4, 2048, 8192, "custom mask", mask=True)
decode(64, 8192, "plain"); decode(64, 8192, "soft cap", logits_soft_cap=30.0); decode(64, 8192, "alibi", pos_encoding_mode="ALIBI")

"""Cost of the fused-RoPE variants (pos_encoding_mode="ROPE_LLAMA") against the plain kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")
hq, hkv, d, ps = 32, 8, 128, 16
def prefill(b, qo, kv, mode):
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = b * kv // ps
    cache = torch.randn(npages, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16, generator=g)
    q = torch.randn(b * qo, hq, d, device=DEV, dtype=torch.bfloat16, generator=g)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(torch.zeros(256 << 20, dtype=torch.uint8, device=DEV), "NHD")
    w.plan((torch.arange(b + 1, dtype=torch.int32) * qo).to(DEV), (torch.arange(b + 1, dtype=torch.int32) * (kv // ps)).to(DEV),
           torch.randperm(npages, device=DEV, generator=g).to(torch.int32), torch.full((b,), ps, dtype=torch.int32, device=DEV),
           hq, hkv, d, ps, causal=True, pos_encoding_mode=mode, q_data_type=torch.bfloat16)
    med, _ = bench(lambda: w.run(q, cache), iters=10, warm=3)
    fl = b * (2 * kv - qo) * qo * hq * 2 * d
    print(f"prefill bs={b} qo={qo} kv={kv} {mode:10s}: {med:8.3f} ms  {fl/med/1e9:7.1f} TFLOP/s", flush=True)
def decode(b, L, mode):
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = b * L // ps
    cache = torch.randn(npages, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16, generator=g)
    q = torch.randn(b, hq, d, device=DEV, dtype=torch.bfloat16, generator=g)
    w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(torch.zeros(128 << 20, dtype=torch.uint8, device=DEV), "NHD")
    w.plan((torch.arange(b + 1, dtype=torch.int32) * (L // ps)).to(DEV), torch.randperm(npages, device=DEV, generator=g).to(torch.int32),
           torch.full((b,), ps, dtype=torch.int32, device=DEV), hq, hkv, d, ps, pos_encoding_mode=mode, q_data_type=torch.bfloat16)
    med, _ = bench(lambda: w.run(q, cache))
    print(f"decode bs={b} kv={L} {mode:10s}: {med*1e3:8.1f} us  {2*b*L*hkv*d*2/med/1e6:7.1f} GB/s", flush=True)
for m in ("NONE", "ROPE_LLAMA"):
    prefill(16, 2048, 8192, m)
for m in ("NONE", "ROPE_LLAMA"):
    decode(64, 8192, m)

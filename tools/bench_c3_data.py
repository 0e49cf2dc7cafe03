"""C3 (fp8 causal batch prefill) on different DATA, same instruction stream: random e4m3 values, all zeros, and a
constant.  A kernel that is power / clock bound runs faster on zeros (the matrix pipe toggles nothing and the chip holds
a higher clock: MI355X_MICROARCH.md 'DVFS give-back'); one that is issue bound does not."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")

def run(kind, dtype=torch.float8_e4m3fn, b=16, qo=2048, kv=8192, hq=32, hkv=8, d=128, ps=16):
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = b * kv // ps
    if kind == "random":
        cache = torch.randn(npages, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16, generator=g).to(dtype)
        q = torch.randn(b * qo, hq, d, device=DEV, dtype=torch.bfloat16, generator=g).to(dtype)
    elif kind == "zeros":
        cache = torch.zeros(npages, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16).to(dtype)
        q = torch.zeros(b * qo, hq, d, device=DEV, dtype=torch.bfloat16).to(dtype)
    elif kind == "ones":
        cache = torch.ones(npages, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16).to(dtype)
        q = torch.full((b * qo, hq, d), 0.0625, device=DEV, dtype=torch.bfloat16).to(dtype)
    else:  # wide: values spread over the whole e4m3 range (what per-head amax quantisation produces)
        cache = (torch.randn(npages, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16, generator=g) * 100).to(dtype)
        q = (torch.randn(b * qo, hq, d, device=DEV, dtype=torch.bfloat16, generator=g) * 100).to(dtype)
    qo_indptr = (torch.arange(b + 1, dtype=torch.int32) * qo).to(DEV)
    indptr = (torch.arange(b + 1, dtype=torch.int32) * (kv // ps)).to(DEV)
    indices = torch.randperm(npages, device=DEV, generator=g).to(torch.int32)
    last = torch.full((b,), ps, dtype=torch.int32, device=DEV)
    ws = torch.zeros(128 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, "NHD")
    w.plan(qo_indptr, indptr, indices, last, hq, hkv, d, ps, causal=True, q_data_type=dtype, kv_data_type=dtype,
           o_data_type=torch.bfloat16, sm_scale=(1e-4 if kind == "wide" else None))
    out = torch.empty(b * qo, hq, d, device=DEV, dtype=torch.bfloat16)
    med, mn = bench(lambda: w.run(q, cache, out=out), iters=10, warm=3)
    flops = b * (2 * kv - qo) * qo * hq * 2 * d
    print(f"C3 data={kind:8s} med={med:8.3f} ms min={mn:8.3f} ms  {flops/med/1e9:8.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    for kind in ("random", "zeros", "ones", "wide", "random"):
        run(kind)

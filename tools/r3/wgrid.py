import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")
def prefill(bs, qo, kv, window, soft_cap=0.0, hq=32, hkv=8, d=128, page=16, dtype=torch.bfloat16):
    pages_per = (kv + page - 1) // page; nb = bs * pages_per
    q = torch.randn(bs * qo, hq, d, device=DEV, dtype=dtype); cache = torch.randn(nb, 2, page, hkv, d, device=DEV, dtype=dtype)
    qo_indptr = (torch.arange(bs + 1, dtype=torch.int32) * qo).to(DEV); kv_indptr = (torch.arange(bs + 1, dtype=torch.int32) * pages_per).to(DEV)
    last = torch.full((bs,), (kv - 1) % page + 1, dtype=torch.int32, device=DEV)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(torch.empty(256 << 20, dtype=torch.uint8, device=DEV), "NHD")
    w.plan(qo_indptr, kv_indptr, torch.arange(nb, dtype=torch.int32, device=DEV), last, hq, hkv, d, page, causal=True, window_left=window, logits_soft_cap=soft_cap, q_data_type=dtype, kv_data_type=dtype)
    med, _ = bench(lambda: w.run(q, cache), iters=7, warm=2)
    eff = sum(min(kv - qo + i + 1, (window + 1) if window >= 0 else 10**9) for i in range(qo)) * bs * hq * 4 * d
    print(f"prefill bs={bs} qo={qo} kv={kv} window={window} cap={soft_cap} {med*1e3:8.1f} us  {eff/med/1e9:8.1f} TFLOP/s (visible keys only)", flush=True)
def decode(bs, kv, window, hq=32, hkv=8, d=128, page=16, dtype=torch.bfloat16):
    pages_per = (kv + page - 1) // page; nb = bs * pages_per
    q = torch.randn(bs, hq, d, device=DEV, dtype=dtype); cache = torch.randn(nb, 2, page, hkv, d, device=DEV, dtype=dtype)
    kv_indptr = (torch.arange(bs + 1, dtype=torch.int32) * pages_per).to(DEV); last = torch.full((bs,), (kv - 1) % page + 1, dtype=torch.int32, device=DEV)
    w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(torch.empty(256 << 20, dtype=torch.uint8, device=DEV), "NHD")
    w.plan(kv_indptr, torch.arange(nb, dtype=torch.int32, device=DEV), last, hq, hkv, d, page, window_left=window, q_data_type=dtype, kv_data_type=dtype)
    med, _ = bench(lambda: w.run(q, cache), iters=15, warm=3)
    vis = min(kv, window + 1) if window >= 0 else kv
    print(f"decode bs={bs} kv={kv} window={window} {med*1e3:8.1f} us  {bs*vis*hkv*d*4/med/1e9:8.3f} TB/s of visible KV", flush=True)
for window in (-1, 4095, 1023, 255):
    prefill(16, 2048, 8192, window)
prefill(16, 2048, 8192, -1, soft_cap=30.0)
prefill(4, 8192, 8192, 1023)
for window in (-1, 4095, 1023, 255):
    decode(64, 8192, window)
decode(256, 32768 // 8, 511)

"""Sliding-window decode on long contexts: the planner spreads only the window's pages over the grid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")
def run(b, L, wl, hq=32, hkv=8, d=128, ps=16):
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = b * L // ps
    cache = torch.randn(npages, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16, generator=g)
    q = torch.randn(b, hq, d, device=DEV, dtype=torch.bfloat16, generator=g)
    w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(torch.zeros(128 << 20, dtype=torch.uint8, device=DEV), "NHD")
    w.plan((torch.arange(b + 1, dtype=torch.int32) * (L // ps)).to(DEV), torch.randperm(npages, device=DEV, generator=g).to(torch.int32),
           torch.full((b,), ps, dtype=torch.int32, device=DEV), hq, hkv, d, ps, window_left=wl, q_data_type=torch.bfloat16)
    med, mn = bench(lambda: w.run(q, cache))
    vis = min(L, wl + 1) if wl >= 0 else L
    print(f"bs={b} kv={L} window_left={wl:6d}: {med*1e3:8.1f} us  ({2*b*vis*hkv*d*2/med/1e6:8.1f} GB/s of visible KV) chunk={w._plan_info[10]} work={w._plan_info[11]}", flush=True)
run(4, 65536, -1); run(4, 65536, 4095); run(64, 8192, 1023); run(1, 131072, 8191)

"""Tuning sweep for the C2 decode workload: planner grid size (waves/CU), page order, layout."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd"))
import torch
import flashinfer

DEV = torch.device("cuda:0")

def bench(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        s[i].record(); fn(); e[i].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in zip(s, e))
    return ts[len(ts)//2], ts[0]

def run(b=64, L=8192, hq=32, hkv=8, d=128, ps=16, layout="NHD", permute=True, dtype=torch.bfloat16, wpc=None, tag=""):
    if wpc: os.environ["FI_DECODE_WAVES_PER_CU"] = str(wpc)
    else: os.environ.pop("FI_DECODE_WAVES_PER_CU", None)
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = b * L // ps
    shape = (npages, 2, ps, hkv, d) if layout == "NHD" else (npages, 2, hkv, ps, d)
    cache = torch.randn(shape, device=DEV, dtype=torch.bfloat16, generator=g).to(dtype)
    q = torch.randn(b, hq, d, device=DEV, dtype=torch.bfloat16, generator=g)
    indptr = (torch.arange(b + 1, dtype=torch.int32) * (L // ps)).to(DEV)
    indices = (torch.randperm(npages, device=DEV, generator=g) if permute else torch.arange(npages, device=DEV)).to(torch.int32)
    last = torch.full((b,), ps, dtype=torch.int32, device=DEV)
    ws = torch.zeros(256 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, layout)
    w.plan(indptr, indices, last, hq, hkv, d, ps, q_data_type=torch.bfloat16, kv_data_type=dtype)
    out = torch.empty_like(q)
    med, mn = bench(lambda: w.run(q, cache, out=out))
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): w.run(q, cache, out=out)
    host_us = (time.perf_counter() - t0) / 50 * 1e6
    torch.cuda.synchronize()
    nbytes = 2 * b * L * hkv * d * cache.element_size() + 2 * q.numel() * 2
    print(f"{tag:34s} wpc={wpc} layout={layout} perm={int(permute)} chunk={w._plan_info[10]:5d} work={w._plan_info[11]:5d} "
          f"host={host_us:6.1f}us med={med*1e3:7.1f}us min={mn*1e3:7.1f}us  {nbytes/med/1e6:7.1f} GB/s (min-time {nbytes/mn/1e6:7.1f})", flush=True)
    del cache

if __name__ == "__main__":
    for wpc in (4, 6, 8, 12, 16, 24, 32):
        run(wpc=wpc, tag="C2 bf16")
    run(wpc=8, permute=False, tag="C2 arange pages")
    run(wpc=8, layout="HND", tag="C2 HND")
    run(wpc=8, dtype=torch.float8_e4m3fn, tag="C2 fp8 kv")
    run(wpc=8, hq=8, tag="G=1 (hq=8)")
    run(wpc=8, hq=64, tag="G=8 (hq=64)")
    run(wpc=8, b=256, L=2048, tag="bs256 kv2048")
    run(wpc=8, b=16, L=32768, tag="bs16 kv32768")
    run(wpc=8, b=1, L=131072, tag="bs1 kv131072")

"""Short-sequence prefill shapes (prologue / epilogue dominated): A/B against another build via FI_MI355_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_prefill import run
for _ in range(2):
    run(torch.bfloat16, b=32, qo=1024, kv=1024, hq=64, hkv=8, tag="bs32 1k/1k 64/8")
    run(torch.bfloat16, b=64, qo=512, kv=512, hq=32, hkv=8, tag="bs64 512/512 32/8")
    run(torch.bfloat16, b=1, qo=128, kv=32768, hq=32, hkv=8, tag="bs1 append 128 on 32k")
    run(torch.bfloat16, b=4, qo=16, kv=16384, hq=32, hkv=8, tag="bs4 append 16 on 16k")
# sliding window: only the tiles inside the window are walked
import flashinfer
def run_window(b, qo, kv, wl, tag):
    import torch
    from bench_decode_sweep import bench
    DEV = torch.device("cuda:0"); hq, hkv, d, ps = 32, 8, 128, 16
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = b * kv // ps
    cache = torch.randn(npages, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16, generator=g)
    q = torch.randn(b * qo, hq, d, device=DEV, dtype=torch.bfloat16, generator=g)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(torch.zeros(256 << 20, dtype=torch.uint8, device=DEV), "NHD")
    w.plan((torch.arange(b + 1, dtype=torch.int32) * qo).to(DEV), (torch.arange(b + 1, dtype=torch.int32) * (kv // ps)).to(DEV),
           torch.randperm(npages, device=DEV, generator=g).to(torch.int32), torch.full((b,), ps, dtype=torch.int32, device=DEV),
           hq, hkv, d, ps, causal=True, window_left=wl, q_data_type=torch.bfloat16)
    med, mn = bench(lambda: w.run(q, cache), iters=10, warm=3)
    print(f"{tag:40s} window_left={wl:6d} med={med:8.3f} ms", flush=True)
run_window(4, 2048, 32768, -1, "bs4 qo2048 kv32768 full causal")
run_window(4, 2048, 32768, 1024, "bs4 qo2048 kv32768 sliding window")

import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")
def run(bs, qo, kv, hq=32, hkv=8, d=128, page=16, dtype=torch.bfloat16, causal=True, **kw):
    pages_per = (kv + page - 1) // page
    nb = bs * pages_per
    q = torch.randn(bs * qo, hq, d, device=DEV, dtype=dtype)
    cache = torch.randn(nb, 2, page, hkv, d, device=DEV, dtype=dtype)
    qo_indptr = (torch.arange(bs + 1, dtype=torch.int32) * qo).to(DEV)
    kv_indptr = (torch.arange(bs + 1, dtype=torch.int32) * pages_per).to(DEV)
    last = torch.full((bs,), (kv - 1) % page + 1, dtype=torch.int32, device=DEV)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, "NHD")
    w.plan(qo_indptr, kv_indptr, torch.arange(nb, dtype=torch.int32, device=DEV), last, hq, hkv, d, page, causal=causal, q_data_type=dtype, kv_data_type=dtype, **kw)
    med, _ = bench(lambda: w.run(q, cache), iters=9, warm=3)
    fl = bs * ((2 * kv - qo) * qo if causal else 2 * qo * kv) * hq * 2 * d
    print(f"bs={bs:4d} qo={qo:6d} kv={kv:6d} {str(kw):40s} chunk={w._plan_info[9]:6d} work={w._plan_info[12]:5d} {med*1e3:8.1f} us {fl / med / 1e9:8.1f} TFLOP/s", flush=True)
for (bs, qo, kv) in [(1, 512, 512), (1, 1024, 1024), (2, 1024, 1024), (1, 2048, 2048), (1, 4096, 4096), (1, 128, 4096), (4, 128, 2048), (1, 16, 8192), (8, 16, 2048), (1, 256, 8192), (1, 512, 8192), (1, 1024, 8192), (2, 512, 4096), (1, 128, 32768), (4, 128, 16384), (16, 64, 8192), (1, 4096, 65536)]:
    for kw in ({},):
        try:
            run(bs, qo, kv, **kw)
        except RuntimeError as e:
            print('skip', bs, qo, kv, kw, str(e)[:60], flush=True)

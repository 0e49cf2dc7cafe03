import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")
def run(bs, qo, kv, hq=32, hkv=8, d=128, page=16, dtype=torch.bfloat16, causal=True):
    pages_per = (kv + page - 1) // page
    nb = bs * pages_per
    q = torch.randn(bs * qo, hq, d, device=DEV, dtype=dtype)
    cache = torch.randn(nb, 2, page, hkv, d, device=DEV, dtype=dtype)
    qo_indptr = (torch.arange(bs + 1, dtype=torch.int32) * qo).to(DEV)
    kv_indptr = (torch.arange(bs + 1, dtype=torch.int32) * pages_per).to(DEV)
    last = torch.full((bs,), (kv - 1) % page + 1, dtype=torch.int32, device=DEV)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, "NHD")
    w.plan(qo_indptr, kv_indptr, torch.arange(nb, dtype=torch.int32, device=DEV), last, hq, hkv, d, page, causal=causal, q_data_type=dtype, kv_data_type=dtype)
    med, _ = bench(lambda: w.run(q, cache), iters=7, warm=2)
    fl = bs * ((2 * kv - qo) * qo if causal else 2 * qo * kv) * hq * 2 * d
    print(f"bs={bs:4d} qo={qo:6d} kv={kv:6d} {hq}/{hkv} d{d} {str(dtype)[6:]:9s} {med:8.3f} ms {fl / med / 1e9:8.1f} TFLOP/s", flush=True)
for bs in (1, 4, 16, 64):
    for s in (512, 2048, 8192):
        if bs * s > 262144: continue
        run(bs, s, s)
for (bs, qo, kv) in [(1, 128, 32768), (4, 128, 16384), (16, 64, 8192), (64, 16, 4096), (8, 512, 8192), (1, 4096, 65536), (256, 8, 2048), (32, 2048, 2048)]:
    run(bs, qo, kv)

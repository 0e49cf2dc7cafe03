import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")
for (qo, kv, hq, hkv) in [(512, 512, 32, 32), (1024, 1024, 32, 32), (2048, 2048, 32, 8), (512, 512, 32, 8), (1024, 1024, 32, 8), (128, 8192, 32, 8), (16, 16384, 32, 8), (1024, 8192, 32, 8), (4096, 4096, 32, 8), (256, 4096, 32, 8)]:
    for dt in (torch.float16,):
        q = torch.randn(qo, hq, 128, device=DEV, dtype=dt); k = torch.randn(kv, hkv, 128, device=DEV, dtype=dt); v = torch.randn(kv, hkv, 128, device=DEV, dtype=dt)
        med, _ = bench(lambda: flashinfer.single_prefill_with_kv_cache(q, k, v, causal=True), iters=9, warm=3)
        fl = (2 * kv - qo) * qo * hq * 256
        print(f"single prefill qo={qo:5d} kv={kv:6d} {hq}/{hkv} {med*1e3:8.1f} us {fl/med/1e9:8.1f} TFLOP/s", flush=True)

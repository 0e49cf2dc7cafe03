import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")
torch.manual_seed(0)
for (kv, hq, hkv) in [(2048, 32, 32), (8192, 32, 8), (32768, 32, 8), (131072, 32, 8), (512, 32, 32)]:
    q = torch.randn(hq, 128, device=DEV).half(); k = torch.randn(kv, hkv, 128, device=DEV).half(); v = torch.randn(kv, hkv, 128, device=DEV).half()
    med, mn = bench(lambda: flashinfer.single_decode_with_kv_cache(q, k, v), iters=20, warm=5)
    print(f"single_decode kv={kv:7d} {hq}/{hkv}  {med*1e3:7.1f} us (min {mn*1e3:6.1f})  {kv*hkv*128*4/med/1e9:7.3f} TB/s", flush=True)

import os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import bench_decode_sweep as S
import torch
for (hq, hkv) in [(8, 1), (16, 2), (4, 1)]:
    for (b, L) in [(1, 8192), (8, 8192), (64, 8192), (256, 8192), (64, 2048), (256, 2048), (16, 32768)]:
        S.run(b=b, L=L, hq=hq, hkv=hkv, permute=True, tag=f"bs{b} kv{L} {hq}/{hkv}")

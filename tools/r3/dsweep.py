import os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import bench_decode_sweep as S
import torch
for (b, L, hq, hkv, dt) in [(64, 8192, 32, 8, torch.bfloat16), (64, 8192, 32, 8, torch.float8_e4m3fn), (32, 4096, 32, 8, torch.bfloat16), (16, 8192, 32, 8, torch.bfloat16),
                        (64, 2048, 32, 4, torch.bfloat16), (256, 1024, 32, 4, torch.bfloat16), (64, 4096, 32, 4, torch.float8_e4m3fn), (16, 4096, 8, 8, torch.bfloat16), (32, 8192, 64, 8, torch.bfloat16), (8, 32768, 32, 8, torch.bfloat16)]:
    for wpc in (3, 4, 5, 6, 8):
        S.run(b=b, L=L, hq=hq, hkv=hkv, permute=True, dtype=dt, wpc=wpc, tag=f"bs{b} kv{L} {hq}/{hkv} {str(dt)[6:10]}")

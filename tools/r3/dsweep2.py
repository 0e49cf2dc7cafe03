import os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import bench_decode_sweep as S
for (b, L) in [(1, 512), (1, 2048), (1, 8192), (4, 512), (4, 2048), (16, 512), (16, 1024), (16, 2048), (64, 512)]:
    for wpc in (None, 1, 2):
        S.run(b=b, L=L, hq=32, hkv=4, permute=False, wpc=wpc, tag=f"bs{b} kv{L} 32/4")

"""From how many 256 x 256 tiles on does the 256 x 256 kernel (gemm_big.hip) beat the 256 x 128 LDS-DMA kernel?
Each threshold runs in a child process (the library reads FI_GEMM_BIG_MIN_TILES once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(8, 512, 4096, 7168), (8, 1024, 4096, 7168), (8, 1536, 4096, 7168), (1, 4096, 4096, 4096), (1, 8192, 4096, 4096),
          (1, 4096, 14336, 4096), (4, 1024, 7168, 2048), (16, 512, 7168, 2048), (8, 768, 14336, 4096)]
if os.environ.get("FI_GEMM_SHAPES") == "small":
    SHAPES = [(1, 2048, 4096, 4096), (1, 4096, 2048, 4096), (8, 256, 4096, 7168), (1, 1024, 4096, 4096), (4, 512, 2048, 2048),
              (1, 2048, 2048, 8192), (2, 1024, 7168, 2048), (1, 3072, 4096, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_gemm
    for g, m, n, k in SHAPES:
        bench_gemm.run(g=g, m=m, n=n, k=k, tag=f"min_tiles={os.environ.get('FI_GEMM_BIG_MIN_TILES', 'default')}",
                       quantised=os.environ.get("FI_GEMM_RAND_SCALES") != "1")
else:
    small = len(sys.argv) > 1 and sys.argv[1] == "small"
    for thr in ("100000", "0"):
        env = dict(os.environ, FI_GEMM_BIG_MIN_TILES=thr)
        if small:  # below the persistent kernels' own threshold: 128 x 128 kernel (default) against forced 256 x 256
            env["FI_GEMM_SHAPES"] = "small"
            if thr == "0": env["FI_GEMM_WS_MIN_TILES"] = "0"
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=600)
        print(r.stdout, flush=True)
        if r.returncode: print(r.stderr[-2000:])

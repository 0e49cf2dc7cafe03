"""Per-tile fixed cost of the GEMM kernels: the same (G, M, N) at several K; time = tiles * (fixed + ksteps * step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_gemm import run
for k in (2048, 4096, 8192):
    run(g=1, m=32768, n=14336, k=k, tag=f"one group, K={k}")
for k in (2048, 4096, 8192):
    run(k=k, tag=f"8 groups, K={k}")

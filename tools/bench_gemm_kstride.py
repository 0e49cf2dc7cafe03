"""Does a power-of-two row stride (K bytes per fp8 row) cost the GEMM L2-channel parallelism?  Same problem at
K = 4096 and at neighbouring K values."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_gemm import run
for k in (4096, 4224, 3968, 4352, 8192, 8320):
    run(k=k, tag=f"C4 shape, K={k}")

"""C4 only (fp8 groupwise grouped GEMM 8 x 4096 x 14336 x 4096), for A/B runs of library variants (FI_MI355_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm import run
run(tag=os.environ.get("FI_MI355_LIB", "default")[-28:])
run(tag=os.environ.get("FI_MI355_LIB", "default")[-28:], quantised=False)
if len(sys.argv) > 1 and sys.argv[1] == "more":
    run(g=1, m=8192, n=8192, k=8192, tag="square 8k")
    run(g=1, m=32768, n=14336, k=8192, tag="32k x 14336 x 8192")

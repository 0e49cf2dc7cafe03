"""C4: fp8 groupwise grouped GEMM, 8 groups x (M=4096, N=14336, K=4096), block 128.  FLOPs = 2*G*M*N*K
(ref: benchmarks/bench_groupwise_grouped_gemm_fp8_blackwell.py:51)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")

def run(g=8, m=4096, n=14336, k=4096, tag=""):
    gen = torch.Generator(device=DEV).manual_seed(0)
    a = torch.randn(g * m, k, device=DEV, generator=gen).to(torch.float8_e4m3fn)
    b = (torch.randn(g, n, k, device=DEV, generator=gen) / k ** 0.5).to(torch.float8_e4m3fn)
    sa = torch.rand(k // 128, g * m, device=DEV, generator=gen) + 0.5
    sb = torch.rand(g, k // 128, n // 128, device=DEV, generator=gen) + 0.5
    m_indptr = (torch.arange(g + 1, dtype=torch.int32) * m).to(DEV)
    out = torch.empty(g * m, n, device=DEV, dtype=torch.bfloat16)
    med, mn = bench(lambda: flashinfer.group_gemm_fp8_nt_groupwise(a, b, sa, sb, m_indptr, out=out), iters=10, warm=3)
    fl = 2 * g * m * n * k
    print(f"{tag:24s} G={g} M={m} N={n} K={k} med={med:8.3f} ms min={mn:8.3f} ms  {fl/med/1e9:8.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    run(tag="C4")
    run(g=1, m=8192, n=8192, k=8192, tag="square 8k")
    run(g=8, m=512, n=4096, k=7168, tag="moe small m")
    run(g=256, m=128, n=4096, k=7168, tag="256 experts")

"""The reference's grouped-GEMM benchmark grid (benchmarks/bench_groupwise_grouped_gemm_fp8_blackwell.py:58-64: batch 1 /
3 / 8 / 16, m 128..8192, n and k 1024..8192, e5m2 operands, bf16 output, scales = torch.randn) on this library; beside it
the same shapes with power-of-two scales (what the reference's block quantiser produces, flashinfer/testing/utils.py:96).
TFLOP/s = 2 b m n k / median time.  Usage: python tools/bench_gemm_ref_grid.py [--quick]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")

def one(bs, m, n, k):
    torch.random.manual_seed(0)
    a = torch.randn(bs * m, k, device=DEV).to(torch.float8_e5m2)
    b = torch.randn(bs, n, k, device=DEV).to(torch.float8_e5m2)
    out = torch.empty(bs * m, n, device=DEV, dtype=torch.bfloat16)
    sa = torch.randn(k // 128, bs * m, device=DEV)
    sb = torch.randn(bs, k // 128, n // 128, device=DEV)
    seg = torch.arange(0, (bs + 1) * m, m, device=DEV, dtype=torch.int32)
    res = []
    for scales in ((sa, sb), (torch.pow(2.0, torch.randint(-4, 5, sa.shape, device=DEV).float()),
                              torch.pow(2.0, torch.randint(-4, 5, sb.shape, device=DEV).float()))):
        med, _ = bench(lambda: flashinfer.group_gemm_fp8_nt_groupwise(a, b, scales[0], scales[1], seg, out=out), iters=7, warm=2)
        res.append(2 * bs * m * n * k / med / 1e9)
    return res

if __name__ == "__main__":
    quick = "--quick" in sys.argv
    print("batch     m     n     k   TFLOP/s (randn scales, the reference's inputs)   TFLOP/s (power-of-two scales)", flush=True)
    for bs in [1, 3, 8, 16]:
        for m in [128, 512, 1024, 2048, 4096, 8192]:
            for n in [1024, 2048, 4096, 8192]:
                for k in [1024, 2048, 4096, 8192]:
                    if quick and (n != k or m not in (128, 1024, 8192)): continue
                    r = one(bs, m, n, k)
                    print(f"{bs:5d} {m:5d} {n:5d} {k:5d}   {r[0]:8.1f}   {r[1]:8.1f}", flush=True)

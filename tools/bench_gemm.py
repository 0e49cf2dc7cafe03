"""C4: fp8 groupwise grouped GEMM, 8 groups x (M=4096, N=14336, K=4096), block 128.  FLOPs = 2*G*M*N*K
(ref: benchmarks/bench_groupwise_grouped_gemm_fp8_blackwell.py:51)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")

def quantize_blockwise(x, tr, tk):
    """The reference's block quantiser restated for the GPU (flashinfer/testing/utils.py:66-161): per (tr, tk) tile
    scale = amax.clamp(1e-4) / 448 rounded UP to a power of two, values x / (scale + 1e-8) -> e4m3.
    x (g, rows, k) -> (x8, scale (g, k / tk, rows / tr)), the "MN"-major layout."""
    g, rows, k = x.shape
    xt = x.float().reshape(g, rows // tr, tr, k // tk, tk)
    amax = xt.abs().amax(dim=(2, 4)).clamp(1e-4)
    scale = torch.pow(2.0, torch.ceil(torch.log2(amax / 448.0)))
    x8 = (xt / (scale[:, :, None, :, None] + 1e-8)).reshape(g, rows, k).to(torch.float8_e4m3fn)
    return x8, scale.transpose(1, 2).contiguous()


def make_inputs(g, m, n, k, gen, quantised=True):
    """a = randn, b = randn / sqrt(k), quantised with the reference's scheme (power-of-two scales);
    quantised=False: the r1 / r2 bench inputs (randn cast to e4m3, uniform random scales in [0.5, 1.5))."""
    if quantised:
        a8 = torch.empty(g * m, k, device=DEV, dtype=torch.float8_e4m3fn)
        sa = torch.empty(k // 128, g * m, device=DEV)
        for i in range(g):  # group by group: the f32 staging copies stay small
            q, s = quantize_blockwise(torch.randn(1, m, k, device=DEV, generator=gen), 1, 128)
            a8[i * m:(i + 1) * m] = q[0]
            sa[:, i * m:(i + 1) * m] = s[0]
        b8 = torch.empty(g, n, k, device=DEV, dtype=torch.float8_e4m3fn)
        sb = torch.empty(g, k // 128, n // 128, device=DEV)
        for i in range(g):
            q, s = quantize_blockwise(torch.randn(1, n, k, device=DEV, generator=gen) / k ** 0.5, 128, 128)
            b8[i] = q[0]
            sb[i] = s[0]
        return a8, b8, sa, sb
    a = torch.randn(g * m, k, device=DEV, generator=gen).to(torch.float8_e4m3fn)
    b = (torch.randn(g, n, k, device=DEV, generator=gen) / k ** 0.5).to(torch.float8_e4m3fn)
    sa = torch.rand(k // 128, g * m, device=DEV, generator=gen) + 0.5
    sb = torch.rand(g, k // 128, n // 128, device=DEV, generator=gen) + 0.5
    return a, b, sa, sb


def run(g=8, m=4096, n=14336, k=4096, tag="", quantised=True):
    gen = torch.Generator(device=DEV).manual_seed(0)
    a, b, sa, sb = make_inputs(g, m, n, k, gen, quantised)
    tag = tag + (" [ref-quantised]" if quantised else " [rand scales]")
    m_indptr = (torch.arange(g + 1, dtype=torch.int32) * m).to(DEV)
    out = torch.empty(g * m, n, device=DEV, dtype=torch.bfloat16)
    med, mn = bench(lambda: flashinfer.group_gemm_fp8_nt_groupwise(a, b, sa, sb, m_indptr, out=out), iters=10, warm=3)
    fl = 2 * g * m * n * k
    print(f"{tag:40s} G={g} M={m} N={n} K={k} med={med:8.3f} ms min={mn:8.3f} ms  {fl/med/1e9:8.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    run(tag="C4")
    run(g=1, m=8192, n=8192, k=8192, tag="square 8k")
    run(g=8, m=512, n=4096, k=7168, tag="moe small m")
    run(g=256, m=128, n=4096, k=7168, tag="256 experts")

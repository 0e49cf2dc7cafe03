"""GPU parity: fp8 groupwise (grouped) GEMM against dequantise-then-matmul, the reference's own check
(tests/GEMM/test_groupwise_scaled_gemm_fp8.py:35-71, 135-192; atol = rtol = 1e-2 in bf16 output)."""
import math

import pytest
import torch

from oracle import gemm_ref as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("m", [4, 128, 300, 1024])
@pytest.mark.parametrize("n,k", [(128, 128), (256, 512), (1032, 256)])
@pytest.mark.parametrize("mode", ["MN", "K"])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float16])
def test_gemm_fp8_nt_groupwise(m, n, k, mode, out_dtype):
    import flashinfer

    torch.manual_seed(0)
    a = torch.randn(m, k)
    b = torch.randn(n, k) / math.sqrt(k)
    n_pad = -(-n // 128) * 128
    b_pad = torch.zeros(n_pad, k)
    b_pad[:n] = b
    a8, sa = G.quantize_fp8(a, (1, 128), mode)
    b8p, sb = G.quantize_fp8(b_pad, (128, 128), mode)
    b8 = b8p[:n].contiguous()
    out = flashinfer.gemm_fp8_nt_groupwise(a8.to(DEV), b8.to(DEV), sa.to(DEV), sb.to(DEV), scale_major_mode=mode,
                                           out_dtype=out_dtype)
    ref = G.gemm_fp8_nt_groupwise_ref(a8, b8p, sa, sb, mode)[:, :n]
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("mode", ["MN", "K"])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float16])
def test_gemm_fp8_nt_groupwise_large(mode, out_dtype):
    """Enough 256 x 128 tiles (33 x 16 >= 2 x CUs) for the persistent large-problem kernel; random data."""
    import flashinfer

    torch.manual_seed(5)
    m, n, k = 8195, 2048, 640
    a = torch.randn(m, k)
    b = torch.randn(n, k) / math.sqrt(k)
    a8, sa = G.quantize_fp8(a, (1, 128), mode)
    b8, sb = G.quantize_fp8(b, (128, 128), mode)
    out = flashinfer.gemm_fp8_nt_groupwise(a8.to(DEV), b8.to(DEV), sa.to(DEV), sb.to(DEV), scale_major_mode=mode,
                                           out_dtype=out_dtype)
    ref = G.gemm_fp8_nt_groupwise_ref(a8, b8, sa, sb, mode)
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("ms", [[4, 128, 0, 260], [512, 512], [128] * 8, [4]])
@pytest.mark.parametrize("n,k", [(256, 256), (512, 1024)])
@pytest.mark.parametrize("mode", ["MN", "K"])
def test_group_gemm_fp8_nt_groupwise(ms, n, k, mode):
    import flashinfer

    torch.manual_seed(0)
    g = len(ms)
    cum = sum(ms)
    a = torch.randn(cum, k)
    b = torch.randn(g, n, k) / math.sqrt(k)
    a8, sa = G.quantize_fp8(a, (1, 128), mode)
    b8, sb = G.quantize_fp8(b, (1, 128, 128), mode)
    m_indptr = torch.tensor([0] + list(torch.tensor(ms).cumsum(0)), dtype=torch.int32)
    out = flashinfer.group_gemm_fp8_nt_groupwise(a8.to(DEV), b8.to(DEV), sa.to(DEV), sb.to(DEV), m_indptr.to(DEV),
                                                 scale_major_mode=mode)
    ref = G.group_gemm_fp8_nt_groupwise_ref(a8, b8, sa, sb, m_indptr, mode)
    assert out.dtype == torch.bfloat16
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("scales", ["arbitrary", "one_not_pow2", "subnormal_and_huge"])
@pytest.mark.parametrize("mode", ["MN", "K"])
def test_group_gemm_scales_that_are_not_powers_of_two(scales, mode):
    """The reference takes any f32 scale (csrc/group_gemm_fp8_groupwise_sm100.cu:89-124); its own quantiser happens to
    produce powers of two (flashinfer/testing/utils.py:96-98).  The 256 x 256 kernel feeds power-of-two scales to the
    MFMA's hardware block scales and must fall back to the general fold when a single scale is not one -- checked on
    the device, per call.  (tests/test_gemm_variants_gpu.py sends this case through every large-problem kernel.)"""
    import flashinfer

    torch.manual_seed(5)
    ms, n, k = [260, 4, 512], 512, 512
    g, cum = len(ms), sum(ms)
    a8 = torch.randn(cum, k).to(torch.float8_e4m3fn)
    b8 = (torch.randn(g, n, k) / math.sqrt(k)).to(torch.float8_e4m3fn)
    if scales == "arbitrary":
        sa = torch.rand(k // 128, cum) + 0.5
        sb = torch.rand(g, k // 128, n // 128) + 0.5
    elif scales == "one_not_pow2":
        sa = torch.pow(2.0, torch.randint(-3, 4, (k // 128, cum)).float())
        sb = torch.pow(2.0, torch.randint(-3, 4, (g, k // 128, n // 128)).float())
        sb[2, 3, 1] = 0.75
    else:  # powers of two, but outside what an E8M0 byte of a NORMAL f32 holds, or not positive: no hardware path
        sa = torch.pow(2.0, torch.randint(-3, 4, (k // 128, cum)).float())
        sb = torch.pow(2.0, torch.randint(-3, 4, (g, k // 128, n // 128)).float())
        sa[1, 7] = 2.0 ** -130   # subnormal f32
        sa[2, 300] = -2.0        # negative
    if mode == "K":
        sa, sb = sa.t().contiguous(), sb.transpose(1, 2).contiguous()
    m_indptr = torch.tensor([0] + list(torch.tensor(ms).cumsum(0)), dtype=torch.int32)
    out = flashinfer.group_gemm_fp8_nt_groupwise(a8.to(DEV), b8.to(DEV), sa.to(DEV), sb.to(DEV), m_indptr.to(DEV),
                                                 scale_major_mode=mode)
    ref = G.group_gemm_fp8_nt_groupwise_ref(a8, b8, sa, sb, m_indptr, mode)
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=1e-2, rtol=1e-2)


def test_group_gemm_exact_small_integers():
    """Known-answer: small-integer operands and power-of-two scales are exact in fp8 x fp8 -> f32, so the
    kernel must reproduce the integer result bit for bit (catches operand-layout / k-order mistakes that a
    random-data tolerance could hide)."""
    import flashinfer

    torch.manual_seed(1)
    g, m, n, k = 2, 132, 136, 256
    a = torch.randint(-4, 5, (g * m, k)).float()
    b = torch.randint(-4, 5, (g, n, k)).float()
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    sa = torch.pow(2.0, torch.randint(-2, 3, (k // 128, g * m)).float())
    sb = torch.pow(2.0, torch.randint(-2, 3, (g, k // 128, -(-n // 128))).float())
    m_indptr = torch.tensor([0, m, 2 * m], dtype=torch.int32)
    out = flashinfer.group_gemm_fp8_nt_groupwise(a8.to(DEV), b8.to(DEV), sa.to(DEV), sb.to(DEV), m_indptr.to(DEV),
                                                 out_dtype=torch.float16)
    ref = torch.zeros(g * m, n, dtype=torch.float64)
    for gi in range(g):
        for kb in range(k // 128):
            part = a[gi * m:(gi + 1) * m, kb * 128:(kb + 1) * 128].double() @ b[gi, :, kb * 128:(kb + 1) * 128].double().T
            ref[gi * m:(gi + 1) * m] += part * sa[kb, gi * m:(gi + 1) * m, None].double() * \
                sb[gi, kb].double().repeat_interleave(128)[:n][None]
    assert ref.abs().max() < 60000
    torch.testing.assert_close(out.float().cpu(), ref.float().half().float(), atol=0, rtol=0)


def test_gemm_errors():
    import flashinfer

    a = torch.zeros(8, 128, device=DEV).to(torch.float8_e4m3fn)
    b = torch.zeros(16, 256, device=DEV).to(torch.float8_e4m3fn)
    s = torch.ones(1, 8, device=DEV)
    with pytest.raises(ValueError):
        flashinfer.gemm_fp8_nt_groupwise(a, b, s, s, scale_major_mode="MN")
    with pytest.raises(ValueError):
        flashinfer.gemm_fp8_nt_groupwise(a, a, s, s, scale_major_mode="MN", out_dtype=torch.float32)


@pytest.mark.parametrize("ms,n,k", [([1300, 4, 0, 2300, 520], 400, 256), ([2500], 136, 256), ([128] * 37, 264, 256),
                                    # >= 2 x CUs tiles of 256 x 128: the persistent 256 x 128 kernel (ragged groups,
                                    # an empty group, a partial n tile; odd / even k block counts, one k block)
                                    ([2300, 4, 0, 3100, 1000, 777], 2696, 256), ([8000], 2056, 384),
                                    ([255, 257, 6000, 1], 3584, 128),
                                    # more than 64 groups (two passes of the wave-parallel group search), empty
                                    # groups in between
                                    ([(37 * i) % 301 if i % 5 else 0 for i in range(150)], 264, 128)])
def test_group_gemm_exact_many_tiles_banded_order(ms, n, k):
    _exact_many_tiles(ms, n, k, odd_scale=False)


@pytest.mark.parametrize("ms,n,k", [([8000], 2056, 384), ([255, 257, 6000, 1], 3584, 128)])
def test_group_gemm_mid_size_with_one_scale_that_is_no_power_of_two(ms, n, k):
    """Between one and four 256 x 256 tiles per CU the launcher runs the scale check and the hardware-scale 256 x 256
    variant in front of the kernel it would have chosen anyway; ONE scale of 1.5 must send the whole call to the latter
    (the shapes above with all-power-of-two scales take the former).  Still exact: 1.5 x small integers."""
    _exact_many_tiles(ms, n, k, odd_scale=True)


def _exact_many_tiles(ms, n, k, odd_scale):
    """More m tiles than one band and a ragged last band / last n tile: every (group, m tile, n tile) must
    be visited exactly once by the banded tile order of either kernel.  Small integers -> the result is exact."""
    import flashinfer

    torch.manual_seed(3)
    g = len(ms)
    cum = sum(ms)
    a = torch.randint(-3, 4, (cum, k)).float()
    b = torch.randint(-3, 4, (g, n, k)).float()
    sa = torch.pow(2.0, torch.randint(-1, 2, (k // 128, cum)).float())
    sb = torch.pow(2.0, torch.randint(-1, 2, (g, k // 128, -(-n // 128))).float())
    if odd_scale:
        sa[0, cum // 2] = 1.5
    m_indptr = torch.tensor([0] + list(torch.tensor(ms).cumsum(0)), dtype=torch.int32)
    out = flashinfer.group_gemm_fp8_nt_groupwise(a.to(torch.float8_e4m3fn).to(DEV), b.to(torch.float8_e4m3fn).to(DEV),
                                                 sa.to(DEV), sb.to(DEV), m_indptr.to(DEV), out_dtype=torch.float16)
    ref = torch.zeros(cum, n, dtype=torch.float64)
    for gi in range(g):
        lo, hi = int(m_indptr[gi]), int(m_indptr[gi + 1])
        for kb in range(k // 128):
            part = a[lo:hi, kb * 128:(kb + 1) * 128].double() @ b[gi, :, kb * 128:(kb + 1) * 128].double().T
            ref[lo:hi] += part * sa[kb, lo:hi, None].double() * sb[gi, kb].double().repeat_interleave(128)[:n][None]
    assert ref.abs().max() < 60000
    torch.testing.assert_close(out.float().cpu(), ref.float().half().float(), atol=0, rtol=0)


@pytest.mark.parametrize("seed", range(16))
def test_group_gemm_random_shapes_exact(seed):
    """Seeded random (groups, rows per group, n, k, scale layout, output type) with small-integer operands and
    power-of-two scales: the result is exact, whichever kernel / tile shape the size-based dispatch picks."""
    import random

    import flashinfer

    rng = random.Random(1000 + seed)
    torch.manual_seed(1000 + seed)
    g = rng.choice([1, 2, 5, 8, 33, 70])
    big = rng.random() < 0.5
    ms = [rng.choice([0, 1, 7, 64, 128, 129, 255, 256, 300]) * (rng.choice([1, 4, 9]) if big else 1) for _ in range(g)]
    if sum(ms) == 0:
        ms[0] = 5
    n = 8 * rng.randint(1, 380 if big else 40)
    k = 128 * rng.randint(1, 5)
    mode = rng.choice(["MN", "K"])
    out_dtype = rng.choice([torch.float16, torch.bfloat16])
    cum = sum(ms)
    lim = 3 if out_dtype == torch.float16 else 1  # keep every partial sum exactly representable in the output type
    a = torch.randint(-lim, lim + 1, (cum, k)).float()
    b = torch.randint(-1, 2, (g, n, k)).float()
    sa = torch.pow(2.0, torch.randint(-1, 1, (k // 128, cum)).float())
    sb = torch.pow(2.0, torch.randint(-1, 1, (g, k // 128, -(-n // 128))).float())
    m_indptr = torch.tensor([0] + list(torch.tensor(ms).cumsum(0)), dtype=torch.int32)
    sa_dev = sa if mode == "MN" else sa.t().contiguous()
    sb_dev = sb if mode == "MN" else sb.permute(0, 2, 1).contiguous()
    out = flashinfer.group_gemm_fp8_nt_groupwise(a.to(torch.float8_e4m3fn).to(DEV), b.to(torch.float8_e4m3fn).to(DEV),
                                                 sa_dev.to(DEV), sb_dev.to(DEV), m_indptr.to(DEV), scale_major_mode=mode,
                                                 out_dtype=out_dtype)
    ref = torch.zeros(cum, n, dtype=torch.float64)
    for gi in range(g):
        lo, hi = int(m_indptr[gi]), int(m_indptr[gi + 1])
        for kb in range(k // 128):
            part = a[lo:hi, kb * 128:(kb + 1) * 128].double() @ b[gi, :, kb * 128:(kb + 1) * 128].double().T
            ref[lo:hi] += part * sa[kb, lo:hi, None].double() * sb[gi, kb].double().repeat_interleave(128)[:n][None]
    torch.testing.assert_close(out.float().cpu(), ref.float().to(out_dtype).float(), atol=0, rtol=0)


def test_group_gemm_graph_replay_follows_the_scale_kind():
    """The choice between the hardware-scale 256 x 256 variant and the fold kernels is made on the DEVICE from the
    scale values of each launch: a captured call replayed after the scale tensor was overwritten in place (powers of
    two -> one scale of 1.5 -> powers of two) must give the exact result every time."""
    import flashinfer

    torch.manual_seed(5)
    m, n, k = 8000, 2056, 384  # 32 x 9 tiles of 256 x 256: between one and four per CU
    a = torch.randint(-3, 4, (m, k)).float()
    b = torch.randint(-3, 4, (1, n, k)).float()
    sa = torch.pow(2.0, torch.randint(-1, 2, (k // 128, m)).float())
    sb = torch.pow(2.0, torch.randint(-1, 2, (1, k // 128, -(-n // 128))).float())
    m_indptr = torch.tensor([0, m], dtype=torch.int32).to(DEV)
    a8, b8 = a.to(torch.float8_e4m3fn).to(DEV), b.to(torch.float8_e4m3fn).to(DEV)
    sa_d, sb_d = sa.to(DEV), sb.to(DEV)
    out = torch.empty(m, n, device=DEV, dtype=torch.float16)

    def reference(sa_h):
        ref = torch.zeros(m, n, dtype=torch.float64)
        for kb in range(k // 128):
            part = a[:, kb * 128:(kb + 1) * 128].double() @ b[0, :, kb * 128:(kb + 1) * 128].double().T
            ref += part * sa_h[kb, :, None].double() * sb[0, kb].double().repeat_interleave(128)[:n][None]
        return ref.float().half().float()

    flashinfer.group_gemm_fp8_nt_groupwise(a8, b8, sa_d, sb_d, m_indptr, out=out)  # warm-up: allocates the flag ring
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        flashinfer.group_gemm_fp8_nt_groupwise(a8, b8, sa_d, sb_d, m_indptr, out=out)
    for odd in (False, True, False):
        sa_h = sa.clone()
        if odd:
            sa_h[1, 4321] = 1.5
        sa_d.copy_(sa_h)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        torch.testing.assert_close(out.float().cpu(), reference(sa_h), atol=0, rtol=0)

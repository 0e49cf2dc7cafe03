"""Parity at BASELINE.json's FULL sizes for C3 (fp8 causal batch prefill, bs 16 x qo 2048 x kv 8192, GQA 32/8,
d 128, page 16) and C4 (fp8 groupwise grouped GEMM, 8 x 4096 x 14336 x 4096) -- the shapes bench.py times.
The oracle cannot run these sizes, so each test uses (1) size-independent properties over the WHOLE output
(split-KV invariance; exact row / column checksums on integer data) and (2) the oracle on a sample chosen to hit
what only exists at full size: first / last rows of the first / last request, q-tile boundaries, all heads of a
kv head (C3); first / last / interior m tiles of every group and the first / last n tile (C4)."""
import math

import pytest
import torch

from oracle import attention_ref as R
from oracle import gemm_ref as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _per_head_quant_gpu(x):
    """per_head_symmetric_quant (ref: tests/attention/test_hopper_fp8_attention.py:12-41) on device tensors
    [n, H, D]; bit-identical to oracle.attention_ref.per_head_symmetric_quant (checked below on a slice)."""
    amax = x.abs().amax(dim=(0, 2)).to(torch.float32)
    # the scale is formed on the host: torch divides a device tensor by a python scalar as a multiplication by
    # its reciprocal, which is one ulp off the true quotient the reference (and the oracle) compute
    s = torch.clamp(amax.cpu() / 448.0, min=1e-6).to(x.device)
    return torch.clamp(x.float() / s.view(1, -1, 1), min=-448.0, max=448.0).to(torch.float8_e4m3fn), s


def test_full_size_c3_fp8_prefill():
    import flashinfer

    b, qo, kv, hq, hkv, d, ps = 16, 2048, 8192, 32, 8, 128, 16
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = b * kv // ps
    q16 = torch.randn(b * qo, hq, d, device=DEV, dtype=torch.float16, generator=g)
    q8, sq = _per_head_quant_gpu(q16)
    xq_ref, s_ref = R.per_head_symmetric_quant(q16[:64].cpu())  # the device quantiser == the oracle's
    x_dev, s_dev = _per_head_quant_gpu(q16[:64])
    assert torch.equal(xq_ref.view(torch.uint8), x_dev.cpu().view(torch.uint8)) and torch.equal(s_ref, s_dev.cpu())
    del q16
    # K / V rows of all requests, quantised per kv head, then scattered into shuffled pages
    indices = torch.randperm(npages, device=DEV, generator=g).to(torch.int32)
    k16 = torch.randn(b * kv, hkv, d, device=DEV, dtype=torch.float16, generator=g)
    k8, sk = _per_head_quant_gpu(k16)
    del k16
    v16 = torch.randn(b * kv, hkv, d, device=DEV, dtype=torch.float16, generator=g)
    v8, sv = _per_head_quant_gpu(v16)
    del v16
    cache = torch.empty(npages, 2, ps, hkv, d, device=DEV, dtype=torch.float8_e4m3fn)
    cache.view(torch.uint8)[indices.long(), 0] = k8.view(torch.uint8).view(npages, ps, hkv, d)
    cache.view(torch.uint8)[indices.long(), 1] = v8.view(torch.uint8).view(npages, ps, hkv, d)
    qo_indptr = (torch.arange(b + 1, dtype=torch.int32) * qo).to(DEV)
    indptr = (torch.arange(b + 1, dtype=torch.int32) * (kv // ps)).to(DEV)
    last = torch.full((b,), ps, dtype=torch.int32, device=DEV)

    def run(**kw):
        ws = torch.zeros((1280 if kw else 128) << 20, dtype=torch.uint8, device=DEV)
        w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, "NHD")
        w.plan(qo_indptr, indptr, indices, last, hq, hkv, d, ps, causal=True, q_data_type=torch.float8_e4m3fn,
               kv_data_type=torch.float8_e4m3fn, o_data_type=torch.bfloat16, **kw)
        o, lse = w.run(q8, cache, return_lse=True, scale_q=sq, scale_k=sk, scale_v=sv)
        torch.cuda.synchronize()
        return o, lse, w

    o, lse, w = run()  # the plan bench.py times
    assert w._plan_info[14] == 0  # 1024 q tiles per kv head fill the chip: no split
    # (1) split-KV invariance over the whole output: two 4096-token chunks merged == one pass
    o_s, lse_s, w_s = run(fixed_split_size=4096)
    assert w_s._plan_info[14] == 1
    diff = (o.float() - o_s.float()).abs()
    assert diff.max() < 5e-2 and diff.mean() < 1e-3  # e4m3 rounding of P follows the running max: not bit-equal
    torch.testing.assert_close(lse, lse_s, rtol=1e-3, atol=2e-3)
    del o_s, lse_s, diff
    # (2) oracle on sampled rows: a causal row of request r sees keys [0, kv - qo + i]
    samples = [(0, 0), (0, 1), (0, 31), (0, 32), (0, 2047), (15, 0), (15, 1023), (15, 1024), (15, 2047), (7, 777),
               (3, 127), (3, 128)]
    for r, i in samples:
        visible = kv - qo + i + 1
        kk = k8[r * kv: r * kv + visible].cpu()
        vv = v8[r * kv: r * kv + visible].cpu()
        row = r * qo + i
        o_ref, lse_ref = R.fp8_attention_ref(q8[row:row + 1].cpu(), kk, vv, sq.cpu(), sk.cpu(), sv.cpu(), causal=False)
        torch.testing.assert_close(o[row:row + 1].float().cpu(), o_ref.float(), rtol=5e-2, atol=5e-2)  # fp8 bar
        torch.testing.assert_close(lse[row:row + 1].cpu(), lse_ref.float(), rtol=1e-3, atol=2e-3)


def _group_tiles(G_, m, n):
    """(group, first row, first col) of ~64 sampled 128 x 256 output tiles: first / last / an interior m tile of
    every group x first / last / an interior n tile."""
    out = []
    for gi in range(G_):
        for mt in (0, (5 * gi + 3) % (m // 128), m // 128 - 1):
            for nt in (0, (7 * gi + 11) % (n // 256), n // 256 - 1):
                out.append((gi, gi * m + mt * 128, nt * 256))
    return out[::1]


def test_full_size_c4_group_gemm_random_data():
    import flashinfer

    G_, m, n, k = 8, 4096, 14336, 4096
    g = torch.Generator(device=DEV).manual_seed(1)
    a = torch.randn(G_ * m, k, device=DEV, generator=g).to(torch.float8_e4m3fn)
    bm = (torch.randn(G_, n, k, device=DEV, generator=g) / k ** 0.5).to(torch.float8_e4m3fn)
    sa = torch.rand(k // 128, G_ * m, device=DEV, generator=g) + 0.5
    sb = torch.rand(G_, k // 128, n // 128, device=DEV, generator=g) + 0.5
    m_indptr = (torch.arange(G_ + 1, dtype=torch.int32) * m).to(DEV)
    out = flashinfer.group_gemm_fp8_nt_groupwise(a, bm, sa, sb, m_indptr, out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    tiles = _group_tiles(G_, m, n)
    assert len(tiles) == 72
    for gi, r0, c0 in tiles:
        a_t, sa_t = a[r0:r0 + 128].cpu(), sa[:, r0:r0 + 128].cpu()
        b_t, sb_t = bm[gi, c0:c0 + 256].cpu(), sb[gi, :, c0 // 128: c0 // 128 + 2].cpu()
        ref = G.gemm_fp8_nt_groupwise_ref(a_t, b_t, sa_t, sb_t, "MN")  # fp64 dequantise -> matmul
        torch.testing.assert_close(out[r0:r0 + 128, c0:c0 + 256].float().cpu(), ref.float(), rtol=1e-2, atol=1e-2)


def test_full_size_c4_group_gemm_exact_checksums():
    """Integer operands in {-1, 0, 1}, scales in {1, 2}: every partial sum is an exact integer below 2^11, so the
    fp16 output is exact and so are its row and column sums.  Expected sums need only matrix-vector products:
      sum_n D[m, n] = sum_kb sa[kb, m] * (a[m, kb] . w[g, kb]),   w[g, kb][k] = sum_n sb[g, kb, n/128] b[g, n, k]
      sum_{m in g} D[m, n] = sum_kb sb[g, kb, n/128] * (u[g, kb] . b[g, n, kb]),   u[g, kb][k] = sum_m sa[kb, m] a[m, k]
    Any tile the persistent kernel's walk skipped, repeated or misplaced changes a row AND a column sum."""
    import flashinfer

    G_, m, n, k = 8, 4096, 14336, 4096
    kb_n = k // 128
    g = torch.Generator(device=DEV).manual_seed(2)
    a = torch.randint(-1, 2, (G_ * m, k), device=DEV, generator=g).float()
    bm = torch.randint(-1, 2, (G_, n, k), device=DEV, generator=g).float()
    sa = torch.pow(2.0, torch.randint(0, 2, (kb_n, G_ * m), device=DEV, generator=g).float())
    sb = torch.pow(2.0, torch.randint(0, 2, (G_, kb_n, n // 128), device=DEV, generator=g).float())
    m_indptr = (torch.arange(G_ + 1, dtype=torch.int32) * m).to(DEV)
    out = flashinfer.group_gemm_fp8_nt_groupwise(a.to(torch.float8_e4m3fn), bm.to(torch.float8_e4m3fn), sa, sb,
                                                 m_indptr, out_dtype=torch.float16)
    torch.cuda.synchronize()
    assert out.abs().max() < 2048  # exact integers in fp16
    row_sum = out.double().sum(dim=1).cpu()                       # [cum_m]
    col_sum = out.double().view(G_, m, n).sum(dim=1).cpu()        # [G, n]
    a_c, sa_c = a.cpu().double(), sa.cpu().double()
    exp_row = torch.zeros(G_ * m, dtype=torch.float64)
    exp_col = torch.zeros(G_, n, dtype=torch.float64)
    for gi in range(G_):
        b_g = bm[gi].cpu().double()                                # [n, k]
        sb_g = sb[gi].cpu().double()                               # [k/128, n/128]
        rows = slice(gi * m, (gi + 1) * m)
        for kb in range(kb_n):
            ks = slice(kb * 128, (kb + 1) * 128)
            w = (sb_g[kb].repeat_interleave(128)[:, None] * b_g[:, ks]).sum(0)          # [128]
            exp_row[rows] += sa_c[kb, rows] * (a_c[rows, ks] @ w)
            u = (sa_c[kb, rows, None] * a_c[rows, ks]).sum(0)                            # [128]
            exp_col[gi] += sb_g[kb].repeat_interleave(128) * (b_g[:, ks] @ u)
    assert torch.equal(row_sum, exp_row)
    assert torch.equal(col_sum, exp_col)
    # and a few tiles element by element (exact)
    for gi, r0, c0 in _group_tiles(G_, m, n)[::9]:
        ref = torch.zeros(128, 256, dtype=torch.float64)
        for kb in range(kb_n):
            ks = slice(kb * 128, (kb + 1) * 128)
            part = a_c[r0:r0 + 128, ks] @ bm[gi, c0:c0 + 256, ks].cpu().double().T
            ref += part * sa_c[kb, r0:r0 + 128, None] * sb[gi, kb, c0 // 128: c0 // 128 + 2].cpu().double().repeat_interleave(128)[None]
        assert torch.equal(out[r0:r0 + 128, c0:c0 + 256].double().cpu(), ref)

"""CPU restatement of the fp8 groupwise-scaled (grouped) GEMM and of the block quantiser its tests use.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  The GEMM arithmetic of the reference lives in NVIDIA
CUTLASS (un-vendored submodule 3rdparty/cutlass, commit unknown), so parity is anchored on
the reference's own pure-torch check: dequantise, then matmul (tests/GEMM/test_groupwise_scaled_gemm_fp8.py
:135-192, atol = rtol = 1e-2).  quantize_fp8 / dequantize_fp8 restate flashinfer/testing/utils.py:66-216
(that module imports flashinfer and cannot be imported here).
"""
from __future__ import annotations

import torch

FP8_MAX = 448.0  # torch.finfo(torch.float8_e4m3fn).max


def quantize_fp8(x: torch.Tensor, tile_shape, scale_major_mode: str):
    """Per-tile symmetric quantisation to e4m3; the scale is amax.clamp(1e-4) / 448 rounded UP to a power
    of two (ref: flashinfer/testing/utils.py:96-98).  x is 2-D (rows, k) with tile (tr, tk) or 3-D
    (g, rows, k) with tile (1, tr, tk).  Returns (x_fp8, scale) with scale laid out as the reference does:
    "K": (rows/tr, k/tk) [3-D: (g, rows/tr, k/tk)];  "MN": (k/tk, rows/tr) [3-D: (g, k/tk, rows/tr)]."""
    if x.ndim == 2:
        xq, s = quantize_fp8(x[None], (1,) + tuple(tile_shape), scale_major_mode)
        return xq[0], s[0]
    g, rows, k = x.shape
    _, tr, tk = tile_shape
    xt = x.float().reshape(g, rows // tr, tr, k // tk, tk)
    amax = xt.abs().amax(dim=(2, 4)).clamp(1e-4)  # (g, rows/tr, k/tk)
    scale = torch.pow(2.0, torch.ceil(torch.log2(amax / FP8_MAX)))
    xq = (xt / (scale[:, :, None, :, None] + 1e-8)).reshape(g, rows, k).to(torch.float8_e4m3fn)
    if scale_major_mode != "K":
        scale = scale.transpose(1, 2).contiguous()
    return xq, scale


def dequantize_fp8(xq: torch.Tensor, scale: torch.Tensor, scale_major_mode: str, dtype=torch.float32):
    """Inverse layout of quantize_fp8 (ref: flashinfer/testing/utils.py:164-216)."""
    if xq.ndim == 2:
        return dequantize_fp8(xq[None], scale[None], scale_major_mode, dtype)[0]
    g, rows, k = xq.shape
    s = scale if scale_major_mode == "K" else scale.transpose(1, 2)  # (g, rows/tr, k/tk)
    tr, tk = rows // s.shape[1], k // s.shape[2]
    xt = xq.to(torch.float32).to(dtype).reshape(g, s.shape[1], tr, s.shape[2], tk)
    return (xt * s.to(dtype)[:, :, None, :, None]).reshape(g, rows, k)


def group_gemm_fp8_nt_groupwise_ref(a8, b8, a_scale, b_scale, m_indptr, scale_major_mode="MN",
                                    dtype=torch.float64):
    """out[m_indptr[g]:m_indptr[g+1]] = dequant(A rows of group g) @ dequant(B[g])^T.
    a8 (cum_m, k), b8 (G, n, k); a_scale granularity (1, 128) or (128, 128) inferred from its shape.
    ref: tests/GEMM/test_groupwise_scaled_gemm_fp8.py:170-192 (dequant -> einsum)."""
    a_d = dequantize_fp8(a8, a_scale, scale_major_mode, dtype)
    b_d = dequantize_fp8(b8, b_scale, scale_major_mode, dtype)
    out = torch.zeros(a8.shape[0], b8.shape[1], dtype=dtype)
    for g in range(b8.shape[0]):
        lo, hi = int(m_indptr[g]), int(m_indptr[g + 1])
        out[lo:hi] = a_d[lo:hi] @ b_d[g].transpose(0, 1)
    return out


def gemm_fp8_nt_groupwise_ref(a8, b8, a_scale, b_scale, scale_major_mode="MN", dtype=torch.float64):
    """ref: tests/GEMM/test_groupwise_scaled_gemm_fp8.py:35-71."""
    a_d = dequantize_fp8(a8, a_scale, scale_major_mode, dtype)
    b_d = dequantize_fp8(b8, b_scale, scale_major_mode, dtype)
    return a_d @ b_d.transpose(0, 1)

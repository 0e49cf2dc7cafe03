"""fp8 groupwise-scaled GEMM operators (``gemm_fp8_nt_groupwise``, ``group_gemm_fp8_nt_groupwise``).

API of the reference's ``flashinfer/gemm.py`` (:2321-2484, :2657-2811); the kernel is csrc/gemm.hip
(MFMA fp8, two-level accumulation per 128-wide K block).  NVIDIA-only knobs (``mma_sm``, ``backend``)
are accepted and ignored.
"""
from __future__ import annotations

from typing import Literal, Optional, Tuple

import torch

from . import _lib

_FP8 = (torch.float8_e4m3fn, torch.float8_e5m2)


def _validate_fp8_output_dtype(dtype: torch.dtype) -> None:
    if dtype not in (torch.bfloat16, torch.float16):
        raise ValueError(f"Unsupported output dtype: {dtype}. Only torch.bfloat16 and torch.float16 are supported.")


def gemm_fp8_nt_groupwise(
    a: torch.Tensor,
    b: torch.Tensor,
    a_scale: torch.Tensor,
    b_scale: torch.Tensor,
    scale_major_mode: Optional[Literal["MN", "K"]] = None,
    mma_sm: int = 1,
    scale_granularity_mnk: Tuple[int, int, int] = (1, 128, 128),
    out: Optional[torch.Tensor] = None,
    out_dtype: Optional[torch.dtype] = None,
    backend: Literal["cutlass", "trtllm"] = "cutlass",
) -> torch.Tensor:
    r"""``out = (a * a_scale) @ (b * b_scale)^T`` with fp8 inputs and groupwise scales.

    a : ``(m, k)`` fp8 row-major; b : ``(n, k)`` fp8.
    a_scale : ``(m // gm, k // 128)`` if ``scale_major_mode == "K"`` else ``(k // 128, m // gm)``.
    b_scale : ``(n // 128, k // 128)`` if ``"K"`` else ``(k // 128, n // 128)``.
    scale_granularity_mnk : ``(1, 128, 128)`` or ``(128, 128, 128)``.
    out / out_dtype : ``(m, n)`` bf16 (default) or fp16.
    """
    for t, name in ((a, "a"), (b, "b"), (a_scale, "a_scale"), (b_scale, "b_scale")):
        _lib.require_gpu_tensor(t, name)
    if a.ndim != 2 or b.ndim != 2:
        raise ValueError(f"Shape mismatch. a.shape = {a.shape}, b.shape = {b.shape}")
    if a.shape[1] != b.shape[1]:
        raise ValueError(f"Shape mismatch. a.shape[1] = {a.shape[1]}, b.shape[1] = {b.shape[1]}")
    if a.dtype not in _FP8 or b.dtype not in _FP8:
        raise ValueError("a and b must be float8_e4m3fn or float8_e5m2")
    if scale_major_mode is None:
        scale_major_mode = "MN"
    if scale_major_mode not in ("MN", "K"):
        raise ValueError(f"Invalid scale_major_mode {scale_major_mode}")
    if out is None:
        out_dtype = out_dtype or torch.bfloat16
    else:
        out_dtype = out.dtype
    _validate_fp8_output_dtype(out_dtype)
    m, k = a.shape
    n = b.shape[0]
    if out is None:
        out = torch.empty(m, n, device=a.device, dtype=out_dtype)
    elif out.shape != (m, n) or not out.is_contiguous():
        raise ValueError("out must be a contiguous (m, n) tensor")
    a, b = a.contiguous(), b.contiguous()
    a_scale = a_scale.to(torch.float32).contiguous()
    b_scale = b_scale.to(torch.float32).contiguous()
    gm, gn, gk = scale_granularity_mnk
    with torch.cuda.device(a.device):
        _lib.check(
            _lib.lib().fi_gemm_fp8_nt_groupwise(
                a.data_ptr(), b.data_ptr(), a_scale.data_ptr(), b_scale.data_ptr(), out.data_ptr(), m, n, k,
                gm, gn, gk, int(scale_major_mode == "K"), _lib.fi_dtype(a.dtype), _lib.fi_dtype(b.dtype),
                _lib.fi_dtype(out_dtype), _lib.current_stream(a.device),
            ),
            "gemm_fp8_nt_groupwise",
        )
    return out


def group_gemm_fp8_nt_groupwise(
    a: torch.Tensor,  # (cum_m, k)
    b: torch.Tensor,  # (batch_size, n, k)
    a_scale: torch.Tensor,  # (k // block_size, cum_m)
    b_scale: torch.Tensor,  # (batch_size, k // block_size, n // block_size)
    m_indptr: torch.Tensor,  # (batch_size + 1, )
    scale_granularity_mnk: Tuple[int, int, int] = (1, 128, 128),
    scale_major_mode: Literal["MN", "K"] = "MN",
    mma_sm: int = 1,
    out: Optional[torch.Tensor] = None,  # (cum_m, n)
    out_dtype: Optional[torch.dtype] = None,
) -> torch.Tensor:
    r"""Grouped GEMM with fp8 inputs and groupwise scales: rows ``m_indptr[g]:m_indptr[g+1]`` of ``a``
    are multiplied with ``b[g]^T``.

    a : ``(cum_m, k)`` fp8; b : ``(batch_size, n, k)`` fp8.
    a_scale : ``(cum_m, k // 128)`` if ``"K"`` else ``(k // 128, cum_m)`` (float32).
    b_scale : ``(batch_size, n // 128, k // 128)`` if ``"K"`` else ``(batch_size, k // 128, n // 128)``.
    m_indptr : ``(batch_size + 1,)`` int32, each entry a multiple of 4.
    out : ``(cum_m, n)`` bf16 (default) / fp16.
    """
    for t, name in ((a, "a"), (b, "b"), (a_scale, "a_scale"), (b_scale, "b_scale"), (m_indptr, "m_indptr")):
        _lib.require_gpu_tensor(t, name)
    assert a.dtype in _FP8
    assert b.dtype in _FP8
    assert a_scale.dtype == torch.float32
    assert b_scale.dtype == torch.float32
    assert m_indptr.dtype == torch.int32
    assert scale_major_mode in ["MN", "K"]
    assert mma_sm in [1, 2]
    if out is None:
        if out_dtype is None:
            out_dtype = torch.bfloat16
    else:
        if out_dtype is None:
            out_dtype = out.dtype
    _validate_fp8_output_dtype(out_dtype)
    num_groups = m_indptr.shape[0] - 1
    assert b.shape[0] == num_groups
    n = b.shape[1]
    k = b.shape[2]
    assert a.shape[1] == k
    assert n % 8 == 0
    assert k % 16 == 0
    out_shape = (a.shape[0], n)
    if out is None:
        out = torch.empty(out_shape, dtype=out_dtype, device=a.device)
    else:
        assert out.shape == out_shape
        assert out.dtype == out_dtype
        assert out.is_contiguous()
    a, b = a.contiguous(), b.contiguous()
    a_scale, b_scale, m_indptr = a_scale.contiguous(), b_scale.contiguous(), m_indptr.contiguous()
    gm, gn, gk = scale_granularity_mnk
    with torch.cuda.device(a.device):
        _lib.check(
            _lib.lib().fi_group_gemm_fp8_nt_groupwise(
                a.data_ptr(), b.data_ptr(), a_scale.data_ptr(), b_scale.data_ptr(), out.data_ptr(),
                m_indptr.data_ptr(), num_groups, a.shape[0], n, k, gm, gn, gk, int(scale_major_mode == "K"),
                _lib.fi_dtype(a.dtype), _lib.fi_dtype(b.dtype), _lib.fi_dtype(out_dtype),
                _lib.current_stream(a.device),
            ),
            "group_gemm_fp8_nt_groupwise",
        )
    return out

"""Enums and small host helpers shared by the operator modules.

Mirrors the part of the reference's ``flashinfer/utils.py`` that the attention / GEMM path uses
(enums :30-46, cache unpacking :149-169, module-level cache buffers :183-207, dtype checks :239-251,
``check_shape_dtype_device`` :493-515).  Host logic only -- no kernels live here.
"""
from __future__ import annotations

import math
from enum import Enum
from typing import Dict, Optional, Sequence, Tuple, Union

import torch


class PosEncodingMode(Enum):
    NONE = 0
    ROPE_LLAMA = 1
    ALIBI = 2


class MaskMode(Enum):
    NON_CAUSAL = 0
    CAUSAL = 1
    CUSTOM = 2
    MULTIITEMSCORING = 3


class TensorLayout(Enum):
    NHD = 0
    HND = 1


log2e = 1.44269504088896340736


def _check_pos_encoding_mode(pos_encoding_mode: str) -> None:
    if not hasattr(PosEncodingMode, pos_encoding_mode):
        raise KeyError("Invalid pos_encoding_mode {}".format(pos_encoding_mode))


def _check_kv_layout(kv_layout: str) -> None:
    if not hasattr(TensorLayout, kv_layout):
        raise KeyError("Invalid kv_layout {}".format(kv_layout))


def is_float8(x: torch.Tensor) -> bool:
    return x.dtype in (torch.float8_e4m3fn, torch.float8_e5m2)


def canonicalize_torch_dtype(dtype: Union[torch.dtype, str]) -> torch.dtype:
    if isinstance(dtype, str):
        return getattr(torch, dtype)
    if isinstance(dtype, torch.dtype):
        return dtype
    raise TypeError("dtype must be a string or torch.dtype, got {}".format(type(dtype)))


def get_indptr(x: torch.Tensor) -> torch.Tensor:
    x = x.to(torch.int64)
    out = torch.zeros(x.shape[0] + 1, dtype=x.dtype, device=x.device)
    out[1:] = x.cumsum(0)
    return out


def _add_page_dim(x: torch.Tensor, kv_layout: str, full_ndim: int) -> torch.Tensor:
    """A cache given without the page_size axis has page_size == 1 (ref: utils.py:63-117)."""
    if x.ndim == full_ndim:
        return x
    if x.ndim != full_ndim - 1:
        raise ValueError(f"x must be {full_ndim - 1}D or {full_ndim}D")
    if kv_layout == "NHD":
        return x.unsqueeze(-3)
    if kv_layout == "HND":
        return x.unsqueeze(-2)
    raise KeyError("Invalid kv_layout {}".format(kv_layout))


def _unpack_paged_kv_cache(
    paged_kv_cache: Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]], kv_layout: str
) -> Tuple[torch.Tensor, torch.Tensor]:
    """(k_cache, v_cache) 4-D views of a 5-D cache, a 4-D page_size-1 cache, or a (K, V) tuple."""
    if isinstance(paged_kv_cache, tuple):
        k_cache, v_cache = paged_kv_cache
        return _add_page_dim(k_cache, kv_layout, 4), _add_page_dim(v_cache, kv_layout, 4)
    if torch.is_tensor(paged_kv_cache):
        k_cache, v_cache = _add_page_dim(paged_kv_cache, kv_layout, 5).unbind(dim=1)
        return k_cache, v_cache
    raise KeyError(
        "Unrecognized paged_kv_cache type {}, expect a single tensor or a tuple of tensor.".format(
            type(paged_kv_cache)
        )
    )


def paged_kv_strides(k_cache: torch.Tensor, v_cache: torch.Tensor, kv_layout: str):
    """(page_size, num_kv_heads, head_dim, stride_page, stride_n, stride_h) of a 4-D cache view.
    K and V must share strides (ref: csrc/batch_decode.cu:118-129)."""
    if k_cache.shape != v_cache.shape or k_cache.stride() != v_cache.stride():
        raise ValueError("k_cache and v_cache must have the same shape and strides")
    if k_cache.stride(-1) != 1:
        raise ValueError("the last dimension of the kv cache must be contiguous")
    if kv_layout == "NHD":
        _, page_size, num_kv_heads, head_dim = k_cache.shape
        stride_n, stride_h = k_cache.stride(1), k_cache.stride(2)
    else:
        _, num_kv_heads, page_size, head_dim = k_cache.shape
        stride_h, stride_n = k_cache.stride(1), k_cache.stride(2)
    return page_size, num_kv_heads, head_dim, k_cache.stride(0), stride_n, stride_h


def get_alibi_slopes(n_heads: int) -> torch.Tensor:
    n = 2 ** math.floor(math.log2(n_heads))
    m_0 = 2.0 ** (-8.0 / n)
    m = torch.pow(m_0, torch.arange(1, 1 + n))
    if n < n_heads:
        m_hat_0 = 2.0 ** (-4.0 / n)
        m_hat = torch.pow(m_hat_0, torch.arange(1, 1 + 2 * (n_heads - n), 2))
        m = torch.cat([m, m_hat])
    return m.float()


_cache_buf: Dict[Tuple[str, torch.device], torch.Tensor] = {}


def _get_cache_buf(name: str, nbytes: int, device: torch.device) -> torch.Tensor:
    key = (name, device)
    buf = _cache_buf.get(key)
    if buf is None:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _cache_buf[key] = buf
    return buf


def _ceil_pow2(x: int) -> int:
    return 1 << (x - 1).bit_length()


def _get_range_buf(seq_len: int, device: torch.device) -> torch.Tensor:
    n = _ceil_pow2(max(seq_len, 1))
    key = (f"range_{n}", torch.device(device))
    buf = _cache_buf.get(key)
    if buf is None:
        buf = torch.arange(n, device=device, dtype=torch.int32)
        _cache_buf[key] = buf
    return buf[:seq_len]


def _get_cache_alibi_slopes_buf(num_qo_heads: int, device: torch.device) -> torch.Tensor:
    key = (f"alibi_slopes_{num_qo_heads}", device)
    buf = _cache_buf.get(key)
    if buf is None:
        buf = get_alibi_slopes(num_qo_heads).to(device)
        _cache_buf[key] = buf
    return buf


def _check_cached_qkv_data_type(
    q: torch.Tensor, k: torch.Tensor, dtype_q: torch.dtype, dtype_kv: torch.dtype
) -> None:
    if q.dtype != dtype_q:
        raise ValueError(
            f"The dtype of q {q.dtype} does not match the q_data_type {dtype_q} specified in plan function."
        )
    if k.dtype != dtype_kv:
        raise ValueError(
            f"The dtype of k {k.dtype} does not match the kv_data_type {dtype_kv} specified in plan function."
        )


def check_shape_dtype_device(
    x: torch.Tensor,
    expected_shape: Optional[Sequence[int]],
    expected_dtype: Optional[torch.dtype],
    expected_device: Optional[torch.device],
    name: str,
) -> None:
    if expected_shape and x.shape != torch.Size(expected_shape):
        raise ValueError(f"Invalid shape of {name}: expected {expected_shape}, got {x.shape}")
    if expected_dtype and x.dtype != expected_dtype:
        raise ValueError(f"Invalid dtype of {name}: expected {expected_dtype}, got {x.dtype}")
    if expected_device and x.device != expected_device:
        raise ValueError(f"Invalid device of {name}: expected {expected_device}, got {x.device}")


def device_support_pdl(device: torch.device) -> bool:
    """Programmatic dependent launch is an NVIDIA sm90+ feature; accepted and ignored here."""
    return False


def ceil_div(x: int, y: int) -> int:
    return (x + y - 1) // y


def next_positive_power_of_2(x: int) -> int:
    if x < 1:
        return 1
    return 1 << (x - 1).bit_length()

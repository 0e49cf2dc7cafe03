"""Standalone rotary-embedding operators (ref: flashinfer/rope.py:321-1150).

Same function names, arguments and defaults as the reference; one HIP kernel (csrc/rope.hip) serves every
form: (indptr, offsets) or explicit ``pos_ids``; plain or llama-3.1 frequency scaling; computed angles or a
cos/sin cache; interleaved (GPT-J) or half-split (NeoX) pairs; in place or out of place.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import torch

from . import _lib


def _run(q, k, q_out, k_out, pos_ids, rotary_dim, interleave, rope_scale, rope_theta, smooth_a=0.0,
         smooth_b=0.0, cos_sin_cache=None) -> None:
    for t, name in ((q, "q"), (k, "k")):
        _lib.require_gpu_tensor(t, name)
    if q.dim() != 3 or k.dim() != 3 or q.shape[0] != k.shape[0] or q.shape[2] != k.shape[2]:
        raise ValueError("q and k must be [nnz, num_heads, head_dim] with equal nnz and head_dim")
    if q.dtype != k.dtype or q.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError("q and k must both be float16 or bfloat16")
    if q.stride(-1) != 1 or k.stride(-1) != 1 or q_out.stride(-1) != 1 or k_out.stride(-1) != 1:
        raise ValueError("q/k must be contiguous in head_dim")
    head_dim = q.shape[2]
    if rotary_dim is None:
        rotary_dim = head_dim
    pos_ids = pos_ids.to(device=q.device, dtype=torch.int32).contiguous()
    if pos_ids.numel() != q.shape[0]:
        raise ValueError("pos_ids must have nnz entries")
    params = _lib.RopeParams(
        q=q.data_ptr(), k=k.data_ptr(), q_out=q_out.data_ptr(), k_out=k_out.data_ptr(), pos_ids=pos_ids.data_ptr(),
        cos_sin_cache=_lib.ptr(cos_sin_cache), q_stride_n=q.stride(0), q_stride_h=q.stride(1),
        k_stride_n=k.stride(0), k_stride_h=k.stride(1), qo_stride_n=q_out.stride(0), qo_stride_h=q_out.stride(1),
        ko_stride_n=k_out.stride(0), ko_stride_h=k_out.stride(1), nnz=q.shape[0], num_q_heads=q.shape[1],
        num_k_heads=k.shape[1], head_dim=head_dim, rotary_dim=rotary_dim, interleave=int(interleave),
        dtype=_lib.fi_dtype(q.dtype), rope_rcp_scale=1.0 / rope_scale, rope_rcp_theta=1.0 / rope_theta,
        smooth_a=smooth_a, smooth_b=smooth_b,
    )
    with torch.cuda.device(q.device):
        _lib.check(_lib.lib().fi_apply_rope_pos_ids(C.byref(params), _lib.current_stream(q.device)), "apply_rope")


def _positions(indptr: torch.Tensor, offsets: torch.Tensor, nnz: int) -> torch.Tensor:
    _lib.require_gpu_tensor(indptr, "indptr")
    indptr = indptr.to(torch.int32).contiguous()
    offsets = offsets.to(device=indptr.device, dtype=torch.int32).contiguous()
    pos = torch.empty(nnz, dtype=torch.int32, device=indptr.device)
    with torch.cuda.device(indptr.device):
        _lib.check(
            _lib.lib().fi_rope_positions_from_indptr(indptr.data_ptr(), offsets.data_ptr(), indptr.numel() - 1, nnz,
                                                     pos.data_ptr(), _lib.current_stream(indptr.device)),
            "rope positions",
        )
    return pos


def _llama31_smooth(low_freq_factor: float, high_freq_factor: float, old_context_len: int):
    # ref: include/flashinfer/pos_enc.cuh:976-977
    smooth_a = old_context_len / (2 * math.pi * high_freq_factor - 2 * math.pi * low_freq_factor)
    smooth_b = -1.0 / (high_freq_factor / low_freq_factor - 1.0)
    return smooth_a, smooth_b


def apply_rope_inplace(q, k, indptr, offsets, rotary_dim: Optional[int] = None, interleave: bool = False,
                       rope_scale: float = 1, rope_theta: float = 1e4) -> None:
    r"""Apply RoPE in place to ragged ``q`` / ``k`` (``[nnz, heads, head_dim]``); request ``i`` owns rows
    ``indptr[i]:indptr[i+1]`` and its first row sits at position ``offsets[i]``."""
    _run(q, k, q, k, _positions(indptr, offsets, q.shape[0]), rotary_dim, interleave, rope_scale, rope_theta)


def apply_rope_pos_ids_inplace(q, k, pos_ids, rotary_dim: Optional[int] = None, interleave: bool = False,
                               rope_scale: float = 1, rope_theta: float = 1e4) -> None:
    r"""Apply RoPE in place with explicit per-row positions ``pos_ids`` (``[nnz]``)."""
    _run(q, k, q, k, pos_ids, rotary_dim, interleave, rope_scale, rope_theta)


def apply_llama31_rope_inplace(q, k, indptr, offsets, rotary_dim: Optional[int] = None, interleave: bool = False,
                               rope_scale: float = 8, rope_theta: float = 5e5, low_freq_factor: float = 1,
                               high_freq_factor: float = 4, old_context_len: int = 8192) -> None:
    r"""Llama-3.1 style RoPE (frequency-dependent scaling) in place, (indptr, offsets) form."""
    a, b = _llama31_smooth(low_freq_factor, high_freq_factor, old_context_len)
    _run(q, k, q, k, _positions(indptr, offsets, q.shape[0]), rotary_dim, interleave, rope_scale, rope_theta, a, b)


def apply_llama31_rope_pos_ids_inplace(q, k, pos_ids, rotary_dim: Optional[int] = None, interleave: bool = False,
                                       rope_scale: float = 8, rope_theta: float = 5e5, low_freq_factor: float = 1,
                                       high_freq_factor: float = 4, old_context_len: int = 8192) -> None:
    r"""Llama-3.1 style RoPE in place with explicit positions."""
    a, b = _llama31_smooth(low_freq_factor, high_freq_factor, old_context_len)
    _run(q, k, q, k, pos_ids, rotary_dim, interleave, rope_scale, rope_theta, a, b)


def apply_rope(q, k, indptr, offsets, rotary_dim: Optional[int] = None, interleave: bool = False,
               rope_scale: float = 1, rope_theta: float = 1e4) -> Tuple[torch.Tensor, torch.Tensor]:
    r"""Out-of-place :func:`apply_rope_inplace`; returns ``(q_rope, k_rope)``."""
    q_rope, k_rope = torch.empty_like(q), torch.empty_like(k)
    _run(q, k, q_rope, k_rope, _positions(indptr, offsets, q.shape[0]), rotary_dim, interleave, rope_scale, rope_theta)
    return q_rope, k_rope


def apply_rope_pos_ids(q, k, pos_ids, rotary_dim: Optional[int] = None, interleave: bool = False,
                       rope_scale: float = 1, rope_theta: float = 1e4) -> Tuple[torch.Tensor, torch.Tensor]:
    r"""Out-of-place :func:`apply_rope_pos_ids_inplace`."""
    q_rope, k_rope = torch.empty_like(q), torch.empty_like(k)
    _run(q, k, q_rope, k_rope, pos_ids, rotary_dim, interleave, rope_scale, rope_theta)
    return q_rope, k_rope


def apply_llama31_rope(q, k, indptr, offsets, rotary_dim: Optional[int] = None, interleave: bool = False,
                       rope_scale: float = 8, rope_theta: float = 5e5, low_freq_factor: float = 1,
                       high_freq_factor: float = 4, old_context_len: int = 8192):
    r"""Out-of-place :func:`apply_llama31_rope_inplace`."""
    a, b = _llama31_smooth(low_freq_factor, high_freq_factor, old_context_len)
    q_rope, k_rope = torch.empty_like(q), torch.empty_like(k)
    _run(q, k, q_rope, k_rope, _positions(indptr, offsets, q.shape[0]), rotary_dim, interleave, rope_scale,
         rope_theta, a, b)
    return q_rope, k_rope


def apply_llama31_rope_pos_ids(q, k, pos_ids, rotary_dim: Optional[int] = None, interleave: bool = False,
                               rope_scale: float = 8, rope_theta: float = 5e5, low_freq_factor: float = 1,
                               high_freq_factor: float = 4, old_context_len: int = 8192):
    r"""Out-of-place :func:`apply_llama31_rope_pos_ids_inplace`."""
    a, b = _llama31_smooth(low_freq_factor, high_freq_factor, old_context_len)
    q_rope, k_rope = torch.empty_like(q), torch.empty_like(k)
    _run(q, k, q_rope, k_rope, pos_ids, rotary_dim, interleave, rope_scale, rope_theta, a, b)
    return q_rope, k_rope


def apply_rope_with_cos_sin_cache(positions, query, key, head_size: int, cos_sin_cache, is_neox: bool = True):
    r"""RoPE with a precomputed cache (SGL / vLLM compatible).  ``query`` / ``key``:
    ``(nnz, num_heads * head_size)``; ``cos_sin_cache``: ``(max_seq_len, rotary_dim)`` float32, cosines in the
    first half and sines in the second; ``is_neox``: half-split pairs (True) or interleaved pairs (False)."""
    if cos_sin_cache.dtype != torch.float32:
        raise ValueError("cos_sin_cache should be float32")
    query_out, key_out = torch.empty_like(query), torch.empty_like(key)
    _run(query.view(query.shape[0], -1, head_size), key.view(key.shape[0], -1, head_size),
         query_out.view(query_out.shape[0], -1, head_size), key_out.view(key_out.shape[0], -1, head_size),
         positions, cos_sin_cache.shape[1], not is_neox, 1.0, 1e4, cos_sin_cache=cos_sin_cache.contiguous())
    return query_out, key_out


def apply_rope_with_cos_sin_cache_inplace(positions, query, key, head_size: int, cos_sin_cache,
                                          is_neox: bool = True) -> None:
    r"""In-place :func:`apply_rope_with_cos_sin_cache`."""
    if cos_sin_cache.dtype != torch.float32:
        raise ValueError("cos_sin_cache should be float32")
    qv, kv = query.view(query.shape[0], -1, head_size), key.view(key.shape[0], -1, head_size)
    _run(qv, kv, qv, kv, positions, cos_sin_cache.shape[1], not is_neox, 1.0, 1e4,
         cos_sin_cache=cos_sin_cache.contiguous())

"""Page-table utilities (ref: flashinfer/page.py): sequence lengths, ragged -> COO expansion, and the
K/V append scatter that builds the caches the attention path reads."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple, Union

import torch

from . import _lib
from .utils import TensorLayout, _check_kv_layout, _unpack_paged_kv_cache, paged_kv_strides


def get_seq_lens(
    kv_indptr: torch.Tensor, kv_last_page_len: torch.Tensor, page_size: int
) -> torch.Tensor:
    r"""KV length of every request of a paged cache (ref: flashinfer/page.py:224-247).

    ``kv_len = max(num_pages - 1, 0) * page_size + last_page_len``

    Parameters
    ----------
    kv_indptr : torch.Tensor
        The indptr of the paged kv-cache, shape: ``[batch_size + 1]``.
    kv_last_page_len : torch.Tensor
        Entries in the last page of each request, shape: ``[batch_size]``.
    page_size : int
        The size of a page in the paged kv-cache.
    """
    return (
        torch.clamp(kv_indptr[1:] - kv_indptr[:-1] - 1, min=0) * page_size + kv_last_page_len
    )


def get_batch_indices_positions(
    append_indptr: torch.Tensor, seq_lens: torch.Tensor, nnz: int
) -> Tuple[torch.Tensor, torch.Tensor]:
    r"""Convert append indptr and sequence lengths to per-token batch indices and positions.

    >>> append_indptr = torch.tensor([0, 1, 3, 6, 10], dtype=torch.int32, device="cuda:0")
    >>> seq_lens = torch.tensor([5, 5, 5, 5])
    >>> batch_indices, positions = flashinfer.get_batch_indices_positions(append_indptr, seq_lens, 10)
    >>> batch_indices
    tensor([0, 1, 1, 2, 2, 2, 3, 3, 3, 3], device='cuda:0', dtype=torch.int32)
    >>> positions
    tensor([4, 3, 4, 2, 3, 4, 1, 2, 3, 4], device='cuda:0', dtype=torch.int32)

    (ref: flashinfer/page.py:169-221)
    """
    _lib.require_gpu_tensor(append_indptr, "append_indptr")
    batch_size = append_indptr.size(0) - 1
    append_indptr = append_indptr.to(torch.int32).contiguous()
    seq_lens = seq_lens.to(device=append_indptr.device, dtype=torch.int32).contiguous()
    batch_indices = torch.empty((nnz,), device=append_indptr.device, dtype=torch.int32)
    positions = torch.empty((nnz,), device=append_indptr.device, dtype=torch.int32)
    with torch.cuda.device(append_indptr.device):
        _lib.check(
            _lib.lib().fi_get_batch_indices_positions(
                append_indptr.data_ptr(), seq_lens.data_ptr(), batch_size, nnz, batch_indices.data_ptr(),
                positions.data_ptr(), _lib.current_stream(append_indptr.device),
            ),
            "get_batch_indices_positions",
        )
    return batch_indices, positions


def append_paged_kv_cache(
    append_key: torch.Tensor,
    append_value: torch.Tensor,
    batch_indices: torch.Tensor,
    positions: torch.Tensor,
    paged_kv_cache: Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]],
    kv_indices: torch.Tensor,
    kv_indptr: torch.Tensor,
    kv_last_page_len: torch.Tensor,
    kv_layout: str = "NHD",
) -> None:
    r"""Append a batch of key/value rows to a paged KV cache (in place).

    append_key / append_value : ``[nnz, num_kv_heads, head_dim]``
    batch_indices / positions : ``[nnz]`` int32 (see :func:`get_batch_indices_positions`)
    paged_kv_cache : 5-D tensor or ``(k_cache, v_cache)`` tuple, layout per ``kv_layout``
    kv_indices / kv_indptr / kv_last_page_len : the page table AFTER the append (pages must be allocated).
    (ref: flashinfer/page.py:299-425)
    """
    _check_kv_layout(kv_layout)
    for t, name in ((append_key, "append_key"), (append_value, "append_value")):
        _lib.require_gpu_tensor(t, name)
    k_cache, v_cache = _unpack_paged_kv_cache(paged_kv_cache, kv_layout)
    if append_key.dtype != k_cache.dtype or append_value.dtype != v_cache.dtype:
        raise ValueError("append_key/append_value dtype must match the cache dtype")
    if append_key.dim() != 3 or append_key.shape != append_value.shape:
        raise ValueError("append_key and append_value must be [nnz, num_kv_heads, head_dim]")
    page_size, num_kv_heads, head_dim, stride_page, stride_n, stride_h = paged_kv_strides(k_cache, v_cache, kv_layout)
    if append_key.shape[1] != num_kv_heads or append_key.shape[2] != head_dim:
        raise ValueError("append_key shape does not match the cache")
    if append_key.stride(-1) != 1:
        append_key = append_key.contiguous()
    if append_value.stride(-1) != 1:
        append_value = append_value.contiguous()
    dev = k_cache.device
    batch_indices = batch_indices.to(device=dev, dtype=torch.int32).contiguous()
    positions = positions.to(device=dev, dtype=torch.int32).contiguous()
    kv_indices = kv_indices.to(device=dev, dtype=torch.int32).contiguous()
    kv_indptr = kv_indptr.to(device=dev, dtype=torch.int32).contiguous()
    nnz = append_key.shape[0]
    if batch_indices.numel() != nnz or positions.numel() != nnz:
        raise ValueError("batch_indices and positions must have nnz entries")
    kv = _lib.PagedKV(
        k_data=k_cache.data_ptr(), v_data=v_cache.data_ptr(), indptr=kv_indptr.data_ptr(),
        indices=kv_indices.data_ptr(), last_page_len=None, rope_pos_offset=None, stride_page=stride_page,
        stride_n=stride_n, stride_h=stride_h, page_size=page_size, num_kv_heads=num_kv_heads,
        head_dim=head_dim, batch_size=kv_indptr.numel() - 1, dtype=_lib.fi_dtype(k_cache.dtype),
    )
    with torch.cuda.device(dev):
        _lib.check(
            _lib.lib().fi_append_paged_kv_cache(
                append_key.data_ptr(), append_value.data_ptr(), append_key.stride(0), append_key.stride(1),
                append_value.stride(0), append_value.stride(1), batch_indices.data_ptr(), positions.data_ptr(),
                nnz, C.byref(kv), _lib.current_stream(dev),
            ),
            "append_paged_kv_cache",
        )


def apply_rope_append_paged_kv_cache(
    q: torch.Tensor,
    append_key: torch.Tensor,
    append_value: torch.Tensor,
    batch_indices: torch.Tensor,
    positions: torch.Tensor,
    paged_kv_cache: Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]],
    kv_indices: torch.Tensor,
    kv_indptr: torch.Tensor,
    kv_last_page_len: torch.Tensor,
    kv_layout: str = "NHD",
    rotary_dim: Optional[int] = None,
    interleave: bool = False,
    rope_scale: float = 1,
    rope_theta: float = 1e4,
    pos_ids: Optional[torch.Tensor] = None,
    q_out: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    r"""RoPE fused with the cache append: ``apply_rope_pos_ids(q, append_key, pos_ids)`` followed by
    ``append_paged_kv_cache(k_rope, append_value, ...)`` in one kernel -- every rotated key row is written once,
    straight into its page, instead of out to a temporary and back in (the serving-loop step right before the
    attention path).  Returns the rotated queries (``q_out`` if given; pass ``q_out=q`` for
    in place).  ``pos_ids`` defaults to ``positions`` (the row's position in its sequence); the cache must have the
    dtype of ``append_key``.  Bit-identical to the two calls it replaces.
    (ref: flashinfer/rope.py:321-420 apply_rope_pos_ids, flashinfer/page.py:299-425 append_paged_kv_cache)"""
    _check_kv_layout(kv_layout)
    for t, name in ((q, "q"), (append_key, "append_key"), (append_value, "append_value")):
        _lib.require_gpu_tensor(t, name)
    k_cache, v_cache = _unpack_paged_kv_cache(paged_kv_cache, kv_layout)
    if append_key.dtype != k_cache.dtype or append_value.dtype != v_cache.dtype or q.dtype != append_key.dtype:
        raise ValueError("q / append_key / append_value dtype must match the cache dtype")
    if q.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError("q and append_key must be float16 or bfloat16")
    if append_key.dim() != 3 or append_key.shape != append_value.shape or q.dim() != 3:
        raise ValueError("q, append_key and append_value must be [nnz, heads, head_dim]")
    page_size, num_kv_heads, head_dim, stride_page, stride_n, stride_h = paged_kv_strides(k_cache, v_cache, kv_layout)
    if append_key.shape[1] != num_kv_heads or append_key.shape[2] != head_dim or q.shape[2] != head_dim or \
            q.shape[0] != append_key.shape[0]:
        raise ValueError("q / append_key shape does not match the cache")
    if q.stride(-1) != 1:
        q = q.contiguous()
    if append_key.stride(-1) != 1:
        append_key = append_key.contiguous()
    if append_value.stride(-1) != 1:
        append_value = append_value.contiguous()
    if q_out is None:
        q_out = torch.empty_like(q)
    elif q_out.shape != q.shape or q_out.dtype != q.dtype or q_out.stride(-1) != 1:
        raise ValueError("q_out must match q")
    dev = k_cache.device
    batch_indices = batch_indices.to(device=dev, dtype=torch.int32).contiguous()
    positions = positions.to(device=dev, dtype=torch.int32).contiguous()
    pos_ids = positions if pos_ids is None else pos_ids.to(device=dev, dtype=torch.int32).contiguous()
    kv_indices = kv_indices.to(device=dev, dtype=torch.int32).contiguous()
    kv_indptr = kv_indptr.to(device=dev, dtype=torch.int32).contiguous()
    nnz = append_key.shape[0]
    if batch_indices.numel() != nnz or positions.numel() != nnz or pos_ids.numel() != nnz:
        raise ValueError("batch_indices, positions and pos_ids must have nnz entries")
    kv = _lib.PagedKV(
        k_data=k_cache.data_ptr(), v_data=v_cache.data_ptr(), indptr=kv_indptr.data_ptr(),
        indices=kv_indices.data_ptr(), last_page_len=None, rope_pos_offset=None, stride_page=stride_page,
        stride_n=stride_n, stride_h=stride_h, page_size=page_size, num_kv_heads=num_kv_heads,
        head_dim=head_dim, batch_size=kv_indptr.numel() - 1, dtype=_lib.fi_dtype(k_cache.dtype),
    )
    params = _lib.RopeParams(
        q=q.data_ptr(), k=append_key.data_ptr(), q_out=q_out.data_ptr(), k_out=None, pos_ids=pos_ids.data_ptr(),
        cos_sin_cache=None, q_stride_n=q.stride(0), q_stride_h=q.stride(1), k_stride_n=append_key.stride(0),
        k_stride_h=append_key.stride(1), qo_stride_n=q_out.stride(0), qo_stride_h=q_out.stride(1), ko_stride_n=0,
        ko_stride_h=0, nnz=nnz, num_q_heads=q.shape[1], num_k_heads=num_kv_heads, head_dim=head_dim,
        rotary_dim=head_dim if rotary_dim is None else rotary_dim, interleave=int(interleave),
        dtype=_lib.fi_dtype(q.dtype), rope_rcp_scale=1.0 / rope_scale, rope_rcp_theta=1.0 / rope_theta,
        smooth_a=0.0, smooth_b=0.0,
    )
    with torch.cuda.device(dev):
        _lib.check(
            _lib.lib().fi_apply_rope_append_paged_kv_cache(
                C.byref(params), append_value.data_ptr(), append_value.stride(0), append_value.stride(1),
                batch_indices.data_ptr(), positions.data_ptr(), C.byref(kv), _lib.current_stream(dev),
            ),
            "apply_rope_append_paged_kv_cache",
        )
    return q_out

"""Attention-state merge operators (cascade inference).

API of the reference's ``flashinfer/cascade.py`` (merge_state :45-100, merge_state_in_place
:111-158, merge_states :171-216); the kernels are in csrc/merge.hip behind the C ABI.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib


def _check_vs(v: torch.Tensor, s: torch.Tensor, vd: int, name: str) -> None:
    _lib.require_gpu_tensor(v, name)
    if v.dim() != vd or s.dim() != vd - 1:
        raise ValueError(f"{name}: expected v of {vd} dims and s of {vd - 1} dims")
    if v.shape[:-1] != s.shape:
        raise ValueError(f"{name}: shape of s {tuple(s.shape)} does not match v {tuple(v.shape)}")
    if s.device != v.device:
        raise ValueError(f"{name}: v and s must be on the same device")


def merge_state(
    v_a: torch.Tensor, s_a: torch.Tensor, v_b: torch.Tensor, s_b: torch.Tensor
) -> Tuple[torch.Tensor, torch.Tensor]:
    r"""Merge the attention output ``V`` and the base-2 logsumexp ``S`` from two KV segments.

    Parameters
    ----------
    v_a, v_b : torch.Tensor
        Attention outputs, shape ``[seq_len, num_heads, head_dim]``.
    s_a, s_b : torch.Tensor
        Logsumexp values (float32), shape ``[seq_len, num_heads]``.

    Returns
    -------
    V, S : merged output ``[seq_len, num_heads, head_dim]`` and logsumexp ``[seq_len, num_heads]``.
    """
    _check_vs(v_a, s_a, 3, "merge_state(a)")
    _check_vs(v_b, s_b, 3, "merge_state(b)")
    if v_a.shape != v_b.shape or v_a.dtype != v_b.dtype:
        raise ValueError("merge_state: v_a and v_b must have the same shape and dtype")
    v_a, v_b = v_a.contiguous(), v_b.contiguous()
    s_a = s_a.to(torch.float32).contiguous()
    s_b = s_b.to(torch.float32).contiguous()
    seq_len, num_heads, head_dim = v_a.shape
    v_merged = torch.empty_like(v_a)
    s_merged = torch.empty_like(s_a)
    with torch.cuda.device(v_a.device):
        _lib.check(
            _lib.lib().fi_merge_state(
                v_a.data_ptr(), s_a.data_ptr(), v_b.data_ptr(), s_b.data_ptr(),
                v_merged.data_ptr(), s_merged.data_ptr(), seq_len, num_heads, head_dim,
                _lib.fi_dtype(v_a.dtype), _lib.current_stream(v_a.device),
            ),
            "merge_state",
        )
    return v_merged, s_merged


def merge_state_in_place(
    v: torch.Tensor,
    s: torch.Tensor,
    v_other: torch.Tensor,
    s_other: torch.Tensor,
    mask: Optional[torch.Tensor] = None,
) -> None:
    r"""Merge ``(v_other, s_other)`` into ``(v, s)`` in place.

    ``mask`` (optional, ``[seq_len]``, bool): rows with a false mask keep ``(v, s)`` unchanged
    (ref: flashinfer/cascade.py:111-158).
    """
    _check_vs(v, s, 3, "merge_state_in_place(v)")
    _check_vs(v_other, s_other, 3, "merge_state_in_place(other)")
    if v.shape != v_other.shape or v.dtype != v_other.dtype:
        raise ValueError("merge_state_in_place: v and v_other must have the same shape and dtype")
    if not v.is_contiguous() or not s.is_contiguous() or s.dtype != torch.float32:
        raise ValueError("merge_state_in_place: v, s must be contiguous and s float32")
    v_other = v_other.contiguous()
    s_other = s_other.to(torch.float32).contiguous()
    mask_u8 = None
    if mask is not None:
        if mask.dim() != 1 or mask.shape[0] != v.shape[0]:
            raise ValueError("merge_state_in_place: mask must have shape [seq_len]")
        mask_u8 = mask.to(device=v.device, dtype=torch.uint8).contiguous()
    seq_len, num_heads, head_dim = v.shape
    with torch.cuda.device(v.device):
        _lib.check(
            _lib.lib().fi_merge_state_in_place(
                v.data_ptr(), s.data_ptr(), v_other.data_ptr(), s_other.data_ptr(),
                _lib.ptr(mask_u8), seq_len, num_heads, head_dim, _lib.fi_dtype(v.dtype),
                _lib.current_stream(v.device),
            ),
            "merge_state_in_place",
        )


def merge_states(v: torch.Tensor, s: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    r"""Merge attention states from several KV segments.

    Parameters
    ----------
    v : ``[seq_len, num_states, num_heads, head_dim]``
    s : ``[seq_len, num_states, num_heads]`` float32 base-2 logsumexp
    """
    _check_vs(v, s, 4, "merge_states")
    v = v.contiguous()
    s = s.to(torch.float32).contiguous()
    seq_len, num_sets, num_heads, head_dim = v.shape
    v_merged = torch.empty(seq_len, num_heads, head_dim, dtype=v.dtype, device=v.device)
    s_merged = torch.empty(seq_len, num_heads, dtype=torch.float32, device=v.device)
    with torch.cuda.device(v.device):
        _lib.check(
            _lib.lib().fi_merge_states(
                v.data_ptr(), s.data_ptr(), v_merged.data_ptr(), s_merged.data_ptr(), num_sets,
                seq_len, num_heads, head_dim, _lib.fi_dtype(v.dtype), _lib.current_stream(v.device),
            ),
            "merge_states",
        )
    return v_merged, s_merged

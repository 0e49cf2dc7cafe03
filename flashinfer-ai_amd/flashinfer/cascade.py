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


# --------------------------------------------------------------------------------------------------
# cascade wrappers (host orchestration only; ref: flashinfer/cascade.py:228-795)
# --------------------------------------------------------------------------------------------------
from typing import List, Union  # noqa: E402

from .decode import BatchDecodeWithPagedKVCacheWrapper  # noqa: E402
from .prefill import BatchPrefillWithPagedKVCacheWrapper, single_prefill_with_kv_cache  # noqa: E402


class MultiLevelCascadeAttentionWrapper:
    r"""Multi-level cascade attention over one unified page table: level ``i`` is a
    :class:`BatchPrefillWithPagedKVCacheWrapper` whose "requests" are the KV segments shared at that
    level; per-level partial states are folded with :func:`merge_state_in_place`.

    >>> wrapper = flashinfer.MultiLevelCascadeAttentionWrapper(2, workspace_buffer, "NHD")
    >>> wrapper.plan([shared_qo_indptr, unique_qo_indptr], [shared_kv_indptr, unique_kv_indptr],
    ...              [shared_kv_indices, unique_kv_indices], [shared_last_page_len, unique_last_page_len],
    ...              num_qo_heads, num_kv_heads, head_dim, page_size)
    >>> o = wrapper.run(q, kv_cache)

    (ref: flashinfer/cascade.py:228-555; layout docs/tutorials/kv_layout.rst:182-201)
    """

    def __init__(
        self,
        num_levels,
        float_workspace_buffer: torch.Tensor,
        kv_layout: str = "NHD",
        use_cuda_graph: bool = False,
        qo_indptr_buf_arr: Optional[List[torch.Tensor]] = None,
        paged_kv_indptr_buf_arr: Optional[List[torch.Tensor]] = None,
        paged_kv_indices_buf_arr: Optional[List[torch.Tensor]] = None,
        paged_kv_last_page_len_buf_arr: Optional[List[torch.Tensor]] = None,
    ) -> None:
        self._use_cuda_graph = use_cuda_graph
        if use_cuda_graph:
            self._batch_prefill_wrappers = [
                BatchPrefillWithPagedKVCacheWrapper(
                    float_workspace_buffer, kv_layout, use_cuda_graph=True, qo_indptr_buf=qo_indptr_buf,
                    paged_kv_indptr_buf=paged_kv_indptr_buf, paged_kv_indices_buf=paged_kv_indices_buf,
                    paged_kv_last_page_len_buf=paged_kv_last_page_len_buf,
                )
                for (qo_indptr_buf, paged_kv_indptr_buf, paged_kv_indices_buf, paged_kv_last_page_len_buf)
                in zip(qo_indptr_buf_arr, paged_kv_indptr_buf_arr, paged_kv_indices_buf_arr,
                       paged_kv_last_page_len_buf_arr)
            ]
        else:
            self._batch_prefill_wrappers = [
                BatchPrefillWithPagedKVCacheWrapper(float_workspace_buffer, kv_layout)
                for _ in range(num_levels)
            ]
        self._num_levels = num_levels
        self._kv_layout = kv_layout

    @property
    def is_cuda_graph_enabled(self) -> bool:
        return self._use_cuda_graph

    def reset_workspace_buffer(
        self, float_workspace_buffer: torch.Tensor, int_workspace_buffers: List[torch.Tensor]
    ) -> None:
        for wrapper, int_workspace_buffer in zip(self._batch_prefill_wrappers, int_workspace_buffers):
            wrapper.reset_workspace_buffer(float_workspace_buffer, int_workspace_buffer)

    def plan(
        self,
        qo_indptr_arr: List[torch.Tensor],
        paged_kv_indptr_arr: List[torch.Tensor],
        paged_kv_indices_arr: List[torch.Tensor],
        paged_kv_last_page_len: List[torch.Tensor],
        num_qo_heads: int,
        num_kv_heads: int,
        head_dim: int,
        page_size: int,
        causal: bool = False,
        pos_encoding_mode: str = "NONE",
        use_fp16_qk_reduction: bool = False,
        sm_scale: Optional[float] = None,
        window_left: int = -1,
        logits_soft_cap: Optional[float] = None,
        rope_scale: Optional[float] = None,
        rope_theta: Optional[float] = None,
        q_data_type: str = "float16",
        kv_data_type: Optional[Union[str, torch.dtype]] = None,
    ):
        r"""Plan every level; the causal mask applies to the LAST level only (ref: cascade.py:505)."""
        for i, (wrapper, qo_indptr, paged_kv_indptr, paged_kv_indices, last_page_len) in enumerate(
            zip(self._batch_prefill_wrappers, qo_indptr_arr, paged_kv_indptr_arr, paged_kv_indices_arr,
                paged_kv_last_page_len)
        ):
            wrapper.plan(
                qo_indptr, paged_kv_indptr, paged_kv_indices, last_page_len, num_qo_heads, num_kv_heads,
                head_dim, page_size, causal=causal if i == self._num_levels - 1 else False,
                pos_encoding_mode=pos_encoding_mode, use_fp16_qk_reduction=use_fp16_qk_reduction,
                sm_scale=sm_scale, window_left=window_left, logits_soft_cap=logits_soft_cap,
                rope_scale=rope_scale, rope_theta=rope_theta, q_data_type=q_data_type,
                kv_data_type=kv_data_type,
            )

    begin_forward = plan

    def run(self, q: torch.Tensor, paged_kv_cache: torch.Tensor):
        r"""q ``[batch_size, num_qo_heads, head_dim]`` against the unified paged cache."""
        out, lse = self._batch_prefill_wrappers[-1].run(q, paged_kv_cache, return_lse=True)
        for wrapper in self._batch_prefill_wrappers[:-1]:
            out_i, lse_i = wrapper.run(q, paged_kv_cache, return_lse=True)
            merge_state_in_place(out, lse, out_i, lse_i)
        return out

    forward = run


class BatchDecodeWithSharedPrefixPagedKVCacheWrapper:
    r"""Two-level decode: one dense shared prefix + per-request paged suffixes
    (ref: flashinfer/cascade.py:558-795)."""

    def __init__(self, float_workspace_buffer: torch.Tensor, kv_layout: str = "NHD") -> None:
        self._batch_decode_wrapper = BatchDecodeWithPagedKVCacheWrapper(float_workspace_buffer, kv_layout)
        self._kv_layout = kv_layout

    def reset_workspace_buffer(self, float_workspace_buffer: torch.Tensor, int_workspace_buffer: torch.Tensor) -> None:
        self._batch_decode_wrapper.reset_workspace_buffer(float_workspace_buffer, int_workspace_buffer)

    def begin_forward(
        self,
        unique_kv_indptr: torch.Tensor,
        unique_kv_indices: torch.Tensor,
        unique_kv_last_page_len: torch.Tensor,
        num_qo_heads: int,
        num_kv_heads: int,
        head_dim: int,
        page_size: int,
        data_type: str = "float16",
    ) -> None:
        dt = getattr(torch, data_type) if isinstance(data_type, str) else data_type
        self._batch_decode_wrapper.plan(
            unique_kv_indptr, unique_kv_indices, unique_kv_last_page_len, num_qo_heads, num_kv_heads,
            head_dim, page_size, pos_encoding_mode="NONE", q_data_type=dt, kv_data_type=dt,
        )

    def forward(self, q: torch.Tensor, k_shared: torch.Tensor, v_shared: torch.Tensor,
                unique_kv_cache: torch.Tensor) -> torch.Tensor:
        r"""q ``[batch_size, num_qo_heads, head_dim]``; k_shared / v_shared the dense shared prefix."""
        V_shared, S_shared = single_prefill_with_kv_cache(
            q, k_shared, v_shared, causal=False, pos_encoding_mode="NONE", kv_layout=self._kv_layout,
            sm_scale=self._batch_decode_wrapper._sm_scale, rope_scale=self._batch_decode_wrapper._rope_scale,
            rope_theta=self._batch_decode_wrapper._rope_theta, return_lse=True,
        )
        V_unique, S_unique = self._batch_decode_wrapper.run(q, unique_kv_cache, return_lse=True)
        merge_state_in_place(V_shared, S_shared, V_unique, S_unique)
        return V_shared

    def end_forward(self) -> None:
        pass


class BatchPrefillWithSharedPrefixPagedKVCacheWrapper:
    r"""Two-level prefill/append: dense shared prefix + per-request paged suffixes
    (ref: flashinfer/cascade.py:798-1075)."""

    def __init__(self, float_workspace_buffer: torch.Tensor, kv_layout: str = "NHD") -> None:
        self._batch_prefill_wrapper = BatchPrefillWithPagedKVCacheWrapper(float_workspace_buffer, kv_layout)
        self._kv_layout = kv_layout

    def reset_workspace_buffer(self, float_workspace_buffer: torch.Tensor, int_workspace_buffer: torch.Tensor) -> None:
        self._batch_prefill_wrapper.reset_workspace_buffer(float_workspace_buffer, int_workspace_buffer)

    def begin_forward(self, qo_indptr, paged_kv_indptr, paged_kv_indices, paged_kv_last_page_len,
                      num_qo_heads: int, num_kv_heads: int, head_dim: int, page_size: int) -> None:
        self._plan_args = (qo_indptr, paged_kv_indptr, paged_kv_indices, paged_kv_last_page_len,
                           num_qo_heads, num_kv_heads, head_dim, page_size)
        self._planned_key = None

    def forward(self, q, k_shared, v_shared, unique_kv_cache, causal: bool = False,
                use_fp16_qk_reduction: bool = False, sm_scale: Optional[float] = None,
                rope_scale: Optional[float] = None, rope_theta: Optional[float] = None) -> torch.Tensor:
        from .utils import _unpack_paged_kv_cache

        k_cache, _ = _unpack_paged_kv_cache(unique_kv_cache, self._kv_layout)
        key = (causal, sm_scale, q.dtype, k_cache.dtype)
        if self._planned_key != key:  # the reference re-plans lazily with the forward-time options too
            self._batch_prefill_wrapper.plan(*self._plan_args, causal=causal, sm_scale=sm_scale,
                                             rope_scale=rope_scale, rope_theta=rope_theta,
                                             q_data_type=q.dtype, kv_data_type=k_cache.dtype)
            self._planned_key = key
        V_shared, S_shared = single_prefill_with_kv_cache(
            q, k_shared, v_shared, causal=False, pos_encoding_mode="NONE", kv_layout=self._kv_layout,
            sm_scale=sm_scale, rope_scale=rope_scale, rope_theta=rope_theta, return_lse=True,
        )
        V_unique, S_unique = self._batch_prefill_wrapper.run(q, unique_kv_cache, return_lse=True)
        merge_state_in_place(V_shared, S_shared, V_unique, S_unique)
        return V_shared

    def end_forward(self) -> None:
        pass

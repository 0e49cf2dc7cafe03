"""Module getters with the REFERENCE's FFI signatures, bound to libfi_mi355.so.

The reference's operator layer calls its kernels through per-specialisation modules returned by
`get_batch_decode_module`, `get_batch_prefill_module`, `get_cascade_module`, `get_page_module`,
`get_gemm_sm100_module` (flashinfer/decode.py:208-302, prefill.py:364-724, cascade.py:32-42, page.py:34-40,
gemm.py:2190-2240); each module exposes typed functions taking torch tensors in a fixed positional order
(TVM_FFI_DLL_EXPORT_TYPED_FUNC in csrc/*_binding.cu).  A maintainer who keeps `flashinfer/*.py` unchanged swaps
only those getters: this file IS that binding -- every function below has the positional signature of the
export it replaces (cited per function) and forwards to the C ABI of include/fi_mi355.h.  What was a
compile-time specialisation in the reference (dtypes, head_dim, pos_encoding_mode, sliding window, soft cap)
is an argument of the getter here too and becomes a run-time field of the C structs.
tests/test_integration_shim.py drives these modules with the reference's call sequences.
"""
from __future__ import annotations

import ctypes as C
import functools
from types import SimpleNamespace
from typing import List, Optional

import torch

from . import _lib


def _nbytes(t: torch.Tensor) -> int:
    return t.numel() * t.element_size()


def _stream(t: torch.Tensor) -> int:
    return _lib.current_stream(t.device)


def _paged_kv(k_cache, v_cache, kv_indptr, kv_indices, kv_last_page_len, layout_code: int, batch_size: int):
    """paged_kv_t from the 4-D K / V views the reference passes (csrc/batch_decode.cu:94-142):
    layout 0 = NHD [pages, page_size, H, D], 1 = HND [pages, H, page_size, D]; strides from the tensors."""
    nhd = layout_code == 0
    if k_cache.stride() != v_cache.stride():
        raise ValueError("k/v strides must be identical")  # ref: csrc/batch_decode.cu:118-129
    return _lib.PagedKV(
        k_data=k_cache.data_ptr(), v_data=v_cache.data_ptr(), indptr=kv_indptr.data_ptr(),
        indices=_lib.ptr(kv_indices), last_page_len=_lib.ptr(kv_last_page_len), rope_pos_offset=None,
        stride_page=k_cache.stride(0), stride_n=k_cache.stride(1 if nhd else 2), stride_h=k_cache.stride(2 if nhd else 1),
        page_size=k_cache.shape[1 if nhd else 2], num_kv_heads=k_cache.shape[2 if nhd else 1],
        head_dim=k_cache.shape[3], batch_size=batch_size, dtype=_lib.fi_dtype(k_cache.dtype))


@functools.cache
def get_batch_decode_module(dtype_q, dtype_kv, dtype_o, idtype, head_dim_qk, head_dim_vo, pos_encoding_mode,
                            use_sliding_window, use_logits_soft_cap):
    """ref getter: flashinfer/decode.py:208-302 (same argument list)."""
    lib = _lib.lib()
    if idtype != torch.int32:
        raise ValueError("page tables must be int32")

    def plan(float_workspace_buffer, int_workspace_buffer, page_locked_int_workspace_buffer, indptr, batch_size,
             num_qo_heads, num_kv_heads, page_size, enable_cuda_graph, window_left, logits_soft_cap, head_dim_qk_,
             head_dim_vo_, empty_q_data, empty_kv_data) -> List[int]:
        """ref: BatchDecodeWithPagedKVCachePlan, csrc/batch_decode.cu:39-44 (15 positional arguments; `indptr` is
        the HOST copy, flashinfer/decode.py:1079-1095)."""
        info = (C.c_int64 * _lib.FI_DECODE_PLAN_INFO_LEN)()
        with torch.cuda.device(float_workspace_buffer.device):
            _lib.check(lib.fi_batch_decode_plan(
                float_workspace_buffer.data_ptr(), _nbytes(float_workspace_buffer), int_workspace_buffer.data_ptr(),
                page_locked_int_workspace_buffer.data_ptr(), _nbytes(int_workspace_buffer), indptr.data_ptr(),
                batch_size, num_qo_heads, num_kv_heads, page_size, int(enable_cuda_graph), head_dim_qk_,
                _lib.fi_dtype(empty_q_data.dtype), _lib.fi_dtype(empty_kv_data.dtype), 0,
                window_left if use_sliding_window else -1, info, _stream(float_workspace_buffer)), "batch_decode.plan")
        return list(info)

    def run(float_workspace_buffer, int_workspace_buffer, plan_info_vec, q, paged_k_cache, paged_v_cache,
            paged_kv_indptr, paged_kv_indices, paged_kv_last_page_len, o, maybe_lse, kv_layout_code, window_left,
            enable_pdl, maybe_alibi_slopes, logits_soft_cap, sm_scale, rope_rcp_scale, rope_rcp_theta) -> None:
        """ref: BatchDecodeWithPagedKVCacheRun, csrc/batch_decode.cu:81-86 + the default additional parameters
        (flashinfer/jit/attention/modules.py:764-772); the reference's Python shim passes 1/rope_scale and
        1/rope_theta here (flashinfer/decode.py:264-268)."""
        kv = _paged_kv(paged_k_cache, paged_v_cache, paged_kv_indptr, paged_kv_indices, paged_kv_last_page_len,
                       kv_layout_code, q.shape[0])
        p = _lib.BatchDecodeParams(
            q=q.data_ptr(), q_stride_n=q.stride(0), q_stride_h=q.stride(1), kv=kv, o=o.data_ptr(),
            lse=_lib.ptr(maybe_lse), alibi_slopes=_lib.ptr(maybe_alibi_slopes), q_rope_offset=None,
            num_qo_heads=q.shape[1], q_dtype=_lib.fi_dtype(q.dtype), pos_encoding_mode=pos_encoding_mode,
            window_left=window_left if use_sliding_window else -1,
            logits_soft_cap=logits_soft_cap if use_logits_soft_cap else 0.0, sm_scale=sm_scale,
            rope_rcp_scale=rope_rcp_scale, rope_rcp_theta=rope_rcp_theta)
        info = (C.c_int64 * _lib.FI_DECODE_PLAN_INFO_LEN)(*plan_info_vec)
        with torch.cuda.device(q.device):
            _lib.check(lib.fi_batch_decode_run(
                float_workspace_buffer.data_ptr(), _nbytes(float_workspace_buffer), int_workspace_buffer.data_ptr(),
                _nbytes(int_workspace_buffer), info, _lib.FI_DECODE_PLAN_INFO_LEN, C.byref(p), _stream(q)),
                "batch_decode.run")

    return SimpleNamespace(plan=plan, run=run)


@functools.cache
def get_batch_prefill_module(backend, dtype_q, dtype_kv, dtype_o, idtype, head_dim_qk, head_dim_vo,
                             pos_encoding_mode, use_sliding_window, use_logits_soft_cap, use_fp16_qk_reduction=False):
    """ref getter: flashinfer/prefill.py:364-724 (backend "fa2" / "fa3" name the same kernels here)."""
    lib = _lib.lib()
    if idtype != torch.int32:
        raise ValueError("page tables must be int32")

    def plan(float_workspace_buffer, int_workspace_buffer, page_locked_int_workspace_buffer, qo_indptr, kv_indptr,
             kv_len_arr, total_num_rows, batch_size, num_qo_heads, num_kv_heads, page_size, enable_cuda_graph,
             head_dim_qk_, head_dim_vo_, causal, window_left, fixed_split_size=-1, disable_split_kv=False) -> List[int]:
        """ref: BatchPrefillWithKVCachePlan, csrc/batch_prefill.cu:47-52 (18 positional arguments; the three index
        tensors are HOST tensors, flashinfer/prefill.py:1884-1908; the fa3 plan stops at window_left,
        csrc/batch_prefill_fp8_sm90.cu:39-44)."""
        info = (C.c_int64 * _lib.FI_PREFILL_PLAN_INFO_LEN)()
        with torch.cuda.device(float_workspace_buffer.device):
            _lib.check(lib.fi_batch_prefill_plan(
                float_workspace_buffer.data_ptr(), _nbytes(float_workspace_buffer), int_workspace_buffer.data_ptr(),
                page_locked_int_workspace_buffer.data_ptr(), _nbytes(int_workspace_buffer), qo_indptr.data_ptr(),
                kv_indptr.data_ptr(), kv_len_arr.data_ptr(), total_num_rows, batch_size, num_qo_heads, num_kv_heads,
                page_size, int(enable_cuda_graph), head_dim_qk_, head_dim_vo_, int(causal),
                window_left if use_sliding_window else -1, fixed_split_size, int(disable_split_kv), info,
                _stream(float_workspace_buffer)), "batch_prefill.plan")
        return list(info)

    def _run(float_ws, int_ws, plan_info_vec, q, k_cache, v_cache, qo_indptr, kv, o, maybe_lse, mask_mode_code,
             window_left, custom_mask, mask_indptr, alibi_slopes, prefix_len_ptr, token_pos_in_items_ptr,
             max_item_len_ptr, token_pos_in_items_len, logits_soft_cap, sm_scale, rope_rcp_scale, rope_rcp_theta,
             scale_q, scale_k, scale_v, what):
        p = _lib.BatchPrefillParams(
            q=q.data_ptr(), q_stride_n=q.stride(0), q_stride_h=q.stride(1), qo_indptr=qo_indptr.data_ptr(), kv=kv,
            o=o.data_ptr(), lse=_lib.ptr(maybe_lse), alibi_slopes=_lib.ptr(alibi_slopes), scale_q=_lib.ptr(scale_q),
            scale_k=_lib.ptr(scale_k), scale_v=_lib.ptr(scale_v), custom_mask=_lib.ptr(custom_mask),
            mask_indptr=_lib.ptr(mask_indptr), prefix_len_ptr=_lib.ptr(prefix_len_ptr),
            token_pos_in_items_ptr=_lib.ptr(token_pos_in_items_ptr), max_item_len_ptr=_lib.ptr(max_item_len_ptr),
            token_pos_in_items_len=int(token_pos_in_items_len or 0),
            num_qo_heads=q.shape[1], q_dtype=_lib.fi_dtype(q.dtype), o_dtype=_lib.fi_dtype(o.dtype),
            mask_mode=mask_mode_code, pos_encoding_mode=pos_encoding_mode,
            window_left=window_left if use_sliding_window else -1,
            logits_soft_cap=logits_soft_cap if use_logits_soft_cap else 0.0, sm_scale=sm_scale,
            rope_rcp_scale=rope_rcp_scale, rope_rcp_theta=rope_rcp_theta)
        info = (C.c_int64 * _lib.FI_PREFILL_PLAN_INFO_LEN)(*plan_info_vec)
        with torch.cuda.device(q.device):
            _lib.check(lib.fi_batch_prefill_paged_run(
                float_ws.data_ptr(), _nbytes(float_ws), int_ws.data_ptr(), _nbytes(int_ws), info,
                _lib.FI_PREFILL_PLAN_INFO_LEN, C.byref(p), _stream(q)), what)

    def _tail16(additional):
        """additional parameters of a 16-bit-q call, normalised to the fa2 order.  The reference's fa3 (Hopper)
        specialisation passes SIX: maybe_prefix_len_ptr, maybe_token_pos_in_items_ptr, maybe_max_item_len_ptr,
        logits_soft_cap, sm_scale, token_pos_in_items_len (flashinfer/prefill.py:624-646) -- no custom mask, ALiBi or
        fused RoPE there; fa2 passes eleven (prefill.py:604-622)."""
        if len(additional) == 6:
            prefix_len_ptr, token_pos_in_items_ptr, max_item_len_ptr, logits_soft_cap, sm_scale, tp_len = additional
            return (None, None, None, prefix_len_ptr, token_pos_in_items_ptr, max_item_len_ptr, logits_soft_cap,
                    sm_scale, 1.0, 1e-4, tp_len)
        if len(additional) != 11:
            raise ValueError(f"batch_prefill run: {len(additional)} additional parameters; the fa2 form has 11, "
                             f"the fa3 form 6, the fp8 form 4")
        return additional

    def paged_run(float_workspace_buffer, int_workspace_buffer, plan_info_vec, q, paged_k_cache, paged_v_cache,
                  qo_indptr, paged_kv_indptr, paged_kv_indices, paged_kv_last_page_len, o, maybe_lse, mask_mode_code,
                  layout, window_left, enable_pdl, *additional) -> None:
        """ref: BatchPrefillWithPagedKVCacheRun, csrc/batch_prefill.cu:199-205, followed by the additional
        parameters of the specialisation (flashinfer/jit/attention/modules.py:815-861):
          16-bit q : maybe_custom_mask, maybe_mask_indptr, maybe_alibi_slopes, maybe_prefix_len_ptr,
                     maybe_token_pos_in_items_ptr, maybe_max_item_len_ptr, logits_soft_cap, sm_scale,
                     rope_rcp_scale, rope_rcp_theta, token_pos_in_items_len          (flashinfer/prefill.py:620-650)
          fp8 q    : scale_q, scale_k, scale_v, sm_scale  (csrc/batch_prefill_fp8_sm90.cu:81-90, prefill.py:676-697)"""
        kv = _paged_kv(paged_k_cache, paged_v_cache, paged_kv_indptr, paged_kv_indices, paged_kv_last_page_len,
                       layout, paged_kv_indptr.shape[0] - 1)
        if q.dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
            scale_q, scale_k, scale_v, sm_scale = additional
            _run(float_workspace_buffer, int_workspace_buffer, plan_info_vec, q, paged_k_cache, paged_v_cache, qo_indptr,
                 kv, o, maybe_lse, mask_mode_code, window_left, None, None, None, None, None, None, 0, 0.0, sm_scale,
                 1.0, 1e-4, scale_q, scale_k, scale_v, "batch_prefill.paged_run(fp8)")
        else:
            (custom_mask, mask_indptr, alibi_slopes, prefix_len_ptr, token_pos_in_items_ptr, max_item_len_ptr,
             logits_soft_cap, sm_scale, rope_rcp_scale, rope_rcp_theta, token_pos_in_items_len) = _tail16(additional)
            _run(float_workspace_buffer, int_workspace_buffer, plan_info_vec, q, paged_k_cache, paged_v_cache, qo_indptr,
                 kv, o, maybe_lse, mask_mode_code, window_left, custom_mask, mask_indptr, alibi_slopes, prefix_len_ptr,
                 token_pos_in_items_ptr, max_item_len_ptr, token_pos_in_items_len, logits_soft_cap, sm_scale,
                 rope_rcp_scale, rope_rcp_theta, None, None, None, "batch_prefill.paged_run")

    def ragged_run(float_workspace_buffer, int_workspace_buffer, plan_info_vec, q, k, v, qo_indptr, kv_indptr, o,
                   maybe_lse, mask_mode_code, layout, window_left, enable_pdl, *additional) -> None:
        """ref: BatchPrefillWithRaggedKVCacheRun, csrc/batch_prefill.cu:76-82 (k, v ragged [nnz, H, D] (layout 0) or
        [H, nnz, D] (layout 1)); same additional parameters as paged_run."""
        nhd = layout == 0
        stride_n, stride_h = (k.stride(0), k.stride(1)) if nhd else (k.stride(1), k.stride(0))
        kv = _lib.PagedKV(
            k_data=k.data_ptr(), v_data=v.data_ptr(), indptr=kv_indptr.data_ptr(), indices=None, last_page_len=None,
            rope_pos_offset=None, stride_page=stride_n, stride_n=stride_n, stride_h=stride_h, page_size=1,
            num_kv_heads=k.shape[1 if nhd else 0], head_dim=k.shape[2], batch_size=kv_indptr.shape[0] - 1,
            dtype=_lib.fi_dtype(k.dtype))
        (custom_mask, mask_indptr, alibi_slopes, prefix_len_ptr, token_pos_in_items_ptr, max_item_len_ptr,
         logits_soft_cap, sm_scale, rope_rcp_scale, rope_rcp_theta, token_pos_in_items_len) = _tail16(additional)
        _run(float_workspace_buffer, int_workspace_buffer, plan_info_vec, q, k, v, qo_indptr, kv, o, maybe_lse,
             mask_mode_code, window_left, custom_mask, mask_indptr, alibi_slopes, prefix_len_ptr,
             token_pos_in_items_ptr, max_item_len_ptr, token_pos_in_items_len, logits_soft_cap, sm_scale,
             rope_rcp_scale, rope_rcp_theta, None, None, None, "batch_prefill.ragged_run")

    return SimpleNamespace(plan=plan, paged_run=paged_run, ragged_run=ragged_run)


@functools.cache
def get_cascade_module():
    """ref getter: flashinfer/cascade.py:32-42; exports csrc/flashinfer_cascade_binding.cu:29-33."""
    lib = _lib.lib()

    def merge_state(v_a, s_a, v_b, s_b, v_merged, s_merged) -> None:
        """ref: csrc/cascade.cu:23-57."""
        n, h, d = v_a.shape
        with torch.cuda.device(v_a.device):
            _lib.check(lib.fi_merge_state(v_a.data_ptr(), s_a.data_ptr(), v_b.data_ptr(), s_b.data_ptr(),
                                          v_merged.data_ptr(), s_merged.data_ptr(), n, h, d, _lib.fi_dtype(v_a.dtype),
                                          _stream(v_a)), "merge_state")

    def merge_state_in_place(v, s, v_other, s_other, mask: Optional[torch.Tensor] = None) -> None:
        """ref: csrc/cascade.cu:59-100 (mask: optional uint8 / bool [seq_len])."""
        n, h, d = v.shape
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        with torch.cuda.device(v.device):
            _lib.check(lib.fi_merge_state_in_place(v.data_ptr(), s.data_ptr(), v_other.data_ptr(), s_other.data_ptr(),
                                                   _lib.ptr(m), n, h, d, _lib.fi_dtype(v.dtype), _stream(v)),
                       "merge_state_in_place")

    def merge_states(v, s, v_merged, s_merged) -> None:
        """ref: csrc/cascade.cu:102-...  v [seq_len, num_index_sets, H, D]."""
        n, sets, h, d = v.shape
        with torch.cuda.device(v.device):
            _lib.check(lib.fi_merge_states(v.data_ptr(), s.data_ptr(), v_merged.data_ptr(), s_merged.data_ptr(), sets,
                                           n, h, d, _lib.fi_dtype(v.dtype), _stream(v)), "merge_states")

    return SimpleNamespace(merge_state=merge_state, merge_state_in_place=merge_state_in_place, merge_states=merge_states)


@functools.cache
def get_page_module():
    """ref getter: flashinfer/page.py:34-40; export csrc/flashinfer_page_binding.cu:36."""
    lib = _lib.lib()

    def append_paged_kv_cache(append_key, append_value, batch_indices, positions, paged_k_cache, paged_v_cache,
                              kv_indices, kv_indptr, kv_last_page_len, layout) -> None:
        """ref: append_paged_kv_cache, csrc/page.cu:28-33 (10 positional arguments, flashinfer/page.py:411-424)."""
        kv = _paged_kv(paged_k_cache, paged_v_cache, kv_indptr, kv_indices, kv_last_page_len, layout,
                       kv_indptr.shape[0] - 1)
        with torch.cuda.device(append_key.device):
            _lib.check(lib.fi_append_paged_kv_cache(
                append_key.data_ptr(), append_value.data_ptr(), append_key.stride(0), append_key.stride(1),
                append_value.stride(0), append_value.stride(1), batch_indices.data_ptr(), positions.data_ptr(),
                append_key.shape[0], C.byref(kv), _stream(append_key)), "append_paged_kv_cache")

    return SimpleNamespace(append_paged_kv_cache=append_paged_kv_cache)


@functools.cache
def get_gemm_sm100_module():
    """ref getter: flashinfer/gemm.py (get_gemm_sm100_module); exports csrc/gemm_sm100_binding.cu:23 and
    csrc/group_gemm_sm100_binding.cu:34.  The workspaces and mma_sm are accepted and unused."""
    lib = _lib.lib()

    def gemm_fp8_nt_groupwise(workspace_buffer, a, b, a_scale, b_scale, out, scale_granularity_m,
                              scale_granularity_n, scale_granularity_k, scale_major_mode: str, mma_sm) -> None:
        """ref: CutlassGemmGroupwiseScaledSM100, csrc/gemm_groupwise_sm100.cu:89-95."""
        with torch.cuda.device(a.device):
            _lib.check(lib.fi_gemm_fp8_nt_groupwise(
                a.data_ptr(), b.data_ptr(), a_scale.data_ptr(), b_scale.data_ptr(), out.data_ptr(), a.shape[0],
                b.shape[0], a.shape[1], scale_granularity_m, scale_granularity_n, scale_granularity_k,
                int(scale_major_mode == "K"), _lib.fi_dtype(a.dtype), _lib.fi_dtype(b.dtype), _lib.fi_dtype(out.dtype),
                _stream(a)), "gemm_fp8_nt_groupwise")

    def group_gemm_fp8_nt_groupwise(int_workspace_buffer, float_workspace_buffer, a, b, a_scale, b_scale, out,
                                    m_indptr, n, k, scale_granularity_m, scale_granularity_n, scale_granularity_k,
                                    scale_major_mode: str, mma_sm) -> None:
        """ref: CutlassGroupGemmFP8GroupwiseScaledSM100, csrc/group_gemm_fp8_groupwise_sm100.cu:89-96
        (flashinfer/gemm.py:2791-2806)."""
        with torch.cuda.device(a.device):
            _lib.check(lib.fi_group_gemm_fp8_nt_groupwise(
                a.data_ptr(), b.data_ptr(), a_scale.data_ptr(), b_scale.data_ptr(), out.data_ptr(),
                m_indptr.data_ptr(), m_indptr.shape[0] - 1, a.shape[0], n, k, scale_granularity_m,
                scale_granularity_n, scale_granularity_k, int(scale_major_mode == "K"), _lib.fi_dtype(a.dtype),
                _lib.fi_dtype(b.dtype), _lib.fi_dtype(out.dtype), _stream(a)), "group_gemm_fp8_nt_groupwise")

    return SimpleNamespace(gemm_fp8_nt_groupwise=gemm_fp8_nt_groupwise,
                           group_gemm_fp8_nt_groupwise=group_gemm_fp8_nt_groupwise)

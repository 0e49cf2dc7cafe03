"""ctypes binding of libfi_mi355.so (the C ABI declared in include/fi_mi355.h).

The reference loads one JIT-built module per kernel specialisation through tvm_ffi
(ref: flashinfer/jit/core.py:247-263); here ONE ahead-of-time library serves every op.  There is no
fallback: if the library is missing every op raises (a CPU path would void the parity claims).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("FI_MI355_LIB", os.path.join(_HERE, "libfi_mi355.so"))

FI_DTYPE_F16, FI_DTYPE_BF16, FI_DTYPE_FP8_E4M3, FI_DTYPE_FP8_E5M2, FI_DTYPE_F32 = range(5)
FI_NEG_INF = -5.0e4
FI_DECODE_PLAN_INFO_LEN = 16

_TORCH2FI = {
    torch.float16: FI_DTYPE_F16,
    torch.bfloat16: FI_DTYPE_BF16,
    torch.float8_e4m3fn: FI_DTYPE_FP8_E4M3,
    torch.float8_e5m2: FI_DTYPE_FP8_E5M2,
    torch.float32: FI_DTYPE_F32,
}


def fi_dtype(dtype: torch.dtype) -> int:
    try:
        return _TORCH2FI[dtype]
    except KeyError:
        raise ValueError(f"unsupported dtype {dtype} for the MI355X kernels") from None


class PagedKV(C.Structure):
    _fields_ = [
        ("k_data", C.c_void_p),
        ("v_data", C.c_void_p),
        ("indptr", C.c_void_p),
        ("indices", C.c_void_p),
        ("last_page_len", C.c_void_p),
        ("rope_pos_offset", C.c_void_p),
        ("stride_page", C.c_int64),
        ("stride_n", C.c_int64),
        ("stride_h", C.c_int64),
        ("page_size", C.c_int32),
        ("num_kv_heads", C.c_int32),
        ("head_dim", C.c_int32),
        ("batch_size", C.c_int32),
        ("dtype", C.c_int32),
    ]


class BatchDecodeParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p),
        ("q_stride_n", C.c_int64),
        ("q_stride_h", C.c_int64),
        ("kv", PagedKV),
        ("o", C.c_void_p),
        ("lse", C.c_void_p),
        ("alibi_slopes", C.c_void_p),
        ("q_rope_offset", C.c_void_p),
        ("num_qo_heads", C.c_int32),
        ("q_dtype", C.c_int32),
        ("pos_encoding_mode", C.c_int32),
        ("window_left", C.c_int32),
        ("logits_soft_cap", C.c_float),
        ("sm_scale", C.c_float),
        ("rope_rcp_scale", C.c_float),
        ("rope_rcp_theta", C.c_float),
    ]


class SingleDecodeParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p),
        ("q_stride_h", C.c_int64),
        ("k", C.c_void_p),
        ("v", C.c_void_p),
        ("kv_stride_n", C.c_int64),
        ("kv_stride_h", C.c_int64),
        ("o", C.c_void_p),
        ("lse", C.c_void_p),
        ("alibi_slopes", C.c_void_p),
        ("kv_len", C.c_int32),
        ("num_qo_heads", C.c_int32),
        ("num_kv_heads", C.c_int32),
        ("head_dim", C.c_int32),
        ("q_dtype", C.c_int32),
        ("kv_dtype", C.c_int32),
        ("pos_encoding_mode", C.c_int32),
        ("window_left", C.c_int32),
        ("logits_soft_cap", C.c_float),
        ("sm_scale", C.c_float),
        ("rope_rcp_scale", C.c_float),
        ("rope_rcp_theta", C.c_float),
    ]


class BatchPrefillParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p),
        ("q_stride_n", C.c_int64),
        ("q_stride_h", C.c_int64),
        ("qo_indptr", C.c_void_p),
        ("kv", PagedKV),
        ("o", C.c_void_p),
        ("lse", C.c_void_p),
        ("alibi_slopes", C.c_void_p),
        ("scale_q", C.c_void_p),
        ("scale_k", C.c_void_p),
        ("scale_v", C.c_void_p),
        ("custom_mask", C.c_void_p),
        ("mask_indptr", C.c_void_p),
        ("prefix_len_ptr", C.c_void_p),
        ("token_pos_in_items_ptr", C.c_void_p),
        ("max_item_len_ptr", C.c_void_p),
        ("token_pos_in_items_len", C.c_int32),
        ("num_qo_heads", C.c_int32),
        ("q_dtype", C.c_int32),
        ("o_dtype", C.c_int32),
        ("mask_mode", C.c_int32),
        ("pos_encoding_mode", C.c_int32),
        ("window_left", C.c_int32),
        ("logits_soft_cap", C.c_float),
        ("sm_scale", C.c_float),
        ("rope_rcp_scale", C.c_float),
        ("rope_rcp_theta", C.c_float),
        ("bf16_pv_mode", C.c_int32),
    ]


class SinglePrefillParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p),
        ("q_stride_n", C.c_int64),
        ("q_stride_h", C.c_int64),
        ("k", C.c_void_p),
        ("v", C.c_void_p),
        ("kv_stride_n", C.c_int64),
        ("kv_stride_h", C.c_int64),
        ("o", C.c_void_p),
        ("lse", C.c_void_p),
        ("alibi_slopes", C.c_void_p),
        ("scale_q", C.c_void_p),
        ("scale_k", C.c_void_p),
        ("scale_v", C.c_void_p),
        ("custom_mask", C.c_void_p),
        ("qo_len", C.c_int32),
        ("kv_len", C.c_int32),
        ("num_qo_heads", C.c_int32),
        ("num_kv_heads", C.c_int32),
        ("head_dim", C.c_int32),
        ("q_dtype", C.c_int32),
        ("kv_dtype", C.c_int32),
        ("o_dtype", C.c_int32),
        ("mask_mode", C.c_int32),
        ("pos_encoding_mode", C.c_int32),
        ("window_left", C.c_int32),
        ("logits_soft_cap", C.c_float),
        ("sm_scale", C.c_float),
        ("rope_rcp_scale", C.c_float),
        ("rope_rcp_theta", C.c_float),
        ("bf16_pv_mode", C.c_int32),
    ]


class RopeParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("q_out", C.c_void_p), ("k_out", C.c_void_p),
        ("pos_ids", C.c_void_p), ("cos_sin_cache", C.c_void_p),
        ("q_stride_n", C.c_int64), ("q_stride_h", C.c_int64), ("k_stride_n", C.c_int64), ("k_stride_h", C.c_int64),
        ("qo_stride_n", C.c_int64), ("qo_stride_h", C.c_int64), ("ko_stride_n", C.c_int64), ("ko_stride_h", C.c_int64),
        ("nnz", C.c_int32), ("num_q_heads", C.c_int32), ("num_k_heads", C.c_int32), ("head_dim", C.c_int32),
        ("rotary_dim", C.c_int32), ("interleave", C.c_int32), ("dtype", C.c_int32),
        ("rope_rcp_scale", C.c_float), ("rope_rcp_theta", C.c_float), ("smooth_a", C.c_float), ("smooth_b", C.c_float),
    ]


FI_PREFILL_PLAN_INFO_LEN = 16

_lib: Optional[C.CDLL] = None

# every symbol include/fi_mi355.h declares; tests check the library exports all of them
EXPORTED_SYMBOLS = [
    "fi_last_error",
    "fi_abi_version",
    "fi_num_compute_units",
    "fi_batch_decode_plan",
    "fi_batch_decode_run",
    "fi_single_decode_run",
    "fi_merge_state",
    "fi_merge_state_in_place",
    "fi_merge_states",
    "fi_variable_length_merge_states",
    "fi_batch_prefill_plan",
    "fi_batch_prefill_plan_tile",
    "fi_batch_prefill_paged_run",
    "fi_single_prefill_run",
    "fi_gemm_fp8_nt_groupwise",
    "fi_group_gemm_fp8_nt_groupwise",
    "fi_get_batch_indices_positions",
    "fi_append_paged_kv_cache",
    "fi_apply_rope_pos_ids",
    "fi_apply_rope_append_paged_kv_cache",
    "fi_rope_positions_from_indptr",
    "fi_packbits",
    "fi_segment_packbits",
]


def lib() -> C.CDLL:
    """Load the library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} not found: build it with `make -C flashinfer-ai_amd/csrc -j8` "
            "(or python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback."
        )
    l = C.CDLL(_LIB_PATH)
    l.fi_last_error.restype = C.c_char_p
    l.fi_abi_version.restype = C.c_int
    l.fi_num_compute_units.restype = C.c_int
    vp, i32, i64p, sz = C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_size_t
    l.fi_batch_decode_plan.argtypes = [vp, sz, vp, vp, sz, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i64p, vp]
    l.fi_batch_decode_run.argtypes = [vp, sz, vp, sz, i64p, i32, C.POINTER(BatchDecodeParams), vp]
    l.fi_single_decode_run.argtypes = [C.POINTER(SingleDecodeParams), vp, sz, vp]
    l.fi_merge_state.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    l.fi_merge_state_in_place.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    l.fi_merge_states.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    l.fi_variable_length_merge_states.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    l.fi_batch_prefill_plan.argtypes = [vp, sz, vp, vp, sz, vp, vp, vp] + [i32] * 12 + [i64p, vp]
    l.fi_batch_prefill_plan_tile.argtypes = [vp, sz, vp, vp, sz, vp, vp, vp] + [i32] * 13 + [i64p, vp]
    l.fi_batch_prefill_paged_run.argtypes = [vp, sz, vp, sz, i64p, i32, C.POINTER(BatchPrefillParams), vp]
    l.fi_single_prefill_run.argtypes = [C.POINTER(SinglePrefillParams), vp, sz, vp]
    l.fi_gemm_fp8_nt_groupwise.argtypes = [vp] * 5 + [i32] * 10 + [vp]
    l.fi_group_gemm_fp8_nt_groupwise.argtypes = [vp] * 6 + [i32] * 11 + [vp]
    l.fi_packbits.argtypes = [vp, C.c_int64, i32, vp, vp]
    l.fi_segment_packbits.argtypes = [vp, vp, vp, i32, C.c_int64, i32, vp, vp]
    l.fi_get_batch_indices_positions.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    l.fi_append_paged_kv_cache.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, vp, vp, i32, C.POINTER(PagedKV), vp]
    l.fi_apply_rope_pos_ids.argtypes = [C.POINTER(RopeParams), vp]
    l.fi_apply_rope_append_paged_kv_cache.argtypes = [C.POINTER(RopeParams), vp, C.c_int64, C.c_int64, vp, vp, C.POINTER(PagedKV), vp]
    l.fi_rope_positions_from_indptr.argtypes = [vp, vp, i32, i32, vp, vp]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(l, name)
        if name not in ("fi_last_error",):
            fn.restype = C.c_int
    _lib = l
    return l


def check(status: int, what: str) -> None:
    """Turn a non-zero status into a Python exception carrying fi_last_error().
    (ref: TVM_FFI_ICHECK / FLASHINFER_ERROR surface as Python exceptions.)"""
    if status != 0:
        msg = lib().fi_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed: {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def current_stream(device: torch.device) -> int:
    """hipStream_t of torch's current stream on `device` (ref: csrc/tvm_ffi_utils.h:256-264)."""
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu_tensor(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} must live on a GPU (got {t.device}); the MI355X kernels have no CPU fallback"
        )

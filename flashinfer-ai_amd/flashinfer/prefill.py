"""Prefill / append attention operators: ``single_prefill_with_kv_cache`` and
``BatchPrefillWithPagedKVCacheWrapper`` (plan / run split).

Same names, arguments and defaults as the reference's ``flashinfer/prefill.py`` (single :960-1194;
wrapper :1226-2238).  The kernels are csrc/prefill_kernel.h (MFMA flash attention, GQA-packed tiles);
there is one backend, so ``backend`` accepts ``auto`` / ``fa2`` / ``fa3`` and means the same thing.
"""
from __future__ import annotations

import ctypes as C
import functools
import math
import os
from typing import Any, List, Optional, Tuple, Union

import torch

from . import _lib
from .page import get_seq_lens
from .quantization import packbits, segment_packbits
from .utils import (
    MaskMode,
    PosEncodingMode,
    _check_cached_qkv_data_type,
    _check_kv_layout,
    _check_pos_encoding_mode,
    _get_cache_alibi_slopes_buf,
    _get_cache_buf,
    _unpack_paged_kv_cache,
    canonicalize_torch_dtype,
    check_shape_dtype_device,
    is_float8,
    paged_kv_strides,
)


def _scale_tensor(x, n: int, device) -> Optional[torch.Tensor]:
    """per-head fp8 scale as a float32 device tensor of n entries (None stays None)."""
    if x is None:
        return None
    if not torch.is_tensor(x):
        # created on the device: no pageable host-to-device copy (which would sync, and fail under capture)
        return torch.full((n,), float(x), dtype=torch.float32, device=device)
    x = x.to(device=device, dtype=torch.float32).contiguous()
    if x.numel() == 1 and n != 1:
        x = x.expand(n).contiguous()
    if x.numel() != n:
        raise ValueError(f"scale tensor must have {n} entries, got {x.numel()}")
    return x


def _pick_cta_tile_q(q_dtype, kv_dtype, head_dim, pos_encoding_mode, logits_soft_cap, window_left, masked,
                     qo_indptr_host, group_size, num_kv_heads) -> int:
    """128, or 256 (fi_batch_prefill_plan_tile: the fp8-native kernel's 8-wave form) on request: FI_PREFILL_FP8_TILE=256
    and a plan that cannot leave that kernel's coverage -- e4m3 q/k/v, head_dim 128, plain logits, no window / mask.
    Measured at C3 the 256-row form is 4 % SLOWER (2.17 vs 2.08 ms: half the LDS-DMA instructions per query row, but
    eight waves per barrier and coarser causal diagonals), so it is not chosen by default."""
    if os.environ.get("FI_PREFILL_FP8_TILE") != "256":
        return 128
    if q_dtype != torch.float8_e4m3fn or kv_dtype != torch.float8_e4m3fn or head_dim != 128:
        return 128
    if pos_encoding_mode not in (None, "NONE") or (logits_soft_cap or 0) > 0 or (window_left is not None and window_left >= 0):
        return 128
    if masked or os.environ.get("FI_PREFILL_FP8_NATIVE", "1") == "0":
        return 128
    return 256


def _check_multi_item_args(prefix_len_ptr, token_pos_in_items_ptr, max_item_len_ptr, token_pos_in_items_len,
                           batch_size, device):
    """multi-item scoring operands (ref: flashinfer/prefill.py:1547-1558): uint32 prefix lengths, uint16 token
    positions (row stride token_pos_in_items_len), uint16 max item lengths (optional here: only used by the
    reference to skip masked tiles)."""
    if prefix_len_ptr is None and token_pos_in_items_ptr is None and max_item_len_ptr is None:
        return None, None, None, 0
    if prefix_len_ptr is None or token_pos_in_items_ptr is None:
        raise ValueError("multi-item scoring needs prefix_len_ptr and token_pos_in_items_ptr")
    if prefix_len_ptr.dtype != torch.uint32 or token_pos_in_items_ptr.dtype != torch.uint16:
        raise ValueError("prefix_len_ptr must be uint32 and token_pos_in_items_ptr uint16")
    if max_item_len_ptr is not None and max_item_len_ptr.dtype != torch.uint16:
        raise ValueError("max_item_len_ptr must be uint16")
    if prefix_len_ptr.numel() != batch_size:
        raise ValueError("prefix_len_ptr must have one entry per request")
    if token_pos_in_items_len <= 0 or token_pos_in_items_ptr.numel() < batch_size * token_pos_in_items_len:
        raise ValueError("token_pos_in_items_ptr is shorter than batch_size rows of token_pos_in_items_len")
    return (prefix_len_ptr.to(device).contiguous(), token_pos_in_items_ptr.to(device).contiguous(),
            None if max_item_len_ptr is None else max_item_len_ptr.to(device).contiguous(), int(token_pos_in_items_len))


def _check_multi_item_rows(prefix_len_ptr, token_pos_in_items_len, kv_lens_host):
    """The kernel reads token_pos_in_items[b][q_pos - prefix_len[b]] for every query position past the prefix, i.e. up
    to kv_len[b] - prefix_len[b] entries of row b: a shorter row would read the next request's row (or past the
    buffer) and the mask would be silently wrong.  plan() holds the kv lengths on the host; the prefix lengths come
    back with one small copy (plan() is synchronous host code anyway, ref: prefill.py:1547-1558)."""
    if prefix_len_ptr is None:
        return
    prefix = prefix_len_ptr.to("cpu").to(torch.int64)
    need = kv_lens_host.to(torch.int64) - prefix
    if bool((need > token_pos_in_items_len).any()):
        b = int(torch.nonzero(need > token_pos_in_items_len)[0])
        raise ValueError(
            f"multi-item scoring: request {b} has kv_len - prefix_len = {int(need[b])} positions past its prefix but "
            f"token_pos_in_items_len is {token_pos_in_items_len}")


def _mask_mode(wrapper) -> int:
    # ref: flashinfer/prefill.py:2091-2100
    if wrapper._custom_mask_buf is not None:
        return MaskMode.CUSTOM.value
    if getattr(wrapper, "_prefix_len_ptr", None) is not None:
        return MaskMode.MULTIITEMSCORING.value
    return MaskMode.CAUSAL.value if wrapper._causal else MaskMode.NON_CAUSAL.value


def _plan_custom_mask(wrapper, custom_mask, packed_custom_mask, qo_indptr_host, kv_lens_host, non_blocking):
    """(packed mask, byte indptr) on the wrapper's device, or (None, None).
    ref: _compute_page_mask_indptr + segment_packbits, flashinfer/prefill.py:1203-1223, 1693-1706; in
    CUDA-graph mode the caller-provided custom_mask_buf / mask_indptr_buf are filled (:1838-1857)."""
    if custom_mask is None and packed_custom_mask is None:
        return None, None
    qo_lens = (qo_indptr_host[1:] - qo_indptr_host[:-1]).to(torch.int64)
    bits = qo_lens * kv_lens_host.to(torch.int64)
    bit_indptr = torch.zeros(len(qo_indptr_host), dtype=torch.int64)
    bit_indptr[1:] = torch.cumsum(bits, 0)
    if int(((bits + 7) // 8).sum()) >= 2 ** 31:
        raise ValueError("custom mask too large: byte offsets must fit int32")
    if packed_custom_mask is None:
        custom_mask = custom_mask.to(wrapper.device).contiguous().view(-1)
        if custom_mask.numel() != int(bit_indptr[-1]):
            raise ValueError(
                f"custom_mask has {custom_mask.numel()} entries, expected sum(qo_len * kv_len) = {int(bit_indptr[-1])}")
        packed_custom_mask, mask_indptr = segment_packbits(
            custom_mask, bit_indptr.to(torch.int32).to(wrapper.device), bitorder="little")
    else:
        mask_indptr = torch.zeros(len(qo_indptr_host), dtype=torch.int32)
        mask_indptr[1:] = torch.cumsum((bits + 7) // 8, 0)
        mask_indptr = mask_indptr.to(wrapper.device, non_blocking=non_blocking)
        packed_custom_mask = packed_custom_mask.to(wrapper.device)
        if packed_custom_mask.dtype != torch.uint8:
            raise ValueError("packed_custom_mask must be uint8")
    if wrapper.is_cuda_graph_enabled:
        mask_buf = getattr(wrapper, "_user_custom_mask_buf", None)
        indptr_buf = getattr(wrapper, "_user_mask_indptr_buf", None)
        if mask_buf is None or indptr_buf is None:
            raise ValueError("custom_mask_buf and mask_indptr_buf are required for custom masks in cuda graph mode")
        if packed_custom_mask.numel() > mask_buf.numel():
            raise ValueError("packed custom mask exceeds custom_mask_buf")
        mask_buf[: packed_custom_mask.numel()].copy_(packed_custom_mask, non_blocking=non_blocking)
        indptr_buf[: len(mask_indptr)].copy_(mask_indptr, non_blocking=non_blocking)
        return mask_buf, indptr_buf
    return packed_custom_mask.contiguous(), mask_indptr.contiguous()


def single_prefill_with_kv_cache(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    scale_q: Optional[torch.Tensor] = None,
    scale_k: Optional[torch.Tensor] = None,
    scale_v: Optional[torch.Tensor] = None,
    o_dtype: Optional[torch.dtype] = None,
    custom_mask: Optional[torch.Tensor] = None,
    packed_custom_mask: Optional[torch.Tensor] = None,
    causal: bool = False,
    kv_layout: str = "NHD",
    pos_encoding_mode: str = "NONE",
    use_fp16_qk_reduction: bool = False,
    sm_scale: Optional[float] = None,
    window_left: int = -1,
    logits_soft_cap: Optional[float] = None,
    rope_scale: Optional[float] = None,
    rope_theta: Optional[float] = None,
    backend: str = "auto",
    return_lse: bool = False,
    bf16_pv_exact_range: bool = False,
) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
    r"""Prefill / append attention with KV cache for a single request.

    Parameters
    ----------
    q : ``[qo_len, num_qo_heads, head_dim]``
    k, v : ``[kv_len, num_kv_heads, head_dim]`` (``NHD``) or ``[num_kv_heads, kv_len, head_dim]`` (``HND``)
    scale_q, scale_k, scale_v : per-head scales (``[num_qo_heads]`` / ``[num_kv_heads]``) for fp8 inputs
    o_dtype : output dtype (required for fp8 attention; defaults to ``q.dtype``)
    causal : apply the causal mask (query i sees keys up to ``i + kv_len - qo_len``)
    pos_encoding_mode : ``NONE`` / ``ROPE_LLAMA`` (applied in-kernel) / ``ALIBI``
    sm_scale, window_left, logits_soft_cap, rope_scale, rope_theta : as the reference
    return_lse : also return the base-2 logsumexp, shape ``[qo_len, num_qo_heads]``
    bf16_pv_exact_range : (extension, bf16 queries) ``True`` when ``v`` may hold ``|v| >= 65504`` or many
        ``|v| < 6e-5``: the kernel then keeps V in bf16 and enters P as hi + lo bf16 halves (about 25 % slower)
        instead of running P.V on the f16 matrix cores, whose V operand is exact only inside f16's normal range

    custom_mask : ``[qo_len, kv_len]`` bool; packed_custom_mask : its ``packbits(..., bitorder="little")``
        form (takes precedence).  With a mask, ``causal`` is ignored (mask mode CUSTOM).
    (ref: flashinfer/prefill.py:960-1194)
    """
    _check_pos_encoding_mode(pos_encoding_mode)
    _check_kv_layout(kv_layout)
    if custom_mask is not None and packed_custom_mask is None:
        # ref: prefill.py:1114-1118
        packed_custom_mask = packbits(custom_mask.contiguous().view(-1), bitorder="little")
    for t, name in ((q, "q"), (k, "k"), (v, "v")):
        _lib.require_gpu_tensor(t, name)
    if q.dim() != 3 or k.dim() != 3 or k.shape != v.shape:
        raise ValueError("q must be [qo_len, num_qo_heads, head_dim]; k, v 3-D with equal shapes")
    if logits_soft_cap is None:
        logits_soft_cap = 0.0
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(q.size(-1))
    if rope_scale is None:
        rope_scale = 1.0
    if rope_theta is None:
        rope_theta = 1e4
    qo_len, num_qo_heads, head_dim = q.shape
    if kv_layout == "NHD":
        kv_len, num_kv_heads = k.shape[0], k.shape[1]
        stride_n, stride_h = k.stride(0), k.stride(1)
    else:
        num_kv_heads, kv_len = k.shape[0], k.shape[1]
        stride_h, stride_n = k.stride(0), k.stride(1)
    if k.stride() != v.stride() or k.stride(-1) != 1 or q.stride(-1) != 1:
        raise ValueError("k and v must share strides and q/k/v must be contiguous in head_dim")
    if is_float8(q):
        assert window_left == -1
        assert q.dtype == k.dtype == v.dtype
        # a missing scale is passed as NULL (= 1, include/fi_mi355.h): no tensor is created
        scale_q = _scale_tensor(scale_q, num_qo_heads, q.device)
        scale_k = _scale_tensor(scale_k, num_kv_heads, q.device)
        scale_v = _scale_tensor(scale_v, num_kv_heads, q.device)
        if o_dtype is None:
            raise ValueError("o_dtype should be provided for FP8 attention")
    else:
        scale_q = scale_k = scale_v = None
    if o_dtype is None:
        o_dtype = q.dtype
    out = torch.empty(q.shape[:-1] + v.shape[-1:], dtype=o_dtype, device=q.device)
    lse = None
    if return_lse:
        lse = torch.empty((qo_len, num_qo_heads), dtype=torch.float32, device=q.device)
    if packed_custom_mask is not None:
        _lib.require_gpu_tensor(packed_custom_mask, "packed_custom_mask")
        if packed_custom_mask.dtype != torch.uint8 or packed_custom_mask.numel() * 8 < qo_len * kv_len:
            raise ValueError("packed_custom_mask must be uint8 with at least qo_len * kv_len bits")
        packed_custom_mask = packed_custom_mask.contiguous()
    alibi = _get_cache_alibi_slopes_buf(num_qo_heads, q.device) if pos_encoding_mode == "ALIBI" else None
    params = _lib.SinglePrefillParams(
        q=q.data_ptr(), q_stride_n=q.stride(0), q_stride_h=q.stride(1), k=k.data_ptr(), v=v.data_ptr(),
        kv_stride_n=stride_n, kv_stride_h=stride_h, o=out.data_ptr(), lse=_lib.ptr(lse),
        alibi_slopes=_lib.ptr(alibi), scale_q=_lib.ptr(scale_q), scale_k=_lib.ptr(scale_k),
        scale_v=_lib.ptr(scale_v), qo_len=qo_len, kv_len=kv_len, num_qo_heads=num_qo_heads,
        num_kv_heads=num_kv_heads, head_dim=head_dim, q_dtype=_lib.fi_dtype(q.dtype),
        kv_dtype=_lib.fi_dtype(k.dtype), o_dtype=_lib.fi_dtype(o_dtype),
        custom_mask=_lib.ptr(packed_custom_mask),
        mask_mode=(MaskMode.CUSTOM.value if packed_custom_mask is not None
                   else MaskMode.CAUSAL.value if causal else MaskMode.NON_CAUSAL.value),
        pos_encoding_mode=PosEncodingMode[pos_encoding_mode].value, window_left=window_left,
        logits_soft_cap=logits_soft_cap, sm_scale=sm_scale, rope_rcp_scale=1.0 / rope_scale,
        rope_rcp_theta=1.0 / rope_theta, bf16_pv_mode=1 if bf16_pv_exact_range else 0,
    )
    # scratch for split-KV partial states (ref: the 32 MB cached buffer of single_prefill, prefill.py:1125)
    tmp = _get_cache_buf("single_prefill_with_kv_cache_tmp", 32 * 1024 * 1024, q.device)
    with torch.cuda.device(q.device):
        _lib.check(
            _lib.lib().fi_single_prefill_run(C.byref(params), tmp.data_ptr(), tmp.numel() * tmp.element_size(),
                                             _lib.current_stream(q.device)),
            "single_prefill_with_kv_cache",
        )
    return (out, lse) if return_lse else out


single_prefill_with_kv_cache_return_lse = functools.partial(
    single_prefill_with_kv_cache, return_lse=True
)


class BatchPrefillWithPagedKVCacheWrapper:
    r"""Prefill / append attention over a paged KV cache for a batch of requests.

    >>> prefill_wrapper = flashinfer.BatchPrefillWithPagedKVCacheWrapper(workspace_buffer, "NHD")
    >>> prefill_wrapper.plan(qo_indptr, paged_kv_indptr, paged_kv_indices, paged_kv_last_page_len,
    ...                      num_qo_heads, num_kv_heads, head_dim, page_size, causal=True)
    >>> o = prefill_wrapper.run(q, kv_cache)          # [qo_indptr[-1], num_qo_heads, head_dim]

    (ref: flashinfer/prefill.py:1226-2238; example page table :1247-1259)
    """

    def __init__(
        self,
        float_workspace_buffer: torch.Tensor,
        kv_layout: str = "NHD",
        use_cuda_graph: bool = False,
        qo_indptr_buf: Optional[torch.Tensor] = None,
        paged_kv_indptr_buf: Optional[torch.Tensor] = None,
        paged_kv_indices_buf: Optional[torch.Tensor] = None,
        paged_kv_last_page_len_buf: Optional[torch.Tensor] = None,
        custom_mask_buf: Optional[torch.Tensor] = None,
        mask_indptr_buf: Optional[torch.Tensor] = None,
        backend: str = "auto",
        jit_args: Optional[List[Any]] = None,
        jit_kwargs: Optional[dict] = None,
    ) -> None:
        _check_kv_layout(kv_layout)
        if jit_args is not None:
            raise ValueError("jit_args is not supported: kernels are built ahead of time")
        if backend not in ("auto", "fa2", "fa3"):
            raise ValueError(f"backend {backend!r} is not available on MI355X (use 'auto')")
        _lib.require_gpu_tensor(float_workspace_buffer, "float_workspace_buffer")
        self._kv_layout = kv_layout
        self._float_workspace_buffer = float_workspace_buffer
        self.device = float_workspace_buffer.device
        self._int_workspace_buffer = torch.empty((8 * 1024 * 1024,), dtype=torch.uint8, device=self.device)
        self._pin_memory_int_workspace_buffer = torch.empty(
            self._int_workspace_buffer.shape, dtype=torch.uint8, pin_memory=True, device="cpu"
        )
        self._use_cuda_graph = use_cuda_graph
        if use_cuda_graph:
            for buf, name in ((qo_indptr_buf, "qo_indptr_buf"), (paged_kv_indptr_buf, "paged_kv_indptr_buf"),
                              (paged_kv_indices_buf, "paged_kv_indices_buf"),
                              (paged_kv_last_page_len_buf, "paged_kv_last_page_len_buf")):
                if not torch.is_tensor(buf):
                    raise ValueError(f"{name} should be a torch.Tensor in CUDA graph mode")
            self._fixed_batch_size = len(qo_indptr_buf) - 1
            if len(paged_kv_indptr_buf) != self._fixed_batch_size + 1:
                raise ValueError("The length of paged_kv_indptr_buf should be batch_size + 1.")
            if len(paged_kv_last_page_len_buf) != self._fixed_batch_size:
                raise ValueError("The length of paged_kv_last_page_len_buf should be batch_size.")
        else:
            self._fixed_batch_size = 0
        self._qo_indptr_buf = qo_indptr_buf
        self._paged_kv_indptr_buf = paged_kv_indptr_buf
        self._paged_kv_indices_buf = paged_kv_indices_buf
        self._paged_kv_last_page_len_buf = paged_kv_last_page_len_buf
        self._user_custom_mask_buf = custom_mask_buf
        self._user_mask_indptr_buf = mask_indptr_buf
        self._custom_mask_buf = self._mask_indptr_buf = None
        self._backend = backend
        self._plan_info = None

    @property
    def is_cuda_graph_enabled(self) -> bool:
        return self._use_cuda_graph

    def reset_workspace_buffer(
        self, float_workspace_buffer: torch.Tensor, int_workspace_buffer: torch.Tensor
    ) -> None:
        self._float_workspace_buffer = float_workspace_buffer
        self._int_workspace_buffer = int_workspace_buffer
        self._pin_memory_int_workspace_buffer = torch.empty(
            self._int_workspace_buffer.shape, dtype=self._int_workspace_buffer.dtype, device="cpu",
            pin_memory=True,
        )

    def plan(
        self,
        qo_indptr: torch.Tensor,
        paged_kv_indptr: torch.Tensor,
        paged_kv_indices: torch.Tensor,
        paged_kv_last_page_len: torch.Tensor,
        num_qo_heads: int,
        num_kv_heads: int,
        head_dim_qk: int,
        page_size: int,
        head_dim_vo: Optional[int] = None,
        custom_mask: Optional[torch.Tensor] = None,
        packed_custom_mask: Optional[torch.Tensor] = None,
        causal: bool = False,
        pos_encoding_mode: str = "NONE",
        use_fp16_qk_reduction: bool = False,
        sm_scale: Optional[float] = None,
        window_left: int = -1,
        logits_soft_cap: Optional[float] = None,
        rope_scale: Optional[float] = None,
        rope_theta: Optional[float] = None,
        q_data_type: Union[str, torch.dtype] = "float16",
        kv_data_type: Optional[Union[str, torch.dtype]] = None,
        non_blocking: bool = True,
        prefix_len_ptr: Optional[torch.Tensor] = None,
        token_pos_in_items_ptr: Optional[torch.Tensor] = None,
        token_pos_in_items_len: int = 0,
        max_item_len_ptr: Optional[torch.Tensor] = None,
        seq_lens: Optional[torch.Tensor] = None,
        seq_lens_q: Optional[torch.Tensor] = None,
        block_tables: Optional[torch.Tensor] = None,
        max_token_per_sequence: Optional[int] = None,
        max_sequence_kv: Optional[int] = None,
        fixed_split_size: Optional[int] = None,
        disable_split_kv: bool = False,
        o_data_type: Optional[Union[str, torch.dtype]] = None,
        bf16_pv_exact_range: bool = False,
    ) -> None:
        r"""Plan batch prefill/append attention for the given ragged queries and page table.

        qo_indptr : ``[batch_size + 1]`` int32; paged_kv_indptr / indices / last_page_len as in decode.
        causal, pos_encoding_mode, sm_scale, window_left, logits_soft_cap, rope_* configure the variant.
        q_data_type / kv_data_type : dtypes the run() tensors will have (fp8 e4m3 for both = fp8 attention).
        o_data_type : (extension) output dtype; defaults to the q dtype, or bfloat16 for fp8 queries.
        bf16_pv_exact_range : (extension, bf16 queries) the cache may hold ``|v| >= 65504`` or many ``|v| < 6e-5``:
            P.V then runs with hi + lo bf16 probabilities instead of on the f16 matrix cores (no range limit on V,
            about 25 % slower); see :func:`single_prefill_with_kv_cache`.
        custom_mask : flattened bool mask, request i contributes ``qo_len[i] * kv_len[i]`` entries
            (row-major ``[qo_len, kv_len]``); packed_custom_mask : its ``segment_packbits(..., "little")`` form.
            With a mask the mask mode is CUSTOM and ``causal`` is ignored (ref: prefill.py:1693-1706, 1890-1905).
        prefix_len_ptr (uint32 ``[batch]``), token_pos_in_items_ptr (uint16 ``[batch * token_pos_in_items_len]``),
        token_pos_in_items_len, max_item_len_ptr (uint16 ``[batch]``): multi-item scoring -- a query past the
        request's prefix sees the prefix and the tokens of its own item (mask mode MULTIITEMSCORING; plan with
        ``causal=True``; ref: prefill.py:1547-1558, 2099-2100, prefill.cuh:795-858).
        (ref: flashinfer/prefill.py:1523-1921)
        """
        self._bf16_pv_mode = 1 if bf16_pv_exact_range else 0
        self._prefix_len_ptr, self._token_pos_in_items_ptr, self._max_item_len_ptr, self._token_pos_in_items_len = \
            _check_multi_item_args(prefix_len_ptr, token_pos_in_items_ptr, max_item_len_ptr, token_pos_in_items_len,
                                   len(qo_indptr) - 1, self.device)
        for tensor, name in [(qo_indptr, "qo_indptr"), (paged_kv_indptr, "paged_kv_indptr"),
                             (paged_kv_indices, "paged_kv_indices"),
                             (paged_kv_last_page_len, "paged_kv_last_page_len")]:
            if tensor.dtype != torch.int32:
                raise ValueError(f"{name} must have dtype torch.int32, got {tensor.dtype}")
        _check_pos_encoding_mode(pos_encoding_mode)
        q_data_type = canonicalize_torch_dtype(q_data_type)
        if kv_data_type is None:
            kv_data_type = q_data_type
        kv_data_type = canonicalize_torch_dtype(kv_data_type)
        if o_data_type is None:
            o_data_type = torch.bfloat16 if q_data_type in (torch.float8_e4m3fn, torch.float8_e5m2) else q_data_type
        o_data_type = canonicalize_torch_dtype(o_data_type)
        if logits_soft_cap is None:
            logits_soft_cap = 0.0
        if head_dim_vo is None:
            head_dim_vo = head_dim_qk
        batch_size = len(qo_indptr) - 1
        if len(paged_kv_indptr) != batch_size + 1 or len(paged_kv_last_page_len) != batch_size:
            raise ValueError("qo_indptr, paged_kv_indptr and paged_kv_last_page_len disagree on the batch size")

        qo_indptr_host = qo_indptr.to("cpu").contiguous()
        paged_kv_indptr_host = paged_kv_indptr.to("cpu").contiguous()
        paged_kv_last_page_len_host = paged_kv_last_page_len.to("cpu")
        if seq_lens is None:
            kv_lens_arr_host = get_seq_lens(paged_kv_indptr_host, paged_kv_last_page_len_host, page_size)
        else:
            kv_lens_arr_host = seq_lens.cpu()
        kv_lens_arr_host = kv_lens_arr_host.to(torch.int32).contiguous()
        _check_multi_item_rows(self._prefix_len_ptr, self._token_pos_in_items_len, kv_lens_arr_host)
        total_num_rows = int(qo_indptr_host[-1])
        self._custom_mask_buf, self._mask_indptr_buf = _plan_custom_mask(
            self, custom_mask, packed_custom_mask, qo_indptr_host, kv_lens_arr_host, non_blocking)
        if self._custom_mask_buf is not None:
            causal = False  # mask mode CUSTOM: every kv tile is visited, the bits decide

        if self.is_cuda_graph_enabled:
            if batch_size != self._fixed_batch_size:
                raise ValueError(
                    "The batch size should be fixed during the lifecycle of the wrapper in cuda graph mode, "
                    f"the runtime batch size {batch_size} mismatches the batch size {self._fixed_batch_size}"
                )
            if len(paged_kv_indices) > len(self._paged_kv_indices_buf):
                raise ValueError("The length of paged_kv_indices exceeds the allocated buffer size.")
            self._qo_indptr_buf.copy_(qo_indptr, non_blocking=non_blocking)
            self._paged_kv_indptr_buf.copy_(paged_kv_indptr, non_blocking=non_blocking)
            self._paged_kv_last_page_len_buf.copy_(paged_kv_last_page_len, non_blocking=non_blocking)
            self._paged_kv_indices_buf[: len(paged_kv_indices)].copy_(
                paged_kv_indices, non_blocking=(paged_kv_indices.device == self.device) and non_blocking
            )
            if max_token_per_sequence is None:
                total_rows_bound = total_num_rows
            else:
                total_rows_bound = max_token_per_sequence * batch_size
        else:
            self._qo_indptr_buf = qo_indptr.to(self.device, non_blocking=non_blocking)
            self._paged_kv_indptr_buf = paged_kv_indptr.to(self.device, non_blocking=non_blocking)
            self._paged_kv_indices_buf = paged_kv_indices.to(self.device, non_blocking=non_blocking)
            self._paged_kv_last_page_len_buf = paged_kv_last_page_len.to(self.device, non_blocking=non_blocking)
            total_rows_bound = total_num_rows

        # q tile: 128 packed rows; 256 (the fp8-native kernel's 8-wave form) only on request, see _pick_cta_tile_q
        cta_tile_q = _pick_cta_tile_q(
            q_data_type, kv_data_type, head_dim_qk, pos_encoding_mode, logits_soft_cap, window_left,
            self._custom_mask_buf is not None or prefix_len_ptr is not None, qo_indptr_host,
            num_qo_heads // num_kv_heads, num_kv_heads)
        plan_info = (C.c_int64 * _lib.FI_PREFILL_PLAN_INFO_LEN)()
        with torch.cuda.device(self.device):
            _lib.check(
                _lib.lib().fi_batch_prefill_plan_tile(
                    self._float_workspace_buffer.data_ptr(),
                    self._float_workspace_buffer.numel() * self._float_workspace_buffer.element_size(),
                    self._int_workspace_buffer.data_ptr(),
                    self._pin_memory_int_workspace_buffer.data_ptr(),
                    self._int_workspace_buffer.numel(),
                    qo_indptr_host.data_ptr(), paged_kv_indptr_host.data_ptr(), kv_lens_arr_host.data_ptr(),
                    total_rows_bound, batch_size, num_qo_heads, num_kv_heads, page_size,
                    int(self.is_cuda_graph_enabled), head_dim_qk, head_dim_vo, int(causal), window_left,
                    -1 if fixed_split_size is None else fixed_split_size, int(disable_split_kv), cta_tile_q,
                    plan_info, _lib.current_stream(self.device),
                ),
                "BatchPrefillWithPagedKVCacheWrapper.plan",
            )
        self._plan_info = list(plan_info)
        self._plan_info_c = plan_info
        self._batch_size = batch_size
        self._num_qo_heads = num_qo_heads
        self._num_kv_heads = num_kv_heads
        self._head_dim = head_dim_qk
        self._page_size = page_size
        self._total_num_rows = total_num_rows
        self._cached_q_data_type = q_data_type
        self._cached_kv_data_type = kv_data_type
        self._cached_o_data_type = o_data_type
        self._causal = causal
        self._pos_encoding_mode = pos_encoding_mode
        self._window_left = window_left
        self._logits_soft_cap = logits_soft_cap
        self._sm_scale = sm_scale
        self._rope_scale = rope_scale
        self._rope_theta = rope_theta

    begin_forward = plan

    def forward(self, q, paged_kv_cache, causal=False, pos_encoding_mode="NONE", use_fp16_qk_reduction=False,
                k_scale=None, v_scale=None, window_left=-1, logits_soft_cap=None, sm_scale=None,
                rope_scale=None, rope_theta=None) -> torch.Tensor:
        r"""Warning: This function is deprecated, please use :meth:`run` instead."""
        self._causal = causal
        self._pos_encoding_mode = pos_encoding_mode
        self._window_left = window_left
        self._logits_soft_cap = logits_soft_cap
        self._sm_scale = sm_scale
        self._rope_scale = rope_scale
        self._rope_theta = rope_theta
        return self.run(q, paged_kv_cache, k_scale=k_scale, v_scale=v_scale)

    def run(
        self,
        q: torch.Tensor,
        paged_kv_cache: Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]],
        *args,
        q_scale: Optional[float] = None,
        k_scale: Optional[float] = None,
        v_scale: Optional[float] = None,
        out: Optional[torch.Tensor] = None,
        lse: Optional[torch.Tensor] = None,
        return_lse: bool = False,
        enable_pdl: Optional[bool] = None,
        window_left: Optional[int] = None,
        sinks: Optional[torch.Tensor] = None,
        scale_q: Optional[torch.Tensor] = None,
        scale_k: Optional[torch.Tensor] = None,
        scale_v: Optional[torch.Tensor] = None,
    ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        r"""Compute batch prefill/append attention between ``q`` and the paged KV cache.

        q : ``[qo_indptr[-1], num_qo_heads, head_dim]``; paged_kv_cache as in decode.
        q_scale / k_scale / v_scale : scalar calibration scales (folded into sm_scale / the output).
        scale_q / scale_k / scale_v : (extension) per-head fp8 scales ``[num_qo_heads]`` / ``[num_kv_heads]``
            for fp8 attention (the reference's FA3 kernel takes them, csrc/batch_prefill_fp8_sm90.cu:81-185,
            but its wrapper passes None).
        Returns ``[qo_indptr[-1], num_qo_heads, head_dim]`` (+ base-2 logsumexp ``[nnz, num_qo_heads]``).
        (ref: flashinfer/prefill.py:1979-2206)
        """
        if self._plan_info is None:
            raise RuntimeError("plan() must be called before run()")
        if sinks is not None:
            raise ValueError("attention sinks are not supported by this backend")
        if args:
            raise ValueError("additional kernel arguments require jit_args, which is not supported")
        _lib.require_gpu_tensor(q, "q")
        k_cache, v_cache = _unpack_paged_kv_cache(paged_kv_cache, self._kv_layout)
        _check_cached_qkv_data_type(q, k_cache, self._cached_q_data_type, self._cached_kv_data_type)
        page_size, num_kv_heads, head_dim, stride_page, stride_n, stride_h = paged_kv_strides(
            k_cache, v_cache, self._kv_layout
        )
        window_left = self._window_left if window_left is None else window_left
        assert window_left == self._window_left
        logits_soft_cap = self._logits_soft_cap
        sm_scale = self._sm_scale
        rope_scale = self._rope_scale
        rope_theta = self._rope_theta
        if logits_soft_cap is None:
            logits_soft_cap = 0.0
        if sm_scale is None:
            sm_scale = 1.0 / math.sqrt(q.size(-1))
        if q_scale is not None:
            sm_scale *= q_scale
        if k_scale is not None:
            sm_scale *= k_scale
        if rope_scale is None:
            rope_scale = 1.0
        if rope_theta is None:
            rope_theta = 1e4
        if q.dim() != 3 or q.shape[0] != self._total_num_rows or q.shape[1] != self._num_qo_heads:
            raise ValueError(
                f"q must have shape [{self._total_num_rows}, {self._num_qo_heads}, head_dim], got {tuple(q.shape)}"
            )
        if q.shape[2] != head_dim or head_dim != self._head_dim:
            raise ValueError("head_dim of q / kv cache does not match the planned head_dim")
        if num_kv_heads != self._num_kv_heads or page_size != self._page_size:
            raise ValueError("kv cache shape does not match the planned num_kv_heads / page_size")
        if q.stride(-1) != 1:
            q = q.contiguous()
        o_dtype = self._cached_o_data_type
        if return_lse:
            if lse is None:
                lse = torch.empty((q.size(0), q.size(1)), dtype=torch.float32, device=q.device)
            else:
                check_shape_dtype_device(lse, (q.size(0), q.size(1)), torch.float32, q.device, "lse")
        out_shape = q.shape[:-1] + v_cache.shape[-1:]
        if out is None:
            out = torch.empty(out_shape, dtype=o_dtype, device=q.device)
        else:
            check_shape_dtype_device(out, out_shape, o_dtype, q.device, "out")
            if not out.is_contiguous():
                raise ValueError("out must be contiguous")
        # missing scales are NULL pointers (= 1 in the kernels): run() creates no tensor and stays capturable
        scale_q = _scale_tensor(scale_q, self._num_qo_heads, q.device)
        scale_k = _scale_tensor(scale_k, num_kv_heads, q.device)
        scale_v = _scale_tensor(scale_v, num_kv_heads, q.device)
        alibi = None
        if self._pos_encoding_mode == "ALIBI":
            alibi = _get_cache_alibi_slopes_buf(q.shape[1], q.device)
        params = _lib.BatchPrefillParams(
            q=q.data_ptr(), q_stride_n=q.stride(0), q_stride_h=q.stride(1),
            qo_indptr=self._qo_indptr_buf.data_ptr(),
            kv=_lib.PagedKV(
                k_data=k_cache.data_ptr(), v_data=v_cache.data_ptr(),
                indptr=self._paged_kv_indptr_buf.data_ptr(), indices=self._paged_kv_indices_buf.data_ptr(),
                last_page_len=self._paged_kv_last_page_len_buf.data_ptr(), rope_pos_offset=None,
                stride_page=stride_page, stride_n=stride_n, stride_h=stride_h, page_size=page_size,
                num_kv_heads=num_kv_heads, head_dim=head_dim, batch_size=self._batch_size,
                dtype=_lib.fi_dtype(k_cache.dtype),
            ),
            o=out.data_ptr(), lse=_lib.ptr(lse) if return_lse else None, alibi_slopes=_lib.ptr(alibi),
            scale_q=_lib.ptr(scale_q), scale_k=_lib.ptr(scale_k), scale_v=_lib.ptr(scale_v),
            num_qo_heads=self._num_qo_heads, q_dtype=_lib.fi_dtype(q.dtype), o_dtype=_lib.fi_dtype(o_dtype),
            custom_mask=_lib.ptr(self._custom_mask_buf), mask_indptr=_lib.ptr(self._mask_indptr_buf),
            prefix_len_ptr=_lib.ptr(self._prefix_len_ptr), token_pos_in_items_ptr=_lib.ptr(self._token_pos_in_items_ptr),
            max_item_len_ptr=_lib.ptr(self._max_item_len_ptr), token_pos_in_items_len=self._token_pos_in_items_len,
            mask_mode=_mask_mode(self),
            pos_encoding_mode=PosEncodingMode[self._pos_encoding_mode].value, window_left=window_left,
            logits_soft_cap=logits_soft_cap, sm_scale=sm_scale, rope_rcp_scale=1.0 / rope_scale,
            rope_rcp_theta=1.0 / rope_theta, bf16_pv_mode=getattr(self, "_bf16_pv_mode", 0),
        )
        with torch.cuda.device(q.device):
            _lib.check(
                _lib.lib().fi_batch_prefill_paged_run(
                    self._float_workspace_buffer.data_ptr(),
                    self._float_workspace_buffer.numel() * self._float_workspace_buffer.element_size(),
                    self._int_workspace_buffer.data_ptr(), self._int_workspace_buffer.numel(),
                    self._plan_info_c, _lib.FI_PREFILL_PLAN_INFO_LEN, C.byref(params),
                    _lib.current_stream(q.device),
                ),
                "BatchPrefillWithPagedKVCacheWrapper.run",
            )
        if v_scale is not None:
            if is_float8(out):
                out = (out.to(torch.float32) * v_scale).to(out.dtype)
            else:
                out *= v_scale
        return (out, lse) if return_lse else out

    run_return_lse = functools.partialmethod(run, return_lse=True)

    def forward_return_lse(self, q, paged_kv_cache, causal=False, pos_encoding_mode="NONE",
                           use_fp16_qk_reduction=False, k_scale=None, v_scale=None, window_left=-1,
                           logits_soft_cap=None, sm_scale=None, rope_scale=None, rope_theta=None):
        r"""Warning: This function is deprecated, please use :meth:`run_return_lse` instead."""
        self._causal = causal
        self._pos_encoding_mode = pos_encoding_mode
        self._window_left = window_left
        self._logits_soft_cap = logits_soft_cap
        self._sm_scale = sm_scale
        self._rope_scale = rope_scale
        self._rope_theta = rope_theta
        return self.run_return_lse(q, paged_kv_cache, k_scale=k_scale, v_scale=v_scale)

    def end_forward(self) -> None:
        r"""Warning: this function is deprecated and has no effect."""
        pass


class BatchPrefillWithRaggedKVCacheWrapper:
    r"""Prefill / append attention with ragged (tensor) KV for a batch of requests: ``k``/``v`` are
    ``[kv_indptr[-1], num_kv_heads, head_dim]`` (``NHD``) or ``[num_kv_heads, kv_indptr[-1], head_dim]`` (``HND``).

    Runs the paged kernels with an identity page table of one-token pages (see include/fi_mi355.h).
    (ref: flashinfer/prefill.py:2255-3007)
    """

    def __init__(
        self,
        float_workspace_buffer: torch.Tensor,
        kv_layout: str = "NHD",
        use_cuda_graph: bool = False,
        qo_indptr_buf: Optional[torch.Tensor] = None,
        kv_indptr_buf: Optional[torch.Tensor] = None,
        custom_mask_buf: Optional[torch.Tensor] = None,
        mask_indptr_buf: Optional[torch.Tensor] = None,
        backend: str = "auto",
        jit_args: Optional[List[Any]] = None,
        jit_kwargs: Optional[dict] = None,
    ) -> None:
        _check_kv_layout(kv_layout)
        if jit_args is not None:
            raise ValueError("jit_args is not supported: kernels are built ahead of time")
        if backend not in ("auto", "fa2", "fa3"):
            raise ValueError(f"backend {backend!r} is not available on MI355X (use 'auto')")
        _lib.require_gpu_tensor(float_workspace_buffer, "float_workspace_buffer")
        self._kv_layout = kv_layout
        self._float_workspace_buffer = float_workspace_buffer
        self.device = float_workspace_buffer.device
        self._int_workspace_buffer = torch.empty((8 * 1024 * 1024,), dtype=torch.uint8, device=self.device)
        self._pin_memory_int_workspace_buffer = torch.empty(
            self._int_workspace_buffer.shape, dtype=torch.uint8, pin_memory=True, device="cpu"
        )
        self._use_cuda_graph = use_cuda_graph
        if use_cuda_graph:
            if not torch.is_tensor(qo_indptr_buf) or not torch.is_tensor(kv_indptr_buf):
                raise ValueError("qo_indptr_buf and kv_indptr_buf should be torch.Tensor in cuda graph mode")
            self._fixed_batch_size = len(qo_indptr_buf) - 1
            if len(kv_indptr_buf) != self._fixed_batch_size + 1:
                raise ValueError("The length of kv_indptr_buf should be batch_size + 1.")
        else:
            self._fixed_batch_size = 0
        self._qo_indptr_buf = qo_indptr_buf
        self._kv_indptr_buf = kv_indptr_buf
        self._user_custom_mask_buf = custom_mask_buf
        self._user_mask_indptr_buf = mask_indptr_buf
        self._custom_mask_buf = self._mask_indptr_buf = None
        self._plan_info = None

    @property
    def is_cuda_graph_enabled(self) -> bool:
        return self._use_cuda_graph

    def reset_workspace_buffer(self, float_workspace_buffer: torch.Tensor, int_workspace_buffer) -> None:
        self._float_workspace_buffer = float_workspace_buffer
        self._int_workspace_buffer = int_workspace_buffer
        self._pin_memory_int_workspace_buffer = torch.empty(
            self._int_workspace_buffer.shape, dtype=self._int_workspace_buffer.dtype, device="cpu", pin_memory=True
        )

    def plan(
        self,
        qo_indptr: torch.Tensor,
        kv_indptr: torch.Tensor,
        num_qo_heads: int,
        num_kv_heads: int,
        head_dim_qk: int,
        head_dim_vo: Optional[int] = None,
        custom_mask: Optional[torch.Tensor] = None,
        packed_custom_mask: Optional[torch.Tensor] = None,
        causal: bool = False,
        pos_encoding_mode: str = "NONE",
        use_fp16_qk_reduction: bool = False,
        window_left: int = -1,
        logits_soft_cap: Optional[float] = None,
        sm_scale: Optional[float] = None,
        rope_scale: Optional[float] = None,
        rope_theta: Optional[float] = None,
        q_data_type: Union[str, torch.dtype] = "float16",
        kv_data_type: Optional[Union[str, torch.dtype]] = None,
        non_blocking: bool = True,
        prefix_len_ptr: Optional[torch.Tensor] = None,
        token_pos_in_items_ptr: Optional[torch.Tensor] = None,
        token_pos_in_items_len: int = 0,
        max_item_len_ptr: Optional[torch.Tensor] = None,
        fixed_split_size: Optional[int] = None,
        disable_split_kv: bool = False,
        bf16_pv_exact_range: bool = False,
    ) -> None:
        r"""Plan for ragged queries ``qo_indptr`` and ragged keys/values ``kv_indptr`` (both int32
        ``[batch_size + 1]``).  Options as :meth:`BatchPrefillWithPagedKVCacheWrapper.plan`."""
        self._bf16_pv_mode = 1 if bf16_pv_exact_range else 0
        self._prefix_len_ptr, self._token_pos_in_items_ptr, self._max_item_len_ptr, self._token_pos_in_items_len = \
            _check_multi_item_args(prefix_len_ptr, token_pos_in_items_ptr, max_item_len_ptr, token_pos_in_items_len,
                                   len(qo_indptr) - 1, self.device)
        for tensor, name in [(qo_indptr, "qo_indptr"), (kv_indptr, "kv_indptr")]:
            if tensor.dtype != torch.int32:
                raise ValueError(f"{name} must have dtype torch.int32, got {tensor.dtype}")
        _check_pos_encoding_mode(pos_encoding_mode)
        q_data_type = canonicalize_torch_dtype(q_data_type)
        if kv_data_type is None:
            kv_data_type = q_data_type
        kv_data_type = canonicalize_torch_dtype(kv_data_type)
        if logits_soft_cap is None:
            logits_soft_cap = 0.0
        if head_dim_vo is None:
            head_dim_vo = head_dim_qk
        batch_size = len(qo_indptr) - 1
        if len(kv_indptr) != batch_size + 1:
            raise ValueError("The kv_indptr length should be equal to qo_indptr length.")
        qo_indptr_host = qo_indptr.to("cpu").contiguous()
        kv_indptr_host = kv_indptr.to("cpu").contiguous()
        kv_len_arr = (kv_indptr_host[1:] - kv_indptr_host[:-1]).to(torch.int32).contiguous()
        _check_multi_item_rows(self._prefix_len_ptr, self._token_pos_in_items_len, kv_len_arr)
        total_num_rows = int(qo_indptr_host[-1])
        self._custom_mask_buf, self._mask_indptr_buf = _plan_custom_mask(
            self, custom_mask, packed_custom_mask, qo_indptr_host, kv_len_arr, non_blocking)
        if self._custom_mask_buf is not None:
            causal = False
        if self.is_cuda_graph_enabled:
            if batch_size != self._fixed_batch_size:
                raise ValueError("The batch size should be fixed in cuda graph mode")
            self._qo_indptr_buf.copy_(qo_indptr, non_blocking=non_blocking)
            self._kv_indptr_buf.copy_(kv_indptr, non_blocking=non_blocking)
        else:
            self._qo_indptr_buf = qo_indptr.to(self.device, non_blocking=non_blocking)
            self._kv_indptr_buf = kv_indptr.to(self.device, non_blocking=non_blocking)
        plan_info = (C.c_int64 * _lib.FI_PREFILL_PLAN_INFO_LEN)()
        with torch.cuda.device(self.device):
            _lib.check(
                _lib.lib().fi_batch_prefill_plan(
                    self._float_workspace_buffer.data_ptr(),
                    self._float_workspace_buffer.numel() * self._float_workspace_buffer.element_size(),
                    self._int_workspace_buffer.data_ptr(), self._pin_memory_int_workspace_buffer.data_ptr(),
                    self._int_workspace_buffer.numel(), qo_indptr_host.data_ptr(), kv_indptr_host.data_ptr(),
                    kv_len_arr.data_ptr(), total_num_rows, batch_size, num_qo_heads, num_kv_heads, 1,
                    int(self.is_cuda_graph_enabled), head_dim_qk, head_dim_vo, int(causal), window_left,
                    -1 if fixed_split_size is None else fixed_split_size, int(disable_split_kv), plan_info,
                    _lib.current_stream(self.device),
                ),
                "BatchPrefillWithRaggedKVCacheWrapper.plan",
            )
        self._plan_info = list(plan_info)
        self._plan_info_c = plan_info
        self._batch_size = batch_size
        self._num_qo_heads = num_qo_heads
        self._num_kv_heads = num_kv_heads
        self._head_dim = head_dim_qk
        self._total_num_rows = total_num_rows
        self._total_kv_rows = int(kv_indptr_host[-1])
        self._cached_q_data_type = q_data_type
        self._cached_kv_data_type = kv_data_type
        self._causal = causal
        self._pos_encoding_mode = pos_encoding_mode
        self._window_left = window_left
        self._logits_soft_cap = logits_soft_cap
        self._sm_scale = sm_scale
        self._rope_scale = rope_scale
        self._rope_theta = rope_theta

    begin_forward = plan

    def run(
        self,
        q: torch.Tensor,
        k: torch.Tensor,
        v: torch.Tensor,
        *args,
        q_scale: Optional[float] = None,
        k_scale: Optional[float] = None,
        v_scale: Optional[float] = None,
        out: Optional[torch.Tensor] = None,
        lse: Optional[torch.Tensor] = None,
        return_lse: bool = False,
        enable_pdl: Optional[bool] = None,
    ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        r"""q ``[qo_indptr[-1], num_qo_heads, head_dim]``; k, v ragged as described in the class docstring."""
        if self._plan_info is None:
            raise RuntimeError("plan() must be called before run()")
        if args:
            raise ValueError("additional kernel arguments require jit_args, which is not supported")
        for t, name in ((q, "q"), (k, "k"), (v, "v")):
            _lib.require_gpu_tensor(t, name)
        _check_cached_qkv_data_type(q, k, self._cached_q_data_type, self._cached_kv_data_type)
        if is_float8(q):
            raise ValueError("fp8 queries are supported by the paged wrapper only")
        if k.shape != v.shape or k.stride() != v.stride() or k.dim() != 3 or k.stride(-1) != 1:
            raise ValueError("k and v must be 3-D with equal shapes/strides, contiguous in head_dim")
        if self._kv_layout == "NHD":
            nnz_kv, num_kv_heads, head_dim = k.shape
            stride_n, stride_h = k.stride(0), k.stride(1)
        else:
            num_kv_heads, nnz_kv, head_dim = k.shape
            stride_h, stride_n = k.stride(0), k.stride(1)
        if num_kv_heads != self._num_kv_heads or head_dim != self._head_dim or nnz_kv < self._total_kv_rows:
            raise ValueError("k/v shape does not match the plan")
        if q.dim() != 3 or q.shape[0] != self._total_num_rows or q.shape[1] != self._num_qo_heads:
            raise ValueError("q shape does not match the plan")
        if q.stride(-1) != 1:
            q = q.contiguous()
        logits_soft_cap = self._logits_soft_cap or 0.0
        sm_scale = self._sm_scale if self._sm_scale is not None else 1.0 / math.sqrt(q.size(-1))
        if q_scale is not None:
            sm_scale *= q_scale
        if k_scale is not None:
            sm_scale *= k_scale
        rope_scale = self._rope_scale or 1.0
        rope_theta = self._rope_theta or 1e4
        if return_lse:
            if lse is None:
                lse = torch.empty((q.size(0), q.size(1)), dtype=torch.float32, device=q.device)
            else:
                check_shape_dtype_device(lse, (q.size(0), q.size(1)), torch.float32, q.device, "lse")
        if out is None:
            out = torch.empty(q.shape[:-1] + v.shape[-1:], dtype=q.dtype, device=q.device)
        else:
            check_shape_dtype_device(out, q.shape[:-1] + v.shape[-1:], q.dtype, q.device, "out")
        alibi = _get_cache_alibi_slopes_buf(q.shape[1], q.device) if self._pos_encoding_mode == "ALIBI" else None
        params = _lib.BatchPrefillParams(
            q=q.data_ptr(), q_stride_n=q.stride(0), q_stride_h=q.stride(1), qo_indptr=self._qo_indptr_buf.data_ptr(),
            kv=_lib.PagedKV(
                k_data=k.data_ptr(), v_data=v.data_ptr(), indptr=self._kv_indptr_buf.data_ptr(), indices=None,
                last_page_len=None, rope_pos_offset=None, stride_page=stride_n, stride_n=stride_n, stride_h=stride_h,
                page_size=1, num_kv_heads=num_kv_heads, head_dim=head_dim, batch_size=self._batch_size,
                dtype=_lib.fi_dtype(k.dtype),
            ),
            o=out.data_ptr(), lse=_lib.ptr(lse) if return_lse else None, alibi_slopes=_lib.ptr(alibi),
            scale_q=None, scale_k=None, scale_v=None, num_qo_heads=self._num_qo_heads,
            custom_mask=_lib.ptr(self._custom_mask_buf), mask_indptr=_lib.ptr(self._mask_indptr_buf),
            prefix_len_ptr=_lib.ptr(self._prefix_len_ptr), token_pos_in_items_ptr=_lib.ptr(self._token_pos_in_items_ptr),
            max_item_len_ptr=_lib.ptr(self._max_item_len_ptr), token_pos_in_items_len=self._token_pos_in_items_len,
            q_dtype=_lib.fi_dtype(q.dtype), o_dtype=_lib.fi_dtype(q.dtype),
            mask_mode=_mask_mode(self),
            pos_encoding_mode=PosEncodingMode[self._pos_encoding_mode].value, window_left=self._window_left,
            logits_soft_cap=logits_soft_cap, sm_scale=sm_scale, rope_rcp_scale=1.0 / rope_scale,
            rope_rcp_theta=1.0 / rope_theta, bf16_pv_mode=getattr(self, "_bf16_pv_mode", 0),
        )
        with torch.cuda.device(q.device):
            _lib.check(
                _lib.lib().fi_batch_prefill_paged_run(
                    self._float_workspace_buffer.data_ptr(),
                    self._float_workspace_buffer.numel() * self._float_workspace_buffer.element_size(),
                    self._int_workspace_buffer.data_ptr(), self._int_workspace_buffer.numel(), self._plan_info_c,
                    _lib.FI_PREFILL_PLAN_INFO_LEN, C.byref(params), _lib.current_stream(q.device),
                ),
                "BatchPrefillWithRaggedKVCacheWrapper.run",
            )
        if v_scale is not None:
            out *= v_scale
        return (out, lse) if return_lse else out

    run_return_lse = functools.partialmethod(run, return_lse=True)

    def end_forward(self) -> None:
        pass

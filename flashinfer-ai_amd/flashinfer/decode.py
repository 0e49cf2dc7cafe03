"""Decode attention operators: ``single_decode_with_kv_cache`` and
``BatchDecodeWithPagedKVCacheWrapper`` (plan / run split).

Same names, arguments, defaults and error behaviour as the reference's ``flashinfer/decode.py``
(single :389-578; wrapper :581-1410; CUDA-graph wrapper :1413-1480), with the JIT module layer
replaced by the ahead-of-time C ABI of libfi_mi355.so.  The kernels are csrc/decode_kernel.h.
"""
from __future__ import annotations

import ctypes as C
import functools
import math
from typing import Any, List, Optional, Tuple, Union

import torch

from . import _lib
from .page import get_seq_lens
from .utils import (
    PosEncodingMode,
    TensorLayout,
    _check_cached_qkv_data_type,
    _check_kv_layout,
    _check_pos_encoding_mode,
    _get_cache_alibi_slopes_buf,
    _get_cache_buf,
    _get_range_buf,
    _unpack_paged_kv_cache,
    canonicalize_torch_dtype,
    check_shape_dtype_device,
    is_float8,
    paged_kv_strides,
)


def fast_decode_plan(
    self: "BatchDecodeWithPagedKVCacheWrapper",
    indptr: torch.Tensor,
    indices: torch.Tensor,
    last_page_len: torch.Tensor,
    num_qo_heads: int,
    num_kv_heads: int,
    head_dim: int,
    page_size: int,
    pos_encoding_mode: str = "NONE",
    window_left: int = -1,
    logits_soft_cap: Optional[float] = None,
    q_data_type: Optional[Union[str, torch.dtype]] = None,
    kv_data_type: Optional[Union[str, torch.dtype]] = None,
    data_type: Optional[Union[str, torch.dtype]] = None,
    sm_scale: Optional[float] = None,
    rope_scale: Optional[float] = None,
    rope_theta: Optional[float] = None,
    non_blocking: bool = True,
    fixed_split_size: Optional[int] = None,
    disable_split_kv: bool = False,
    global_override_indptr_cpu: Optional[torch.Tensor] = None,
) -> None:
    """A faster :meth:`BatchDecodeWithPagedKVCacheWrapper.plan` for multi-step draft decoding
    (ref: flashinfer/decode.py:2416-2579): the device-to-device copies into the graph buffers are skipped
    (the caller wrote them in place; outside graph mode the given tensors are adopted without a copy), and
    ``global_override_indptr_cpu`` supplies the host page-table prefix sums so that no device-to-host copy
    is needed.  Bind it with ``wrapper.begin_forward = functools.partial(fast_decode_plan, wrapper)``."""
    if data_type is None and q_data_type is None:
        q_data_type = "float16"
    self.plan(
        indptr, indices, last_page_len, num_qo_heads, num_kv_heads, head_dim, page_size,
        pos_encoding_mode=pos_encoding_mode, window_left=window_left, logits_soft_cap=logits_soft_cap,
        q_data_type=q_data_type, kv_data_type=kv_data_type, data_type=data_type, sm_scale=sm_scale,
        rope_scale=rope_scale, rope_theta=rope_theta, non_blocking=non_blocking,
        fixed_split_size=None,  # as the reference: not forwarded by the fast path
        disable_split_kv=disable_split_kv, _fast=True, _indptr_host=global_override_indptr_cpu,
    )


def single_decode_with_kv_cache(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    kv_layout: str = "NHD",
    pos_encoding_mode: str = "NONE",
    use_tensor_cores: bool = False,
    q_scale: Optional[float] = None,
    k_scale: Optional[float] = None,
    v_scale: Optional[float] = None,
    window_left: int = -1,
    logits_soft_cap: Optional[float] = None,
    sm_scale: Optional[float] = None,
    rope_scale: Optional[float] = None,
    rope_theta: Optional[float] = None,
    return_lse: bool = False,
) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
    r"""Decode attention with KV cache for a single request.

    Parameters
    ----------
    q : torch.Tensor
        The query tensor, shape: ``[num_qo_heads, head_dim]``.
    k, v : torch.Tensor
        ``[kv_len, num_kv_heads, head_dim]`` if :attr:`kv_layout` is ``NHD``, or
        ``[num_kv_heads, kv_len, head_dim]`` if ``HND``.
    kv_layout : str
        ``NHD`` or ``HND``.
    pos_encoding_mode : str
        ``NONE`` / ``ROPE_LLAMA`` (rotary embedding applied inside the kernel) / ``ALIBI``.
    use_tensor_cores : bool
        Use the MFMA (prefill-kernel) path; numerically equivalent.
    q_scale, k_scale, v_scale : Optional[float]
        Calibration scales for fp8 inputs (folded into ``sm_scale`` / applied to the output).
    window_left : int
        Left (inclusive) attention window; ``-1`` = whole sequence.
    logits_soft_cap : Optional[float]
        If > 0, logits are capped as ``cap * tanh(x / cap)``.
    sm_scale : Optional[float]
        Softmax scale, default ``1 / sqrt(head_dim)``.
    rope_scale, rope_theta : Optional[float]
        RoPE interpolation scale (default 1) and theta (default 1e4).
    return_lse : bool
        Also return the base-2 log-sum-exp of the attention logits, shape ``[num_qo_heads]``.
    """
    _check_pos_encoding_mode(pos_encoding_mode)
    _check_kv_layout(kv_layout)
    _lib.require_gpu_tensor(q, "q")
    _lib.require_gpu_tensor(k, "k")
    _lib.require_gpu_tensor(v, "v")
    if q.dim() != 2 or k.dim() != 3 or v.shape != k.shape:
        raise ValueError("q must be [num_qo_heads, head_dim]; k and v must be 3-D with equal shapes")
    head_dim = q.shape[-1]
    num_qo_heads = q.shape[0]
    if logits_soft_cap is None:
        logits_soft_cap = 0.0
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(head_dim)
    if q_scale is not None:
        sm_scale *= q_scale
    if k_scale is not None:
        sm_scale *= k_scale
    if rope_scale is None:
        rope_scale = 1.0
    if rope_theta is None:
        rope_theta = 1e4
    if kv_layout == "NHD":
        kv_len, num_kv_heads = k.shape[0], k.shape[1]
        stride_n, stride_h = k.stride(0), k.stride(1)
    else:
        num_kv_heads, kv_len = k.shape[0], k.shape[1]
        stride_h, stride_n = k.stride(0), k.stride(1)
    if k.stride() != v.stride() or k.stride(-1) != 1 or q.stride(-1) != 1:
        raise ValueError("k and v must share strides and q/k/v must be contiguous in head_dim")
    if num_qo_heads % num_kv_heads != 0:
        raise ValueError("num_qo_heads must be a multiple of num_kv_heads")

    tmp = _get_cache_buf("single_decode_with_kv_cache_tmp", 32 * 1024 * 1024, q.device)
    out = torch.empty_like(q)
    lse = None
    if return_lse:
        lse = torch.empty((num_qo_heads,), dtype=torch.float32, device=q.device)
    alibi = None
    if pos_encoding_mode == "ALIBI":
        alibi = _get_cache_alibi_slopes_buf(num_qo_heads, q.device)

    params = _lib.SingleDecodeParams(
        q=q.data_ptr(), q_stride_h=q.stride(0), k=k.data_ptr(), v=v.data_ptr(),
        kv_stride_n=stride_n, kv_stride_h=stride_h, o=out.data_ptr(), lse=_lib.ptr(lse),
        alibi_slopes=_lib.ptr(alibi), kv_len=kv_len, num_qo_heads=num_qo_heads,
        num_kv_heads=num_kv_heads, head_dim=head_dim, q_dtype=_lib.fi_dtype(q.dtype),
        kv_dtype=_lib.fi_dtype(k.dtype), pos_encoding_mode=PosEncodingMode[pos_encoding_mode].value,
        window_left=window_left, logits_soft_cap=logits_soft_cap, sm_scale=sm_scale,
        rope_rcp_scale=1.0 / rope_scale, rope_rcp_theta=1.0 / rope_theta,
    )
    with torch.cuda.device(q.device):
        _lib.check(
            _lib.lib().fi_single_decode_run(
                C.byref(params), tmp.data_ptr(), tmp.numel(), _lib.current_stream(q.device)
            ),
            "single_decode_with_kv_cache",
        )
    if v_scale is not None:
        if is_float8(out):
            out = (out.to(torch.float32) * v_scale).to(out.dtype)
        else:
            out *= v_scale
    return (out, lse) if return_lse else out


class BatchDecodeWithPagedKVCacheWrapper:
    r"""Decode attention over a paged KV cache for a batch of requests.

    ``plan()`` is host work done once per batch shape and reused by every layer; ``run()`` launches the
    kernels on the current stream.  See the reference docstring (flashinfer/decode.py:581-646) for the
    page-table layout; the example there runs unchanged:

    >>> workspace_buffer = torch.zeros(128 * 1024 * 1024, dtype=torch.uint8, device="cuda:0")
    >>> decode_wrapper = flashinfer.BatchDecodeWithPagedKVCacheWrapper(workspace_buffer, "NHD")
    >>> decode_wrapper.plan(kv_page_indptr, kv_page_indices, kv_last_page_len, num_qo_heads,
    ...                     num_kv_heads, head_dim, page_size, pos_encoding_mode="NONE",
    ...                     data_type=torch.float16)
    >>> o = decode_wrapper.run(q, kv_cache)
    """

    def __init__(
        self,
        float_workspace_buffer: torch.Tensor,
        kv_layout: str = "NHD",
        use_cuda_graph: bool = False,
        use_tensor_cores: bool = False,
        paged_kv_indptr_buffer: Optional[torch.Tensor] = None,
        paged_kv_indices_buffer: Optional[torch.Tensor] = None,
        paged_kv_last_page_len_buffer: Optional[torch.Tensor] = None,
        backend: str = "auto",
        jit_args: Optional[List[Any]] = None,
    ) -> None:
        r"""Parameters as the reference (flashinfer/decode.py:647-776).

        float_workspace_buffer : split-KV partial states live here (128 MB recommended).
        use_cuda_graph : keep the launch shape fixed so ``run`` can be captured in a hipGraph;
            the three ``paged_kv_*_buffer`` tensors are then required and the batch size is fixed.
        use_tensor_cores : accepted; selects the MFMA path when available.
        backend : ``auto`` / ``fa2`` (one native backend exists; NVIDIA-only names are rejected).
        jit_args : must be None -- there is no JIT in this build.
        """
        _check_kv_layout(kv_layout)
        if jit_args is not None:
            raise ValueError("jit_args is not supported: kernels are built ahead of time")
        if backend not in ("auto", "fa2"):
            raise ValueError(f"backend {backend!r} is not available on MI355X (use 'auto')")
        _lib.require_gpu_tensor(float_workspace_buffer, "float_workspace_buffer")
        self._kv_layout = kv_layout
        self._float_workspace_buffer = float_workspace_buffer
        self.device = float_workspace_buffer.device
        self._int_workspace_buffer = torch.empty(
            (8 * 1024 * 1024,), dtype=torch.uint8, device=self.device
        )
        self._pin_memory_int_workspace_buffer = torch.empty(
            (8 * 1024 * 1024,), dtype=torch.uint8, pin_memory=True, device="cpu"
        )
        if use_cuda_graph:
            if not torch.is_tensor(paged_kv_indptr_buffer):
                raise ValueError("paged_kv_indptr_buffer should be a torch.Tensor in cudagraph mode")
            if not torch.is_tensor(paged_kv_indices_buffer):
                raise ValueError("paged_kv_indices_buffer should be a torch.Tensor in cudagraph mode")
            if not torch.is_tensor(paged_kv_last_page_len_buffer):
                raise ValueError(
                    "paged_kv_last_page_len_buffer should be a torch.Tensor in cudagraph mode"
                )
            self._fixed_batch_size = len(paged_kv_last_page_len_buffer)
            if len(paged_kv_indptr_buffer) != self._fixed_batch_size + 1:
                raise ValueError("The size of paged_kv_indptr_buffer should be batch_size + 1")
        else:
            self._fixed_batch_size = 0
        self._paged_kv_indptr_buf = paged_kv_indptr_buffer
        self._paged_kv_indices_buf = paged_kv_indices_buffer
        self._paged_kv_last_page_len_buf = paged_kv_last_page_len_buffer
        self._use_tensor_cores = use_tensor_cores
        self._use_cuda_graph = use_cuda_graph
        self._backend = backend
        self._plan_info: Optional[List[int]] = None
        self._run_cache = None
        self._plan_serial = 0

    @property
    def use_tensor_cores(self) -> bool:
        return self._use_tensor_cores

    @property
    def is_cuda_graph_enabled(self) -> bool:
        return self._use_cuda_graph

    def reset_workspace_buffer(
        self, float_workspace_buffer: torch.Tensor, int_workspace_buffer: torch.Tensor
    ) -> None:
        r"""Swap the workspaces; a new pinned mirror of the int workspace is allocated."""
        self._run_cache = None
        self._float_workspace_buffer = float_workspace_buffer
        self._int_workspace_buffer = int_workspace_buffer
        self._pin_memory_int_workspace_buffer = torch.empty(
            self._int_workspace_buffer.shape,
            dtype=self._int_workspace_buffer.dtype,
            device="cpu",
            pin_memory=True,
        )

    def plan(
        self,
        indptr: torch.Tensor,
        indices: torch.Tensor,
        last_page_len: torch.Tensor,
        num_qo_heads: int,
        num_kv_heads: int,
        head_dim: int,
        page_size: int,
        pos_encoding_mode: str = "NONE",
        window_left: int = -1,
        logits_soft_cap: Optional[float] = None,
        q_data_type: Optional[Union[str, torch.dtype]] = "float16",
        kv_data_type: Optional[Union[str, torch.dtype]] = None,
        data_type: Optional[Union[str, torch.dtype]] = None,
        sm_scale: Optional[float] = None,
        rope_scale: Optional[float] = None,
        rope_theta: Optional[float] = None,
        non_blocking: bool = True,
        block_tables: Optional[torch.Tensor] = None,
        seq_lens: Optional[torch.Tensor] = None,
        fixed_split_size: Optional[int] = None,
        disable_split_kv: bool = False,
        _fast: bool = False,
        _indptr_host: Optional[torch.Tensor] = None,
    ) -> None:
        r"""Plan batch decode for the given page table (ref: flashinfer/decode.py:810-1104).

        indptr : ``[batch_size + 1]`` int32, indices : ``[indptr[-1]]`` int32,
        last_page_len : ``[batch_size]`` int32 (1 <= last_page_len <= page_size).
        The remaining arguments configure the attention variant exactly as in the reference.
        ``plan`` is synchronous host code and must not be captured in a graph.
        """
        for tensor, name in [(indptr, "indptr"), (indices, "indices"), (last_page_len, "last_page_len")]:
            if tensor.dtype != torch.int32:
                raise ValueError(f"{name} must have dtype torch.int32, got {tensor.dtype}")
        _check_pos_encoding_mode(pos_encoding_mode)
        batch_size = len(last_page_len)
        if logits_soft_cap is None:
            logits_soft_cap = 0.0
        if self.is_cuda_graph_enabled:
            if batch_size != self._fixed_batch_size:
                raise ValueError(
                    "The batch size should be fixed in cudagraph mode, the runtime batch size {} "
                    " mismatches the batch size set during initialization {}".format(
                        batch_size, self._fixed_batch_size
                    )
                )
            if len(indices) > len(self._paged_kv_indices_buf):
                raise ValueError(
                    "The size of indices should be less than or equal to the allocated buffer"
                )
            if not _fast:  # fast_decode_plan: the caller already wrote the graph buffers in place
                self._paged_kv_indptr_buf.copy_(indptr, non_blocking=non_blocking)
                self._paged_kv_last_page_len_buf.copy_(last_page_len, non_blocking=non_blocking)
                self._paged_kv_indices_buf[: len(indices)].copy_(
                    indices, non_blocking=(indices.device == self.device) and non_blocking
                )
        elif _fast:
            self._paged_kv_indptr_buf = indptr
            self._paged_kv_indices_buf = indices
            self._paged_kv_last_page_len_buf = last_page_len
        else:
            self._paged_kv_indptr_buf = indptr.to(self.device, non_blocking=non_blocking)
            self._paged_kv_indices_buf = indices.to(self.device, non_blocking=non_blocking)
            self._paged_kv_last_page_len_buf = last_page_len.to(self.device, non_blocking=non_blocking)
        indptr_host = (indptr if _indptr_host is None else _indptr_host).to("cpu").contiguous()
        if indptr_host.dtype != torch.int32 or len(indptr_host) != batch_size + 1:
            raise ValueError("indptr must be int32 with batch_size + 1 entries")

        if data_type is not None:
            if q_data_type is None:
                q_data_type = data_type
            if kv_data_type is None:
                kv_data_type = data_type
        q_data_type = canonicalize_torch_dtype(q_data_type)
        if kv_data_type is None:
            kv_data_type = q_data_type
        kv_data_type = canonicalize_torch_dtype(kv_data_type)
        if fixed_split_size is not None and not self.use_tensor_cores:
            raise ValueError("fixed_split_size is only supported by tensor core decode for now.")

        self._cached_q_data_type = q_data_type
        self._cached_kv_data_type = kv_data_type
        self._batch_size = batch_size
        self._num_qo_heads = num_qo_heads
        self._num_kv_heads = num_kv_heads
        self._head_dim = head_dim
        self._page_size = page_size
        if _fast:
            self._kv_lens_host = None  # not needed by run(); skipping it saves a device-to-host copy
        elif seq_lens is None:
            last_page_len_host = last_page_len.to("cpu")
            self._kv_lens_host = get_seq_lens(indptr_host, last_page_len_host, page_size)
        else:
            self._kv_lens_host = seq_lens.cpu()

        plan_info = (C.c_int64 * _lib.FI_DECODE_PLAN_INFO_LEN)()
        max_grid_hint = 0
        if disable_split_kv:
            max_grid_hint = 1  # batch*heads >= 1 always: the planner never splits
        with torch.cuda.device(self.device):
            _lib.check(
                _lib.lib().fi_batch_decode_plan(
                    self._float_workspace_buffer.data_ptr(),
                    self._float_workspace_buffer.numel() * self._float_workspace_buffer.element_size(),
                    self._int_workspace_buffer.data_ptr(),
                    self._pin_memory_int_workspace_buffer.data_ptr(),
                    self._int_workspace_buffer.numel(),
                    indptr_host.data_ptr(),
                    batch_size,
                    num_qo_heads,
                    num_kv_heads,
                    page_size,
                    int(self.is_cuda_graph_enabled),
                    head_dim,
                    _lib.fi_dtype(q_data_type),
                    _lib.fi_dtype(kv_data_type),
                    max_grid_hint,
                    window_left,
                    plan_info,
                    _lib.current_stream(self.device),
                ),
                "BatchDecodeWithPagedKVCacheWrapper.plan",
            )
        self._plan_info = list(plan_info)
        self._plan_info_c = plan_info
        self._plan_serial = getattr(self, "_plan_serial", 0) + 1
        self._run_cache = None
        self._pos_encoding_mode = pos_encoding_mode
        self._window_left = window_left
        self._logits_soft_cap = logits_soft_cap
        self._sm_scale = sm_scale
        self._rope_scale = rope_scale
        self._rope_theta = rope_theta

    begin_forward = plan

    def forward(
        self,
        q: torch.Tensor,
        paged_kv_cache: Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]],
        pos_encoding_mode: str = "NONE",
        q_scale: Optional[float] = None,
        k_scale: Optional[float] = None,
        v_scale: Optional[float] = None,
        window_left: int = -1,
        logits_soft_cap: Optional[float] = None,
        sm_scale: Optional[float] = None,
        rope_scale: Optional[float] = None,
        rope_theta: Optional[float] = None,
    ) -> torch.Tensor:
        r"""Warning: this function is deprecated, please use :meth:`run` instead."""
        self._pos_encoding_mode = pos_encoding_mode
        self._window_left = window_left
        self._logits_soft_cap = logits_soft_cap
        self._sm_scale = sm_scale
        self._rope_scale = rope_scale
        self._rope_theta = rope_theta
        self._run_cache = None
        return self.run(q, paged_kv_cache, q_scale=q_scale, k_scale=k_scale, v_scale=v_scale)

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
        q_len_per_req: Optional[int] = 1,
    ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        r"""Compute batch decode attention between ``q`` and the paged KV cache.

        q : ``[batch_size, num_qo_heads, head_dim]``.
        paged_kv_cache : a 5-D tensor ``[max_num_pages, 2, page_size, num_kv_heads, head_dim]`` (NHD) /
            ``[max_num_pages, 2, num_kv_heads, page_size, head_dim]`` (HND), or a ``(k_cache, v_cache)``
            tuple of 4-D tensors.
        Returns the output ``[batch_size, num_qo_heads, head_dim]`` (and the base-2 logsumexp
        ``[batch_size, num_qo_heads]`` when ``return_lse``).  (ref: flashinfer/decode.py:1163-1374)
        """
        if self._plan_info is None:
            raise RuntimeError("plan() must be called before run()")
        if sinks is not None:
            raise ValueError("attention sinks are not supported by this backend")
        if args:
            raise ValueError("additional kernel arguments require jit_args, which is not supported")
        window_left = self._window_left if window_left is None else window_left
        # window_left is part of the plan in the reference; keep the same contract
        assert window_left == self._window_left
        if return_lse and lse is None:
            lse = torch.empty((q.size(0), q.size(1)), dtype=torch.float32, device=q.device)
        if out is None:
            out = torch.empty(q.shape, dtype=q.dtype, device=q.device)

        # Steady-state calls (same tensors every layer / step) reuse the validated argument block.
        cache_t = paged_kv_cache if torch.is_tensor(paged_kv_cache) else paged_kv_cache[0]
        key = (q.data_ptr(), q.stride(), q.shape, cache_t.data_ptr(), cache_t.stride(), cache_t.shape,
               None if torch.is_tensor(paged_kv_cache) else paged_kv_cache[1].data_ptr(),
               out.data_ptr(), lse.data_ptr() if return_lse else 0, q_scale, k_scale, self._plan_serial)
        cached = self._run_cache
        if cached is not None and cached[0] == key:
            params = cached[1]
        else:
            params = self._build_run_params(q, paged_kv_cache, q_scale, k_scale, out, lse if return_lse else None)
            self._run_cache = (key, params, q, paged_kv_cache)  # keep the tensors alive with the pointers
        dev_index = q.device.index
        if torch.cuda.current_device() == dev_index:
            status = self._lib_run(self._fws_ptr, self._fws_bytes, self._iws_ptr, self._iws_bytes,
                                   self._plan_info_c, _lib.FI_DECODE_PLAN_INFO_LEN, params,
                                   torch.cuda.current_stream().cuda_stream)
        else:
            with torch.cuda.device(q.device):
                status = self._lib_run(self._fws_ptr, self._fws_bytes, self._iws_ptr, self._iws_bytes,
                                       self._plan_info_c, _lib.FI_DECODE_PLAN_INFO_LEN, params,
                                       torch.cuda.current_stream().cuda_stream)
        if status != 0:
            _lib.check(status, "BatchDecodeWithPagedKVCacheWrapper.run")
        if v_scale is not None:
            if is_float8(out):
                out = (out.to(torch.float32) * v_scale).to(out.dtype)
            else:
                out *= v_scale
        return (out, lse) if return_lse else out

    def _build_run_params(self, q, paged_kv_cache, q_scale, k_scale, out, lse):
        """Validate the run() arguments against the plan and build the C argument block."""
        _lib.require_gpu_tensor(q, "q")
        k_cache, v_cache = _unpack_paged_kv_cache(paged_kv_cache, self._kv_layout)
        _check_cached_qkv_data_type(q, k_cache, self._cached_q_data_type, self._cached_kv_data_type)
        page_size, num_kv_heads, head_dim, stride_page, stride_n, stride_h = paged_kv_strides(
            k_cache, v_cache, self._kv_layout
        )
        pos_encoding_mode = self._pos_encoding_mode
        _check_pos_encoding_mode(pos_encoding_mode)
        logits_soft_cap = self._logits_soft_cap
        sm_scale = self._sm_scale
        rope_scale = self._rope_scale
        rope_theta = self._rope_theta
        if logits_soft_cap is None:
            logits_soft_cap = 0.0
        if sm_scale is None:
            sm_scale = 1.0 / math.sqrt(q.shape[-1])
        if q_scale is not None:
            sm_scale *= q_scale
        if k_scale is not None:
            sm_scale *= k_scale
        if rope_scale is None:
            rope_scale = 1.0
        if rope_theta is None:
            rope_theta = 1e4
        if q.dim() != 3 or q.shape[0] != self._batch_size or q.shape[1] != self._num_qo_heads:
            raise ValueError(
                f"q must have shape [{self._batch_size}, {self._num_qo_heads}, head_dim], got {tuple(q.shape)}"
            )
        if q.shape[2] != head_dim or head_dim != self._head_dim:
            raise ValueError("head_dim of q / kv cache does not match the planned head_dim")
        if num_kv_heads != self._num_kv_heads or page_size != self._page_size:
            raise ValueError("kv cache shape does not match the planned num_kv_heads / page_size")
        if q.stride(-1) != 1:
            raise ValueError("q must be contiguous in head_dim")
        if lse is not None:
            check_shape_dtype_device(lse, (q.size(0), q.size(1)), torch.float32, q.device, "lse")
        check_shape_dtype_device(out, q.shape, q.dtype, q.device, "out")
        if not out.is_contiguous():
            raise ValueError("out must be contiguous")
        alibi = None
        if pos_encoding_mode == "ALIBI":
            alibi = _get_cache_alibi_slopes_buf(q.shape[1], q.device)
        params = _lib.BatchDecodeParams(
            q=q.data_ptr(), q_stride_n=q.stride(0), q_stride_h=q.stride(1),
            kv=_lib.PagedKV(
                k_data=k_cache.data_ptr(), v_data=v_cache.data_ptr(),
                indptr=self._paged_kv_indptr_buf.data_ptr(),
                indices=self._paged_kv_indices_buf.data_ptr(),
                last_page_len=self._paged_kv_last_page_len_buf.data_ptr(),
                rope_pos_offset=None, stride_page=stride_page, stride_n=stride_n, stride_h=stride_h,
                page_size=page_size, num_kv_heads=num_kv_heads, head_dim=head_dim,
                batch_size=self._batch_size, dtype=_lib.fi_dtype(k_cache.dtype),
            ),
            o=out.data_ptr(), lse=_lib.ptr(lse),
            alibi_slopes=_lib.ptr(alibi), q_rope_offset=None, num_qo_heads=self._num_qo_heads,
            q_dtype=_lib.fi_dtype(q.dtype), pos_encoding_mode=PosEncodingMode[pos_encoding_mode].value,
            window_left=self._window_left, logits_soft_cap=logits_soft_cap, sm_scale=sm_scale,
            rope_rcp_scale=1.0 / rope_scale, rope_rcp_theta=1.0 / rope_theta,
        )
        self._fws_ptr = self._float_workspace_buffer.data_ptr()
        self._fws_bytes = self._float_workspace_buffer.numel() * self._float_workspace_buffer.element_size()
        self._iws_ptr = self._int_workspace_buffer.data_ptr()
        self._iws_bytes = self._int_workspace_buffer.numel()
        self._lib_run = _lib.lib().fi_batch_decode_run
        return C.byref(params)

    def forward_return_lse(
        self,
        q: torch.Tensor,
        paged_kv_cache: torch.Tensor,
        pos_encoding_mode: str = "NONE",
        q_scale: Optional[float] = None,
        k_scale: Optional[float] = None,
        v_scale: Optional[float] = None,
        window_left: int = -1,
        logits_soft_cap: Optional[float] = None,
        sm_scale: Optional[float] = None,
        rope_scale: Optional[float] = None,
        rope_theta: Optional[float] = None,
    ) -> Tuple[torch.Tensor, torch.Tensor]:
        r"""Warning: this function is deprecated, please use :meth:`run_return_lse` instead."""
        self._pos_encoding_mode = pos_encoding_mode
        self._window_left = window_left
        self._logits_soft_cap = logits_soft_cap
        self._sm_scale = sm_scale
        self._rope_scale = rope_scale
        self._rope_theta = rope_theta
        self._run_cache = None
        return self.run(
            q, paged_kv_cache, q_scale=q_scale, k_scale=k_scale, v_scale=v_scale, return_lse=True
        )

    run_return_lse = functools.partialmethod(run, return_lse=True)

    def end_forward(self) -> None:
        r"""Warning: this function is deprecated and has no effect."""
        pass


class CUDAGraphBatchDecodeWithPagedKVCacheWrapper(BatchDecodeWithPagedKVCacheWrapper):
    r"""Graph-capturable batch decode wrapper (hipGraph on ROCm): fixed batch size and launch shape.
    (ref: flashinfer/decode.py:1413-1480)"""

    def __init__(
        self,
        workspace_buffer: torch.Tensor,
        indptr_buffer: torch.Tensor,
        indices_buffer: torch.Tensor,
        last_page_len_buffer: torch.Tensor,
        kv_layout: str = "NHD",
        use_tensor_cores: bool = False,
    ) -> None:
        super().__init__(
            workspace_buffer,
            kv_layout,
            use_cuda_graph=True,
            use_tensor_cores=use_tensor_cores,
            paged_kv_indptr_buffer=indptr_buffer,
            paged_kv_indices_buffer=indices_buffer,
            paged_kv_last_page_len_buffer=last_page_len_buffer,
        )

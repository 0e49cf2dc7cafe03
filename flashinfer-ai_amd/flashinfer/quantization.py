"""Bit packing of boolean masks: the producer of the ``packed_custom_mask`` operand of prefill.
Mirrors flashinfer/quantization.py:57-153 (kernels: include/flashinfer/quantization.cuh:29-126)."""
from typing import Tuple

import torch

from . import _lib


def _as_bytes(x: torch.Tensor) -> torch.Tensor:
    _lib.require_gpu_tensor(x, "x")
    if x.dim() != 1:
        raise ValueError("x must be a 1D binary-valued tensor")
    if x.dtype == torch.bool:
        return x.contiguous().view(torch.uint8)
    return (x != 0).view(torch.uint8)


def _bitorder(bitorder: str) -> int:
    if bitorder not in ("big", "little"):
        raise ValueError("bitorder must be either 'big' or 'little'")
    return int(bitorder == "little")


def packbits(x: torch.Tensor, bitorder: str = "big") -> torch.Tensor:
    r"""Pack the elements of a binary-valued array into bits in a uint8 array (``numpy.packbits``
    semantics).  Returns ``((x.size(0) + 7) // 8,)`` uint8.

    >>> x = torch.tensor([1, 0, 1, 1, 0, 0, 1, 1], dtype=torch.bool, device="cuda")
    >>> list(map(bin, packbits(x).tolist()))
    ['0b10110011']
    """
    little = _bitorder(bitorder)
    xb = _as_bytes(x)
    y = torch.empty((xb.numel() + 7) // 8, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(
            _lib.lib().fi_packbits(xb.data_ptr(), xb.numel(), little, y.data_ptr(), _lib.current_stream(x.device)),
            "packbits",
        )
    return y


def segment_packbits(x: torch.Tensor, indptr: torch.Tensor, bitorder: str = "big") -> Tuple[torch.Tensor, torch.Tensor]:
    r"""Pack a batch of binary-valued segments ``x[indptr[i]:indptr[i+1]]`` into bits; every segment
    starts a new byte.  Returns ``(y, new_indptr)`` with
    ``new_indptr[i+1] - new_indptr[i] == (indptr[i+1] - indptr[i] + 7) // 8``.

    >>> x = torch.tensor([1, 0, 1, 1, 0, 0, 1, 1, 1, 0, 1], dtype=torch.bool, device="cuda")
    >>> y, new_indptr = segment_packbits(x, torch.tensor([0, 4, 7, 11], device="cuda"), bitorder="big")
    >>> list(map(bin, y.tolist())), new_indptr.tolist()
    (['0b10110000', '0b100000', '0b11010000'], [0, 1, 2, 3])
    """
    little = _bitorder(bitorder)
    xb = _as_bytes(x)
    seglen = indptr[1:] - indptr[:-1]
    packed_len = (seglen + 7) // 8
    indptr_new = torch.zeros(len(indptr), dtype=indptr.dtype, device=indptr.device)
    indptr_new[1:] = torch.cumsum(packed_len, 0)
    output_nnzs = int(indptr_new[-1].item())
    indptr = indptr.to(device=x.device, dtype=torch.int32)
    indptr_new = indptr_new.to(device=x.device, dtype=torch.int32)
    y = torch.empty(output_nnzs, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(
            _lib.lib().fi_segment_packbits(
                xb.data_ptr(), indptr.data_ptr(), indptr_new.data_ptr(), len(indptr) - 1, output_nnzs, little,
                y.data_ptr(), _lib.current_stream(x.device),
            ),
            "segment_packbits",
        )
    return y, indptr_new

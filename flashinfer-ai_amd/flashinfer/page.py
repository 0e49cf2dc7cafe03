"""Page-table utilities (ref: flashinfer/page.py)."""
from __future__ import annotations

import torch


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

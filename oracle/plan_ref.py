"""Pure-Python restatement of the decode planner (integer work partitioning).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Follows the reference's host scheduler:
  PartitionPagedKVCacheBinarySearchMinNumPagePerBatch  include/flashinfer/attention/scheduler.cuh:73-99
  BatchDecodeWithPagedKVCacheWorkEstimationDispatched  scheduler.cuh:183-207 (split decision)
  DecodeSplitKVIndptr                                   scheduler.cuh:348-364
  DecodePlan (padded batch, block_valid_mask)           scheduler.cuh:424-493
The only build-specific inputs are `max_grid` (the reference derives it from CUDA occupancy x #SM; the
MI355X planner uses CUs x resident waves) and `gdy` (work items per (request, chunk): the reference's
gridDim.y = num_kv_heads; here num_kv_heads x q-head tiles).
"""
from __future__ import annotations

from typing import List, Tuple


def ceil_div(a: int, b: int) -> int:
    return (a + b - 1) // b


def pick_head_tile(group_size: int, kv_bytes: int = 2) -> int:
    if group_size <= 1:
        return 1
    if group_size == 2:
        return 2
    return 4


def head_tiles(group_size: int, kv_bytes: int = 2, head_dim: int = 128, page_size: int = 16) -> int:
    """Work items per (request chunk, kv head).  Groups of >= 5 heads (>= 3 with an fp8 cache; one work item per 32 heads) at
    head_dim 64/128 run on the matrix-core decode kernel, where one wave covers the whole group; smaller
    groups use the VALU kernel's q-head tiles of <= 4 heads (flashinfer-ai_amd/csrc/decode.hip)."""
    min_group = 3 if kv_bytes == 1 else 5
    # pages too small / not a power of two for the VALU kernel's scalar-page path: matrix-core kernel for
    # every group size
    tokens_per_load = 64 // (head_dim // (16 // kv_bytes))
    if page_size & (page_size - 1) or page_size < tokens_per_load:
        min_group = 1
    if min_group <= group_size and head_dim in (64, 128):
        return ceil_div(group_size, 32)  # one wave covers 32 heads of the group
    return ceil_div(group_size, pick_head_tile(group_size, kv_bytes))


def partition_pages(max_grid: int, gdy: int, num_pages: List[int], min_pages: int) -> Tuple[int, int]:
    low, high = min_pages, max(num_pages + [0])
    while low < high:
        mid = (low + high) // 2
        nb = sum(ceil_div(e, mid) for e in num_pages)
        if nb * gdy > max_grid:
            low = mid + 1
        else:
            high = mid
    return low, sum(ceil_div(max(e, 1), low) for e in num_pages)


def decode_plan_ref(indptr: List[int], num_qo_heads: int, num_kv_heads: int, page_size: int,
                    max_grid: int, enable_cuda_graph: bool = False, kv_bytes: int = 2, head_dim: int = 128,
                    window_left: int = -1):
    batch = len(indptr) - 1
    group = num_qo_heads // num_kv_heads
    gdy = num_kv_heads * head_tiles(group, kv_bytes, head_dim, page_size)
    num_pages = [indptr[i + 1] - indptr[i] for i in range(batch)]
    if window_left >= 0:
        # only the pages from the one holding the earliest key the last token can see are partitioned
        # (kv_len >= (pages - 1) * page_size + 1); build-specific, the reference reads and masks them
        num_pages = [n - max((n - 1) * page_size - window_left, 0) // page_size if n > 0 else n for n in num_pages]
    if batch * gdy >= max_grid:
        split, chunk_pages, new_batch = False, max(num_pages + [1]), batch
    else:
        chunk_pages, new_batch = partition_pages(max_grid, gdy, num_pages, max(128 // page_size, 1))
        split = not (new_batch == batch and not enable_cuda_graph)
    if enable_cuda_graph:
        padded = max(max_grid // gdy, new_batch) if split else batch
    else:
        padded = new_batch
    request_indices, kv_tile_indices, o_indptr = [], [], [0]
    for b in range(batch):
        n = ceil_div(max(num_pages[b], 1), chunk_pages) if split else 1
        for t in range(n):
            request_indices.append(b)
            kv_tile_indices.append(t)
        o_indptr.append(o_indptr[-1] + n)
    return dict(split_kv=split, kv_chunk_size=chunk_pages * page_size, padded_batch_size=padded,
                num_work=len(request_indices), request_indices=request_indices,
                kv_tile_indices=kv_tile_indices, o_indptr=o_indptr)


def prefill_plan_ref(qo_indptr: List[int], kv_lens: List[int], num_qo_heads: int, num_kv_heads: int,
                     causal: bool = False, enable_cuda_graph: bool = False, total_num_rows: int = None,
                     fixed_split_size: int = -1, disable_split_kv: bool = False, num_cus: int = 256,
                     tile_q: int = 128, tile_kv: int = 64, float_ws_bytes: int = None, head_dim: int = 128,
                     window_left: int = -1):
    """Work list of the prefill planner (flashinfer-ai_amd/csrc/prefill.hip), restating the reference's
      PrefillBinarySearchKVChunkSize   scheduler.cuh:101-130
      PrefillSplitQOKVIndptr           scheduler.cuh:495-614   (packed_qo_len = qo_len * G, merge_indptr)
      max_batch_size_if_split = max_grid / num_kv_heads        scheduler.cuh:716-718
    with the build's constants: a fixed 128-row q tile, chunk sizes in units of one 64-row kv tile (>= 128
    tokens), max_grid = 2 workgroups x CUs, requests ordered by (rows per q tile x kv length) descending and -- under a causal
    mask -- the later (heavier) q tiles first; plus the build's load-balance rule for mixed batches (below)."""
    batch = len(kv_lens)
    group = num_qo_heads // num_kv_heads
    q_tiles = [ceil_div((qo_indptr[b + 1] - qo_indptr[b]) * group, tile_q) for b in range(batch)]
    kv = [max(k, 1) for k in kv_lens]
    if window_left >= 0:  # span a q tile walks under a sliding window (ref effective_kv_len_arr, scheduler.cuh:561-567)
        kv = [min(kv[b], window_left + (tile_q if causal else qo_indptr[b + 1] - qo_indptr[b]) + tile_kv)
              for b in range(batch)]
    max_kv = max(kv + [1])
    max_items = max(num_cus * 2 // num_kv_heads, 1)
    if total_num_rows is None:
        total_num_rows = qo_indptr[-1]
    chunk = ceil_div(max_kv, tile_kv) * tile_kv
    split = False
    if not disable_split_kv and batch > 0:
        if fixed_split_size > 0:
            chunk = ceil_div(fixed_split_size, tile_kv) * tile_kv
        else:
            low, high = 128 // tile_kv, ceil_div(max_kv, tile_kv)
            while low < high:
                mid = (low + high) // 2
                if sum(q_tiles[b] * ceil_div(kv[b], mid * tile_kv) for b in range(batch)) > max_items:
                    low = mid + 1
                else:
                    high = mid
            chunk = max(low, 128 // tile_kv) * tile_kv
            if not enable_cuda_graph:
                # the build's price model (prefill.hip): chunk, 2 x chunk, ... and "whole"; rounds per CU x tokens per
                # item x 20 ns (per 256 of head_dim_qk + head_dim_vo) + for a split 6 us + partial bytes at 3 TB/s
                tok_ns = max(20 * (2 * head_dim) // 256, 1)

                def cost_ns(c, split_):
                    items = sum(q_tiles[b] * (ceil_div(kv[b], c) if split_ else 1) for b in range(batch))
                    entries = sum((qo_indptr[b + 1] - qo_indptr[b]) * (ceil_div(kv[b], c) if split_ else 1)
                                  for b in range(batch))
                    t = ceil_div(items * num_kv_heads, num_cus) * min(c, max_kv) * tok_ns
                    if split_:
                        t += 6000 + entries * num_qo_heads * head_dim * 8 // 3000
                    return t
                whole = ceil_div(max_kv, tile_kv) * tile_kv
                best, best_cost = whole, cost_ns(whole, False)
                c = chunk
                while c < max_kv:
                    t = cost_ns(c, True)
                    if t < best_cost or (t == best_cost and c > best and best != whole):
                        best, best_cost = c, t
                    c *= 2
                chunk = best
        if fixed_split_size <= 0 and not enable_cuda_graph and chunk >= max_kv and sum(q_tiles) > 0:
            # load balance of mixed batches (the build's own rule, see prefill.hip): where the reference rule left
            # every request whole and the longest one is >= twice the mean item, chunks of at most half the ideal
            # makespan W / max_items, at least 256 tokens, at most 8 x max_items items
            work = sum(q_tiles[b] * kv[b] for b in range(batch))
            bal = ceil_div(max(work // (2 * max_items), 256), tile_kv) * tile_kv
            if max_kv * sum(q_tiles) >= 2 * work and 2 * bal <= max_kv and \
                    sum(q_tiles[b] * ceil_div(kv[b], bal) for b in range(batch)) <= 8 * max_items:
                chunk = bal
        if fixed_split_size <= 0 and float_ws_bytes is not None:
            # partial states (f32 o + lse per entry and head) must fit the caller's float workspace
            def need(c):
                entries = sum((qo_indptr[b + 1] - qo_indptr[b]) * ceil_div(kv[b], c) for b in range(batch))
                return (entries * num_qo_heads * (head_dim + 1) + 64) * 4
            while chunk < max_kv and need(chunk) > float_ws_bytes:
                chunk *= 2
        split = chunk < max_kv or enable_cuda_graph
    def item_cost(b):  # rows of a q tile x kv length
        return kv_lens[b] * max(min((qo_indptr[b + 1] - qo_indptr[b]) * group, tile_q), 1)
    order = sorted(range(batch), key=lambda b: -item_cost(b))  # stable
    req, qt, kt = [], [], []
    for b in order:
        nchunks = ceil_div(kv[b], chunk) if split else 1
        for t in range(q_tiles[b]):
            for c in range(nchunks):
                req.append(b)
                qt.append(q_tiles[b] - 1 - t if causal else t)
                kt.append(c)
    # mixed batches: wide (>= half a q tile of rows, compute-bound) and narrow (decode-like, memory-bound) items are
    # interleaved in proportion, each kind in its costliest-first order (see prefill.hip)
    wide, narrow = [], []
    for i in range(len(req)):
        rows_left = (qo_indptr[req[i] + 1] - qo_indptr[req[i]]) * group - qt[i] * tile_q
        (wide if min(rows_left, tile_q) * 2 >= tile_q else narrow).append(i)
    if wide and narrow:
        n, iw, i_n, idx = len(req), 0, 0, []
        for k in range(n):
            if i_n >= len(narrow) or (iw < len(wide) and iw * n <= k * len(wide)):
                idx.append(wide[iw]); iw += 1
            else:
                idx.append(narrow[i_n]); i_n += 1
        req, qt, kt = [req[i] for i in idx], [qt[i] for i in idx], [kt[i] for i in idx]
    merge_indptr = [0]
    if split:
        for b in range(batch):
            for _ in range(qo_indptr[b + 1] - qo_indptr[b]):
                merge_indptr.append(merge_indptr[-1] + ceil_div(kv[b], chunk))
    padded = len(req)
    if enable_cuda_graph:
        padded = max(padded, max_items, ceil_div(total_num_rows * group, tile_q) + max(batch, 1) - 1)
    return dict(split_kv=split, kv_chunk_size=chunk, request_indices=req, qo_tile_indices=qt,
                kv_tile_indices=kt, merge_indptr=merge_indptr, padded_batch_size=padded, num_work=len(req))

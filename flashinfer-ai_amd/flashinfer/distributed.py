"""Batch sharding across the GPUs of a node, and the one exchange step of the path: the partial
attention states of a sequence-sharded SHARED PREFIX (cascade inference, BASELINE config C5).

The reference has no multi-GPU code on this path; its recursive-attention note
(docs/tutorials/recursive_attention.rst:38-52, 67-73) states that `merge_state` makes KV-sequence
parallelism possible.  Design for MI355X (one process per GPU, torch.distributed "nccl" = RCCL over xGMI):

  * decode / prefill / GEMM shard by request (or group): every rank owns a batch slice, its own page
    table and workspaces -- no communication at all (`shard_range`).
  * shared-prefix decode: the prefix KV is split by pages over the ranks.  Each rank attends ALL queries
    over its prefix shard (one non-causal prefill "request" of qo_len = total batch) and its OWN queries
    over their unique suffixes (batch decode).  The prefix partial states (v: [B, H, D] 16-bit, s: [B, H]
    f32) are exchanged with ONE all_to_all_single -- on the fully connected xGMI mesh all 7 peer links
    carry traffic at once -- and merged locally with merge_states / merge_state_in_place.  The merge
    operator is associative and commutative but not a sum, so a reduce collective cannot express it.

`exchange_partial_states` is backend-agnostic torch.distributed code (gloo on CPU in the tests).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """[begin, end) of the contiguous slice of `total` items owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(total, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def pack_states(v: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """[n, H, D] 16-bit values + [n, H] f32 log-sum-exp -> one uint8 buffer [n, H * (2 D + 4)] so that a
    single collective moves both."""
    n, h, d = v.shape
    vb = v.contiguous().view(torch.uint8).reshape(n, h * d * v.element_size())
    sb = s.to(torch.float32).contiguous().view(torch.uint8).reshape(n, h * 4)
    return torch.cat([vb, sb], dim=1).contiguous()


def unpack_states(buf: torch.Tensor, h: int, d: int, dtype: torch.dtype):
    n = buf.shape[0]
    esz = torch.empty((), dtype=dtype).element_size()
    vb = buf[:, : h * d * esz].contiguous().view(dtype).reshape(n, h, d)
    sb = buf[:, h * d * esz:].contiguous().view(torch.float32).reshape(n, h)
    return vb, sb


def exchange_partial_states(
    v_all: torch.Tensor, s_all: torch.Tensor, group: Optional[dist.ProcessGroup] = None
) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-to-all of prefix partial states.

    v_all [B_total, H, D], s_all [B_total, H]: this rank's partial state (over ITS prefix shard) for every
    query of the global batch, ordered by owner rank (`shard_range`).  Returns
    (v [B_local, world, H, D], s [B_local, world, H]): for each of this rank's own queries, the partial
    states computed by every rank -- the input layout of `merge_states`.
    """
    initialized = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if initialized else 1
    rank = dist.get_rank(group) if initialized else 0
    b_total, h, d = v_all.shape
    send = pack_states(v_all, s_all)
    in_splits = [shard_range(b_total, world, r)[1] - shard_range(b_total, world, r)[0] for r in range(world)]
    lo, hi = shard_range(b_total, world, rank)
    b_local = hi - lo
    out_splits = [b_local] * world
    recv = torch.empty(b_local * world, send.shape[1], dtype=torch.uint8, device=send.device)
    if world == 1:
        recv.copy_(send)
    else:
        dist.all_to_all_single(recv, send, output_split_sizes=out_splits, input_split_sizes=in_splits,
                               group=group)
    v, s = unpack_states(recv, h, d, v_all.dtype)
    # recv is [world, B_local, ...] (grouped by sender) -> [B_local, world, ...]
    v = v.reshape(world, b_local, h, d).transpose(0, 1).contiguous()
    s = s.reshape(world, b_local, h).transpose(0, 1).contiguous()
    return v, s


def sharded_shared_prefix_decode(
    q_local: torch.Tensor,
    prefix_attend: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]],
    unique_attend: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]],
    merge_states_fn: Callable,
    merge_state_fn: Callable,
    group: Optional[dist.ProcessGroup] = None,
) -> torch.Tensor:
    """One decode step with a sequence-sharded shared prefix.

    q_local [B_local, H, D]: this rank's queries.
    prefix_attend(q_all) -> (v, s): attention of ALL queries over this rank's prefix shard (base-2 lse).
    unique_attend(q_local) -> (v, s): attention of the local queries over their unique suffixes.
    merge_states_fn / merge_state_fn: flashinfer.merge_states / flashinfer.merge_state on the GPU.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        sizes = [torch.empty_like(q_local) for _ in range(world)]
        dist.all_gather(sizes, q_local.contiguous(), group=group)  # equal batch slices per rank
        q_all = torch.cat(sizes, dim=0)
    else:
        q_all = q_local
    v_p, s_p = prefix_attend(q_all)
    v_u, s_u = unique_attend(q_local)
    if world > 1:
        v_x, s_x = exchange_partial_states(v_p, s_p, group)
        v_pref, s_pref = merge_states_fn(v_x, s_x)
    else:
        v_pref, s_pref = v_p, s_p
    v, _ = merge_state_fn(v_pref, s_pref, v_u, s_u)
    return v

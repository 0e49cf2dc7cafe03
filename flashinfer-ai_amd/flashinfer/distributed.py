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

The step is latency-bound (tens of microseconds), so `SharedPrefixExchange` owns every buffer of the step
(gathered queries, packed send / receive rows, the unpacked merge inputs) and a step allocates nothing: one
`all_gather_into_tensor`, one `all_to_all_single`, four small copy kernels.  Ranks may own batch slices of
different sizes (`shard_range` slices differ by one when the batch does not divide): every collective runs on
slices padded to the largest one, so all messages have the same size, and the padding rows are dropped.

Backend-agnostic torch.distributed code (gloo on CPU in the tests).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

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


def _world_rank(group) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


class SharedPrefixExchange:
    """Pre-allocated buffers and the two collectives of one shared-prefix decode step.

    batch_sizes[r] = number of requests rank r owns (what the serving scheduler decided; equal sizes when
    omitted).  All collectives run on slices padded to b_max = max(batch_sizes).

    Layouts (row = one query):
      q_all            [world * b_max, H, D]          gathered queries, rank-major, padding rows zero
      send / recv      [world * b_max, H * (D * esize + 4)] bytes: v row | s row of one query
      merge inputs     v [b_max, world, H, D], s [b_max, world, H]   (the layout merge_states takes)
    """

    def __init__(self, num_heads: int, head_dim: int, dtype: torch.dtype, device, local_batch: int,
                 batch_sizes: Optional[Sequence[int]] = None, group: Optional[dist.ProcessGroup] = None,
                 always_collective: bool = False):
        self.group = group
        # run the collectives even with a single rank (exercises the RCCL calls on a one-GPU box)
        self.always_collective = always_collective and dist.is_available() and dist.is_initialized()
        self.world, self.rank = _world_rank(group)
        if batch_sizes is None:
            batch_sizes = [local_batch] * self.world
        batch_sizes = [int(b) for b in batch_sizes]
        if len(batch_sizes) != self.world:
            raise ValueError(f"batch_sizes has {len(batch_sizes)} entries for {self.world} ranks")
        if batch_sizes[self.rank] != local_batch:
            raise ValueError(f"rank {self.rank} holds {local_batch} queries but batch_sizes says {batch_sizes[self.rank]}")
        self.batch_sizes = batch_sizes
        self.b_local = local_batch
        self.b_max = max(batch_sizes) if batch_sizes else 0
        self.h, self.d, self.dtype = num_heads, head_dim, dtype
        esz = torch.empty((), dtype=dtype).element_size()
        self.v_bytes = num_heads * head_dim * esz
        self.row_bytes = self.v_bytes + num_heads * 4
        n = self.world * self.b_max
        self.q_pad = torch.zeros(self.b_max, num_heads, head_dim, dtype=dtype, device=device)
        self.q_all = torch.zeros(n, num_heads, head_dim, dtype=dtype, device=device)
        self.send = torch.zeros(n, self.row_bytes, dtype=torch.uint8, device=device)
        self.recv = torch.zeros(n, self.row_bytes, dtype=torch.uint8, device=device)
        # typed views of the packed rows (no copies): the v part and the s part of every row
        self._send_v = self.send[:, : self.v_bytes].view(dtype).view(n, num_heads, head_dim)
        self._send_s = self.send[:, self.v_bytes:].view(torch.float32)
        rv = self.recv[:, : self.v_bytes].view(dtype).view(self.world, self.b_max, num_heads, head_dim)
        rs = self.recv[:, self.v_bytes:].view(torch.float32).view(self.world, self.b_max, num_heads)
        self._recv_v = rv.transpose(0, 1)  # [b_max, world, H, D] view of the received rows
        self._recv_s = rs.transpose(0, 1)
        self.v_merge = torch.empty(self.b_max, self.world, num_heads, head_dim, dtype=dtype, device=device)
        self.s_merge = torch.empty(self.b_max, self.world, num_heads, dtype=torch.float32, device=device)

    @property
    def bytes_sent_per_step(self) -> int:
        """payload this rank sends to its peers in the all-to-all (its own slice stays local)."""
        return (self.world - 1) * self.b_max * self.row_bytes

    def valid_rows(self) -> torch.Tensor:
        """index of every real (non-padding) row of the rank-major padded layout, in global batch order."""
        idx = [r * self.b_max + i for r in range(self.world) for i in range(self.batch_sizes[r])]
        return torch.tensor(idx, dtype=torch.long)

    def gather_queries(self, q_local: torch.Tensor) -> torch.Tensor:
        """All ranks' queries in the padded rank-major layout [world * b_max, H, D]."""
        if q_local.shape != (self.b_local, self.h, self.d):
            raise ValueError(f"q_local must be {(self.b_local, self.h, self.d)}, got {tuple(q_local.shape)}")
        if self.world == 1 and not self.always_collective:
            return q_local
        if self.b_local == self.b_max:
            src = q_local.contiguous()
        else:
            self.q_pad[: self.b_local].copy_(q_local)
            src = self.q_pad
        dist.all_gather_into_tensor(self.q_all, src, group=self.group)
        return self.q_all

    def exchange(self, v_all: torch.Tensor, s_all: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """v_all [world * b_max, H, D], s_all [world * b_max, H]: this rank's partial state (over ITS prefix
        shard) for every query in the padded layout.  Returns (v [b_local, world, H, D], s [b_local, world, H]):
        for each of this rank's own queries the partial states computed by every rank."""
        self._send_v.copy_(v_all)
        self._send_s.copy_(s_all)
        if self.world == 1 and not self.always_collective:
            self.recv.copy_(self.send)
        else:
            dist.all_to_all_single(self.recv, self.send, group=self.group)  # equal splits of b_max rows
        self.v_merge.copy_(self._recv_v)
        self.s_merge.copy_(self._recv_s)
        return self.v_merge[: self.b_local], self.s_merge[: self.b_local]


def exchange_partial_states(
    v_all: torch.Tensor, s_all: torch.Tensor, group: Optional[dist.ProcessGroup] = None
) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-to-all of prefix partial states for a batch split by `shard_range` (sizes may differ by one).

    v_all [B_total, H, D], s_all [B_total, H]: this rank's partial state (over ITS prefix shard) for every
    query of the global batch, ordered by owner rank.  Returns (v [B_local, world, H, D], s [B_local, world, H]),
    the input layout of `merge_states`.  One-shot form (allocates); steady-state loops use
    `SharedPrefixExchange`.
    """
    world, rank = _world_rank(group)
    b_total, h, d = v_all.shape
    sizes = [shard_range(b_total, world, r)[1] - shard_range(b_total, world, r)[0] for r in range(world)]
    ex = SharedPrefixExchange(h, d, v_all.dtype, v_all.device, sizes[rank], sizes, group)
    rows = ex.valid_rows().to(v_all.device)
    v_pad = torch.zeros(world * ex.b_max, h, d, dtype=v_all.dtype, device=v_all.device)
    s_pad = torch.zeros(world * ex.b_max, h, dtype=torch.float32, device=v_all.device)
    v_pad[rows] = v_all
    s_pad[rows] = s_all.to(torch.float32)
    v, s = ex.exchange(v_pad, s_pad)
    return v.clone(), s.clone()


def sharded_shared_prefix_decode(
    q_local: torch.Tensor,
    prefix_attend: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]],
    unique_attend: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]],
    merge_states_fn: Callable,
    merge_state_fn: Callable,
    group: Optional[dist.ProcessGroup] = None,
    batch_sizes: Optional[Sequence[int]] = None,
    exchange: Optional[SharedPrefixExchange] = None,
    q_all: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """One decode step with a sequence-sharded shared prefix.

    q_local [B_local, H, D]: this rank's queries.
    prefix_attend(q_all) -> (v, s): attention of ALL queries (padded rank-major layout
        [world * max(batch_sizes), H, D]; padding rows are zero and their results are dropped) over this
        rank's prefix shard, base-2 lse.
    unique_attend(q_local) -> (v, s): attention of the local queries over their unique suffixes.
    merge_states_fn / merge_state_fn: flashinfer.merge_states / flashinfer.merge_state on the GPU.
    batch_sizes: requests per rank when they differ (ranks must pass the same list); omitted = every rank
        holds as many queries as this one.  A mismatch cannot be detected without a collective: with
        unequal slices and no batch_sizes the gather fails inside the backend.
    exchange: a `SharedPrefixExchange` built once for the loop (buffers are reused); built here otherwise.
    q_all: every rank's queries in the padded rank-major layout [world * max(batch_sizes), H, D], when the
        caller already holds them on every rank (a tensor-parallel model replicates the activations that
        produce q, SURVEY.md 8e "or replicate q upstream"): the query all-gather is skipped and the step has
        ONE collective, the all-to-all of the partial states.
    """
    world, _ = _world_rank(group)
    if world == 1 and exchange is None:
        v_p, s_p = prefix_attend(q_local)
        v_u, s_u = unique_attend(q_local)
        v, _ = merge_state_fn(v_p, s_p, v_u, s_u)
        return v
    if exchange is None:
        exchange = SharedPrefixExchange(q_local.shape[1], q_local.shape[2], q_local.dtype, q_local.device,
                                        q_local.shape[0], batch_sizes, group)
    if q_all is None:
        q_all = exchange.gather_queries(q_local)
    elif q_all.shape != (exchange.world * exchange.b_max, exchange.h, exchange.d):
        raise ValueError(f"q_all must be {(exchange.world * exchange.b_max, exchange.h, exchange.d)}, "
                         f"got {tuple(q_all.shape)}")
    v_p, s_p = prefix_attend(q_all)
    v_u, s_u = unique_attend(q_local)
    v_x, s_x = exchange.exchange(v_p, s_p)
    v_pref, s_pref = merge_states_fn(v_x, s_x)
    v, _ = merge_state_fn(v_pref, s_pref, v_u, s_u)
    return v

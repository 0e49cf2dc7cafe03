"""CPU, world_size 2, gloo: the N>1 path -- batch sharding and the all-to-all exchange of shared-prefix
partial states -- with the oracle standing in for the GPU kernels.  The communication code under test is
the code the GPU path runs (flashinfer/distributed.py); only the attention / merge callables are swapped."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import attention_ref as R


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret, b_total=6):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flashinfer import distributed as D

        torch.manual_seed(0)  # same global problem on every rank
        hq, hkv, d, prefix_len, suffix = 4, 2, 32, 40, 5
        q = torch.randn(b_total, hq, d).half()
        k_pre, v_pre = torch.randn(prefix_len, hkv, d).half(), torch.randn(prefix_len, hkv, d).half()
        k_uni = torch.randn(b_total, suffix, hkv, d).half()
        v_uni = torch.randn(b_total, suffix, hkv, d).half()
        lo, hi = D.shard_range(b_total, world, rank)
        plo, phi = D.shard_range(prefix_len, world, rank)

        def prefix_attend(q_all):
            o, s = R.attention_ref(q_all.float(), k_pre[plo:phi].float(), v_pre[plo:phi].float())
            return o.half(), s.float()

        def unique_attend(q_loc):
            outs = [R.attention_ref(q_loc[i:i + 1].float(), k_uni[lo + i].float(), v_uni[lo + i].float())
                    for i in range(q_loc.shape[0])]
            return torch.cat([o for o, _ in outs]).half(), torch.cat([s for _, s in outs]).float()

        def merge_states_fn(v, s):
            vm, sm = R.merge_states_ref(v.float(), s)
            return vm.half(), sm.float()

        def merge_state_fn(va, sa, vb, sb):
            vm, sm = R.merge_state_ref(va.float(), sa, vb.float(), sb)
            return vm.half(), sm.float()

        sizes = [D.shard_range(b_total, world, r)[1] - D.shard_range(b_total, world, r)[0] for r in range(world)]
        ex = D.SharedPrefixExchange(hq, d, torch.float16, "cpu", hi - lo, sizes)
        for _ in range(2):  # the second step reuses every buffer
            out = D.sharded_shared_prefix_decode(q[lo:hi], prefix_attend, unique_attend, merge_states_fn,
                                                 merge_state_fn, batch_sizes=sizes, exchange=ex)
        assert out.shape[0] == hi - lo
        # one-shot form on the unpadded global batch
        v_all, s_all = prefix_attend(q)
        vx, sx = D.exchange_partial_states(v_all, s_all)
        vm, _ = merge_states_fn(vx, sx)
        full, _ = R.attention_ref(q[lo:hi].float(), k_pre.float(), v_pre.float())
        assert (vm.float() - full.float()).abs().max().item() < 5e-3
        # reference: plain attention over [prefix | unique suffix] per request
        ref = torch.cat([
            R.attention_ref(q[i:i + 1].float(), torch.cat([k_pre, k_uni[i]]).float(),
                            torch.cat([v_pre, v_uni[i]]).float())[0] for i in range(lo, hi)])
        err = (out.float() - ref.float()).abs().max().item()
        ret[rank] = err
        # pack/unpack round trip is bit exact
        v, s = torch.randn(3, hq, d).bfloat16(), torch.randn(3, hq)
        v2, s2 = D.unpack_states(D.pack_states(v, s), hq, d, torch.bfloat16)
        assert torch.equal(v, v2) and torch.equal(s, s2)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("b_total", [6, 7])  # 7: the ranks own 4 and 3 requests (padded collectives)
def test_sharded_shared_prefix_decode_world2(b_total):
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret, b_total), nprocs=world, join=True)
        assert len(ret) == world
        for r in range(world):
            assert ret[r] < 5e-3, ret[r]  # fp16 rounding of the exchanged states


def test_shard_range_partitions_exactly():
    from flashinfer.distributed import shard_range

    for total in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1

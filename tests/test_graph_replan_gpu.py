"""Captured run() replayed after plan() chose a DIFFERENT split: the kv chunk size and the number of partial
states change between the plans, so every value the kernels take from the plan must come from the workspaces
plan() rewrites, not from launch arguments frozen into the graph (the reference reads *kv_chunk_size_ptr for
this reason, decode.cuh:424, prefill.cuh:2058; serving pattern: capture once, plan every step, replay)."""
import pytest
import torch

from oracle import attention_ref as R
from test_decode_gpu import make_paged

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_decode_graph_replay_after_replan_with_other_chunk_size():
    import flashinfer

    hq, hkv, d, ps, b = 8, 2, 128, 16, 4
    # 2 kv heads x 2048 resident waves: plan A fits 128-token chunks, plan B (3x the pages) needs longer ones
    cases = [[20000, 18000, 16000, 12001], [60000, 50000, 45000, 40017], [333, 70000, 16, 52000]]
    max_pages = max(sum(-(-l // ps) for l in c) for c in cases) + 8
    data = [make_paged(b, c, ps, hkv, d, torch.float16, "NHD", seed=90 + i,
                       extra_pages=max_pages - sum(-(-l // ps) for l in c)) for i, c in enumerate(cases)]
    ws = torch.zeros(128 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.CUDAGraphBatchDecodeWithPagedKVCacheWrapper(
        ws, torch.zeros(b + 1, dtype=torch.int32, device=DEV), torch.zeros(max_pages, dtype=torch.int32, device=DEV),
        torch.zeros(b, dtype=torch.int32, device=DEV), "NHD")
    cache_dev = torch.empty_like(data[0][0], device=DEV)
    q_dev = torch.zeros(b, hq, d, dtype=torch.float16, device=DEV)
    out = torch.zeros_like(q_dev)
    lse = torch.zeros(b, hq, dtype=torch.float32, device=DEV)

    def plan(i):
        cache, indptr, indices, last = data[i]
        w.plan(indptr, indices, last, hq, hkv, d, ps, q_data_type=torch.float16, kv_data_type=torch.float16)
        cache_dev.copy_(cache)
        torch.manual_seed(100 + i)
        q = torch.randn(b, hq, d).half()
        q_dev.copy_(q)
        return q

    plan(0)
    chunk_a = w._plan_info[10]
    w.run(q_dev, cache_dev, out=out, lse=lse, return_lse=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        w.run(q_dev, cache_dev, out=out, lse=lse, return_lse=True)
    chunks = set()
    for i in (1, 0, 2, 1):
        q = plan(i)
        chunks.add(w._plan_info[10])
        out.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        cache, indptr, indices, last = data[i]
        o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), "NHD", indptr, indices, last)
        torch.testing.assert_close(out.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)
    assert len(chunks | {chunk_a}) >= 2, "the plans must differ in chunk size for this test to mean anything"


def test_prefill_graph_replay_after_replan_with_other_chunk_size_and_more_partials():
    import flashinfer

    hq, hkv, d, ps, b = 8, 2, 128, 16, 2
    rows = 96
    cases = [([1000, 300], [40, 56]), ([20000, 15000], [90, 6]), ([70, 2000], [6, 90]), ([9000, 31000], [48, 48])]
    max_pages = max(sum(-(-l // ps) for l in kv) for kv, _ in cases) + 8
    data = []
    for seed, (kv_lens, qo_lens) in enumerate(cases):
        cache, indptr, indices, last = make_paged(b, kv_lens, ps, hkv, d, torch.float16, "NHD", seed=170 + seed,
                                                  extra_pages=max_pages - sum(-(-l // ps) for l in kv_lens))
        data.append((cache, indptr, indices, last, qo_lens))
    ws = torch.zeros(256 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(
        ws, "NHD", use_cuda_graph=True, qo_indptr_buf=torch.zeros(b + 1, dtype=torch.int32, device=DEV),
        paged_kv_indptr_buf=torch.zeros(b + 1, dtype=torch.int32, device=DEV),
        paged_kv_indices_buf=torch.zeros(max_pages, dtype=torch.int32, device=DEV),
        paged_kv_last_page_len_buf=torch.zeros(b, dtype=torch.int32, device=DEV))
    cache_dev = torch.empty_like(data[0][0], device=DEV)
    q_dev = torch.zeros(rows, hq, d, dtype=torch.float16, device=DEV)
    out = torch.zeros_like(q_dev)
    lse = torch.zeros(rows, hq, dtype=torch.float32, device=DEV)

    def plan(i):
        cache, indptr, indices, last, qo_lens = data[i]
        qo_indptr = torch.tensor([0] + list(torch.tensor(qo_lens).cumsum(0)), dtype=torch.int32)
        w.plan(qo_indptr, indptr, indices, last, hq, hkv, d, ps, causal=True, max_token_per_sequence=rows // b)
        cache_dev.copy_(cache)
        torch.manual_seed(i)
        q = torch.randn(rows, hq, d).half()
        q_dev.copy_(q)
        return q, qo_indptr

    plan(0)
    offsets = (w._plan_info[10], w._plan_info[11], w._plan_info[0])  # v_off, s_off, padded items
    w.run(q_dev, cache_dev, out=out, lse=lse, return_lse=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        w.run(q_dev, cache_dev, out=out, lse=lse, return_lse=True)
    chunks = {w._plan_info[9]}
    for i in (1, 0, 3, 2, 1):
        q, qo_indptr = plan(i)
        chunks.add(w._plan_info[9])
        # what a captured launch froze must be the same for every plan of this wrapper
        assert (w._plan_info[10], w._plan_info[11], w._plan_info[0]) == offsets
        out.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        cache, indptr, indices, last, qo_lens = data[i]
        n = sum(qo_lens)
        o_ref, lse_ref = R.batch_prefill_ref(q[:n].float(), qo_indptr, cache.float(), "NHD", indptr, indices, last,
                                             causal=True)
        torch.testing.assert_close(out[:n].float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(lse[:n].cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)
    assert len(chunks) >= 2, "the plans must differ in chunk size for this test to mean anything"


def test_fp8_prefill_run_without_scales_is_graph_capturable():
    """fp8 q/k/v with no scale arguments: run() must not create tensors on the host (a pageable copy is illegal
    while capturing and syncs otherwise); a missing scale is a NULL pointer (= 1)."""
    import flashinfer

    hq, hkv, d, ps = 8, 2, 128, 16
    kv_lens, qo_lens = [300, 64], [200, 64]
    cache, indptr, indices, last = make_paged(2, kv_lens, ps, hkv, d, torch.float8_e4m3fn, "NHD", seed=5)
    torch.manual_seed(6)
    q8 = torch.randn(sum(qo_lens), hq, d).to(torch.float8_e4m3fn)
    qo_indptr = torch.tensor([0] + list(torch.tensor(qo_lens).cumsum(0)), dtype=torch.int32)
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, "NHD")
    w.plan(qo_indptr.to(DEV), indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, ps, causal=True,
           q_data_type=torch.float8_e4m3fn, kv_data_type=torch.float8_e4m3fn, o_data_type=torch.float16)
    qd, cd = q8.to(DEV), cache.to(DEV)
    out = torch.zeros(sum(qo_lens), hq, d, dtype=torch.float16, device=DEV)
    w.run(qd, cd, out=out)
    torch.cuda.synchronize()
    eager = out.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        w.run(qd, cd, out=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    off = 0
    one_q, one_kv = torch.ones(hq), torch.ones(hkv)
    for bi in range(2):
        pages = indices[int(indptr[bi]):int(indptr[bi + 1])].long()
        kk = cache[pages, 0].reshape(-1, hkv, d)[: kv_lens[bi]]
        vv = cache[pages, 1].reshape(-1, hkv, d)[: kv_lens[bi]]
        o_ref, _ = R.fp8_attention_ref(q8[off:off + qo_lens[bi]], kk, vv, one_q, one_kv, one_kv, causal=True)
        torch.testing.assert_close(out[off:off + qo_lens[bi]].float().cpu(), o_ref.float(), rtol=5e-2, atol=5e-2)
        off += qo_lens[bi]

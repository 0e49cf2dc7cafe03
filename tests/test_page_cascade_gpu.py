"""GPU: page-table append kernels and the cascade wrappers.
ref: tests/attention/test_page.py:8, tests/attention/test_shared_prefix_kernels.py:56-226."""
import pytest
import torch

from oracle import attention_ref as R
from test_decode_gpu import make_paged

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_get_batch_indices_positions_doc_example():
    import flashinfer

    # ref: flashinfer/page.py:190-200
    append_indptr = torch.tensor([0, 1, 3, 6, 10], dtype=torch.int32, device=DEV)
    seq_lens = torch.tensor([5, 5, 5, 5])
    bi, pos = flashinfer.get_batch_indices_positions(append_indptr, seq_lens, 10)
    assert bi.tolist() == [0, 1, 1, 2, 2, 2, 3, 3, 3, 3]
    assert pos.tolist() == [4, 3, 4, 2, 3, 4, 1, 2, 3, 4]


@pytest.mark.parametrize("layout", ["NHD", "HND"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.float8_e4m3fn])
def test_append_paged_kv_cache_then_decode(layout, dtype):
    """Build a cache by appending in two steps (prefill chunk, then one decode token per request), check
    the cache contents bit for bit against a host scatter, then run decode on it."""
    import flashinfer

    torch.manual_seed(0)
    hkv, hq, d, ps = 2, 4, 128, 16
    lens0 = [45, 8, 25, 22]
    b = len(lens0)
    max_pages = 32
    shape = (max_pages, 2, ps, hkv, d) if layout == "NHD" else (max_pages, 2, hkv, ps, d)
    cache = torch.zeros(shape, dtype=torch.float16, device=DEV).to(dtype)
    host = torch.zeros(max_pages, 2, ps, hkv, d)  # NHD mirror
    perm = torch.randperm(max_pages)
    final = [l + 1 for l in lens0]
    pages = [-(-l // ps) for l in final]
    indptr = torch.tensor([0] + list(torch.tensor(pages).cumsum(0)), dtype=torch.int32)
    indices = perm[: int(indptr[-1])].to(torch.int32)
    seq_so_far = [0] * b
    for step_lens in (lens0, [1] * b):
        nnz = sum(step_lens)
        k_new = torch.randn(nnz, hkv, d).half().to(dtype)
        v_new = torch.randn(nnz, hkv, d).half().to(dtype)
        seq_after = [s + a for s, a in zip(seq_so_far, step_lens)]
        append_indptr = torch.tensor([0] + list(torch.tensor(step_lens).cumsum(0)), dtype=torch.int32, device=DEV)
        bi, pos = flashinfer.get_batch_indices_positions(append_indptr, torch.tensor(seq_after), nnz)
        last = torch.tensor([(l - 1) % ps + 1 for l in seq_after], dtype=torch.int32)
        flashinfer.append_paged_kv_cache(k_new.to(DEV), v_new.to(DEV), bi, pos, cache, indices.to(DEV),
                                         indptr.to(DEV), last.to(DEV), kv_layout=layout)
        t = 0
        for r in range(b):
            for j in range(step_lens[r]):
                p = seq_so_far[r] + j
                pg = int(indices[int(indptr[r]) + p // ps])
                host[pg, 0, p % ps] = k_new[t].float()
                host[pg, 1, p % ps] = v_new[t].float()
                t += 1
        seq_so_far = seq_after
    got = cache.float().cpu()
    if layout == "HND":
        got = got.transpose(2, 3)
    assert torch.equal(got, host)
    # decode over the appended cache
    last = torch.tensor([(l - 1) % ps + 1 for l in final], dtype=torch.int32)
    q = torch.randn(b, hq, d).half()
    ws = torch.zeros(16 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, layout)
    w.plan(indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, ps, q_data_type=torch.float16, kv_data_type=dtype)
    o = w.run(q.to(DEV), cache)
    o_ref, _ = R.batch_decode_ref(q.float(), host, "NHD", indptr, indices, last)
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)


def _two_level_problem(batch, prefix_len, suffix_lens, hq, hkv, d, ps, seed):
    """One shared prefix + unique suffixes in a unified page table (ref: docs/tutorials/kv_layout.rst:182-201)."""
    g = torch.Generator().manual_seed(seed)
    ppages = prefix_len // ps
    upages = [-(-l // ps) for l in suffix_lens]
    total = ppages + sum(upages)
    cache = torch.randn(total + 2, 2, ps, hkv, d, generator=g).half()
    perm = torch.randperm(total + 2, generator=g).to(torch.int32)
    shared_indices = perm[:ppages]
    unique_indices = perm[ppages:total]
    unique_indptr = torch.tensor([0] + list(torch.tensor(upages).cumsum(0)), dtype=torch.int32)
    unique_last = torch.tensor([(l - 1) % ps + 1 for l in suffix_lens], dtype=torch.int32)
    q = torch.randn(batch, hq, d, generator=g).half()
    return cache, shared_indices, unique_indices, unique_indptr, unique_last, q


def test_multi_level_cascade_matches_flat_attention():
    import flashinfer

    batch, prefix_len, hq, hkv, d, ps = 7, 256, 8, 2, 128, 16
    suffix_lens = [5, 33, 16, 1, 70, 48, 17]
    cache, sh_idx, un_idx, un_indptr, un_last, q = _two_level_problem(batch, prefix_len, suffix_lens, hq, hkv, d, ps, 3)
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.MultiLevelCascadeAttentionWrapper(2, ws, "NHD")
    shared_qo_indptr = torch.tensor([0, batch], dtype=torch.int32)
    unique_qo_indptr = torch.arange(batch + 1, dtype=torch.int32)
    shared_kv_indptr = torch.tensor([0, prefix_len // ps], dtype=torch.int32)
    shared_last = torch.tensor([ps], dtype=torch.int32)
    w.plan([shared_qo_indptr.to(DEV), unique_qo_indptr.to(DEV)], [shared_kv_indptr.to(DEV), un_indptr.to(DEV)],
           [sh_idx.to(DEV), un_idx.to(DEV)], [shared_last.to(DEV), un_last.to(DEV)], hq, hkv, d, ps,
           q_data_type=torch.float16)
    o = w.run(q.to(DEV), cache.to(DEV))
    # flat baseline: every request attends [prefix pages | its own pages]
    for r in range(batch):
        pages = torch.cat([sh_idx, un_idx[int(un_indptr[r]):int(un_indptr[r + 1])]]).long()
        kv = cache[pages].float()
        k = kv[:, 0].reshape(-1, hkv, d)[: prefix_len + suffix_lens[r]]
        v = kv[:, 1].reshape(-1, hkv, d)[: prefix_len + suffix_lens[r]]
        o_ref, _ = R.attention_ref(q[r:r + 1].float(), k, v)
        torch.testing.assert_close(o[r:r + 1].float().cpu(), o_ref.float(), rtol=2e-3, atol=2e-3)


def test_shared_prefix_decode_wrapper_and_sharded_path_world1():
    import flashinfer
    from flashinfer import distributed as D

    batch, prefix_len, hq, hkv, d, ps = 5, 128, 8, 2, 128, 16
    suffix_lens = [9, 16, 31, 2, 64]
    cache, sh_idx, un_idx, un_indptr, un_last, q = _two_level_problem(batch, prefix_len, suffix_lens, hq, hkv, d, ps, 5)
    k_shared = cache[sh_idx.long(), 0].reshape(-1, hkv, d).contiguous()
    v_shared = cache[sh_idx.long(), 1].reshape(-1, hkv, d).contiguous()
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchDecodeWithSharedPrefixPagedKVCacheWrapper(ws, "NHD")
    w.begin_forward(un_indptr.to(DEV), un_idx.to(DEV), un_last.to(DEV), hq, hkv, d, ps, data_type="float16")
    o = w.forward(q.to(DEV), k_shared.to(DEV), v_shared.to(DEV), cache.to(DEV))
    refs = []
    for r in range(batch):
        pages = un_idx[int(un_indptr[r]):int(un_indptr[r + 1])].long()
        ku = cache[pages, 0].reshape(-1, hkv, d)[: suffix_lens[r]]
        vu = cache[pages, 1].reshape(-1, hkv, d)[: suffix_lens[r]]
        refs.append(R.attention_ref(q[r:r + 1].float(), torch.cat([k_shared, ku]).float(), torch.cat([v_shared, vu]).float())[0])
    ref = torch.cat(refs)
    torch.testing.assert_close(o.float().cpu(), ref.float(), rtol=2e-3, atol=2e-3)

    # the sharded (C5) code path with world_size 1: same kernels, exchange degenerates to a copy
    dw = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, "NHD")
    dw.plan(un_indptr.to(DEV), un_idx.to(DEV), un_last.to(DEV), hq, hkv, d, ps, q_data_type=torch.float16)
    cd, ks, vs = cache.to(DEV), k_shared.to(DEV), v_shared.to(DEV)
    out = D.sharded_shared_prefix_decode(
        q.to(DEV),
        lambda q_all: flashinfer.single_prefill_with_kv_cache(q_all, ks, vs, return_lse=True),
        lambda q_loc: dw.run(q_loc, cd, return_lse=True),
        flashinfer.merge_states, flashinfer.merge_state)
    torch.testing.assert_close(out.float().cpu(), ref.float(), rtol=2e-3, atol=2e-3)
    v_x, s_x = D.exchange_partial_states(o, torch.zeros(batch, hq, device=DEV))
    assert v_x.shape == (batch, 1, hq, d) and torch.equal(v_x[:, 0], o)

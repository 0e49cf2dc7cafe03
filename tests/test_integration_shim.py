"""The boundary from the REFERENCE's side: flashinfer/compat.py provides module getters with the reference's
own FFI signatures (get_batch_decode_module / get_batch_prefill_module / get_cascade_module / get_page_module /
get_gemm_sm100_module).  These tests make the call sequences the reference's operator layer makes -- same
positional order, same host / device placement of every tensor (flashinfer/decode.py:1079-1095, 1338-1366;
prefill.py:1884-1908, 2148-2199; cascade.py:87-100; page.py:411-424; gemm.py:2791-2806) -- and compare with the
oracle.  A maintainer swapping only the getters would exercise exactly this."""
import math

import pytest
import torch

from oracle import attention_ref as R
from oracle import gemm_ref as G
from test_decode_gpu import make_paged

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _workspaces():
    # what the reference's wrappers own (flashinfer/decode.py:726-734)
    float_ws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
    int_ws = torch.empty(8 << 20, dtype=torch.uint8, device=DEV)
    pinned = torch.empty(8 << 20, dtype=torch.uint8, pin_memory=True, device="cpu")
    return float_ws, int_ws, pinned


@pytest.mark.parametrize("layout,layout_code", [("NHD", 0), ("HND", 1)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_batch_decode_module_reference_call_sequence(layout, layout_code, dtype):
    from flashinfer import compat
    from flashinfer.utils import _get_cache_alibi_slopes_buf

    hq, hkv, d, ps = 32, 8, 128, 16
    kv_lens = [54, 4000, 1, 700, 33]
    b = len(kv_lens)
    cache, indptr, indices, last = make_paged(b, kv_lens, ps, hkv, d, dtype, layout, seed=3)
    torch.manual_seed(4)
    q = torch.randn(b, hq, d).to(dtype)
    float_ws, int_ws, pinned = _workspaces()
    # decode.py:1068-1078
    mod = compat.get_batch_decode_module(dtype, dtype, dtype, torch.int32, d, d, 0, False, False)
    # decode.py:1079-1095 (indptr_host on the CPU, empty typed tensors carry the dtypes)
    plan_info = mod.plan(float_ws, int_ws, pinned, indptr, b, hq, hkv, ps, False, -1, 0.0, d, d,
                         torch.empty(0, dtype=dtype), torch.empty(0, dtype=dtype))
    assert isinstance(plan_info, list) and all(isinstance(x, int) for x in plan_info)
    cd = cache.to(DEV)
    k_cache, v_cache = cd[:, 0], cd[:, 1]  # _unpack_paged_kv_cache, flashinfer/utils.py:149-169
    qd = q.to(DEV)
    out = torch.empty_like(qd)
    lse = torch.empty(b, hq, dtype=torch.float32, device=DEV)
    sm_scale, rope_scale, rope_theta = 1.0 / math.sqrt(d), 1.0, 1e4
    # decode.py:1338-1366 + the Python shim's inversion of the rope parameters (decode.py:264-268)
    mod.run(float_ws, int_ws, plan_info, qd, k_cache, v_cache, indptr.to(DEV), indices.to(DEV), last.to(DEV), out, lse,
            layout_code, -1, False, _get_cache_alibi_slopes_buf(hq, qd.device), 0.0, sm_scale, 1.0 / rope_scale,
            1.0 / rope_theta)
    o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), layout, indptr, indices, last)
    tol = dict(rtol=1e-3 + 2.0 ** -8, atol=2e-3) if dtype == torch.bfloat16 else dict(rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out.float().cpu(), o_ref.float(), **tol)
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)
    # no lse requested: maybe_lse = None (decode.py:1262-1275)
    out2 = torch.empty_like(qd)
    mod.run(float_ws, int_ws, plan_info, qd, k_cache, v_cache, indptr.to(DEV), indices.to(DEV), last.to(DEV), out2, None,
            layout_code, -1, False, None, 0.0, sm_scale, 1.0, 1e-4)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("causal", [False, True])
def test_batch_prefill_module_reference_call_sequence(causal):
    from flashinfer import compat

    hq, hkv, d, ps = 8, 2, 128, 16
    kv_lens, qo_lens = [54, 300, 1, 129], [37, 300, 1, 17]
    b = len(kv_lens)
    cache, indptr, indices, last = make_paged(b, kv_lens, ps, hkv, d, torch.float16, "NHD", seed=21)
    torch.manual_seed(1)
    q = torch.randn(sum(qo_lens), hq, d).half()
    qo_indptr = torch.tensor([0] + list(torch.tensor(qo_lens).cumsum(0)), dtype=torch.int32)
    kv_len_arr = torch.tensor(kv_lens, dtype=torch.int32)
    float_ws, int_ws, pinned = _workspaces()
    mod = compat.get_batch_prefill_module("fa2", torch.float16, torch.float16, torch.float16, torch.int32, d, d, 0,
                                          False, False, False)
    # prefill.py:1884-1908: host index tensors, fa2 appends fixed_split_size / disable_split_kv
    plan_info = mod.plan(float_ws, int_ws, pinned, qo_indptr, indptr, kv_len_arr, int(qo_indptr[-1]), b, hq, hkv, ps,
                         False, d, d, causal, -1, -1, False)
    cd, qd = cache.to(DEV), q.to(DEV)
    out = torch.empty_like(qd)
    lse = torch.empty(qd.shape[0], hq, dtype=torch.float32, device=DEV)
    # prefill.py:2148-2199 -> the fa2 branch of paged_run (prefill.py:620-650)
    mod.paged_run(float_ws, int_ws, plan_info, qd, cd[:, 0], cd[:, 1], qo_indptr.to(DEV), indptr.to(DEV),
                  indices.to(DEV), last.to(DEV), out, lse, 1 if causal else 0, 0, -1, False,
                  None, None, None, None, None, None, 0.0, 1.0 / math.sqrt(d), 1.0, 1e-4, 0)
    o_ref, lse_ref = R.batch_prefill_ref(q.float(), qo_indptr, cache.float(), "NHD", indptr, indices, last, causal=causal)
    torch.testing.assert_close(out.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)
    # ragged_run of the same module (csrc/batch_prefill.cu:76-82): k, v dense [nnz, H, D]
    ks = [R.gather_paged_kv(cache, "NHD", indptr, indices, last, r) for r in range(b)]
    k_r = torch.cat([k for k, _ in ks]).to(DEV)
    v_r = torch.cat([v for _, v in ks]).to(DEV)
    kv_indptr_r = torch.tensor([0] + list(torch.tensor(kv_lens).cumsum(0)), dtype=torch.int32)
    plan_info = mod.plan(float_ws, int_ws, pinned, qo_indptr, kv_indptr_r, kv_len_arr, int(qo_indptr[-1]), b, hq, hkv, 1,
                         False, d, d, causal, -1, -1, False)
    out2 = torch.empty_like(qd)
    mod.ragged_run(float_ws, int_ws, plan_info, qd, k_r, v_r, qo_indptr.to(DEV), kv_indptr_r.to(DEV), out2, None,
                   1 if causal else 0, 0, -1, False, None, None, None, None, None, None, 0.0, 1.0 / math.sqrt(d), 1.0,
                   1e-4, 0)
    torch.testing.assert_close(out2.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)


def test_batch_prefill_module_fa3_six_argument_tail():
    """The reference's fa3 16-bit specialisation calls paged_run / ragged_run with SIX additional parameters
    (maybe_prefix_len_ptr, maybe_token_pos_in_items_ptr, maybe_max_item_len_ptr, logits_soft_cap, sm_scale,
    token_pos_in_items_len; flashinfer/prefill.py:624-646) and plans without the fa2-only split arguments."""
    from flashinfer import compat

    hq, hkv, d, ps = 8, 2, 128, 16
    kv_lens, qo_lens = [200, 77], [64, 77]
    b = len(kv_lens)
    cache, indptr, indices, last = make_paged(b, kv_lens, ps, hkv, d, torch.float16, "NHD", seed=23)
    torch.manual_seed(2)
    q = torch.randn(sum(qo_lens), hq, d).half()
    qo_indptr = torch.tensor([0] + list(torch.tensor(qo_lens).cumsum(0)), dtype=torch.int32)
    float_ws, int_ws, pinned = _workspaces()
    mod = compat.get_batch_prefill_module("fa3", torch.float16, torch.float16, torch.float16, torch.int32, d, d, 0,
                                          False, False, False)
    plan_info = mod.plan(float_ws, int_ws, pinned, qo_indptr, indptr, torch.tensor(kv_lens, dtype=torch.int32),
                         int(qo_indptr[-1]), b, hq, hkv, ps, False, d, d, True, -1)
    cd, qd = cache.to(DEV), q.to(DEV)
    out = torch.empty_like(qd)
    mod.paged_run(float_ws, int_ws, plan_info, qd, cd[:, 0], cd[:, 1], qo_indptr.to(DEV), indptr.to(DEV),
                  indices.to(DEV), last.to(DEV), out, None, 1, 0, -1, False,
                  None, None, None, 0.0, 1.0 / math.sqrt(d), 0)
    o_ref, _ = R.batch_prefill_ref(q.float(), qo_indptr, cache.float(), "NHD", indptr, indices, last, causal=True)
    torch.testing.assert_close(out.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
    with pytest.raises(ValueError):
        mod.paged_run(float_ws, int_ws, plan_info, qd, cd[:, 0], cd[:, 1], qo_indptr.to(DEV), indptr.to(DEV),
                      indices.to(DEV), last.to(DEV), out, None, 1, 0, -1, False, None, None, None)


def test_multi_item_scoring_rejects_rows_shorter_than_the_queries_past_the_prefix():
    """the kernel indexes token_pos_in_items[b][q_pos - prefix_len[b]]; a row shorter than
    kv_len - prefix_len would read the next request's row.  plan() refuses it."""
    import flashinfer

    hq, hkv, d, ps, b = 4, 4, 128, 16, 2
    kv_lens, qo_len = [64, 64], 40
    cache, indptr, indices, last = make_paged(b, kv_lens, ps, hkv, d, torch.float16, "NHD", seed=3)
    qo_indptr = (torch.arange(b + 1) * qo_len).to(torch.int32)
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, "NHD")
    kw = dict(causal=True, prefix_len_ptr=torch.tensor([30, 30]).to(torch.uint32).to(DEV))
    with pytest.raises(ValueError, match="token_pos_in_items_len"):  # 34 positions past the prefix, rows of 20
        w.plan(qo_indptr.to(DEV), indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, ps,
               token_pos_in_items_ptr=torch.zeros(b * 20, dtype=torch.uint16, device=DEV), token_pos_in_items_len=20, **kw)
    with pytest.raises(ValueError):  # the last row must be whole too: batch * len entries
        w.plan(qo_indptr.to(DEV), indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, ps,
               token_pos_in_items_ptr=torch.zeros(b * 34 - 3, dtype=torch.uint16, device=DEV), token_pos_in_items_len=34,
               **kw)
    w.plan(qo_indptr.to(DEV), indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, ps,
           token_pos_in_items_ptr=torch.zeros(b * 34, dtype=torch.uint16, device=DEV), token_pos_in_items_len=34, **kw)


def test_batch_prefill_fp8_module_reference_call_sequence():
    """fa3 fp8 specialisation: plan stops at window_left, paged_run takes (scale_q, scale_k, scale_v, sm_scale)
    (csrc/batch_prefill_fp8_sm90.cu:39-44, 81-90)."""
    from flashinfer import compat

    hq, hkv, d, ps = 8, 2, 128, 16
    kv_lens, qo_lens = [300, 130], [200, 130]
    cache, indptr, indices, last = make_paged(2, kv_lens, ps, hkv, d, torch.float8_e4m3fn, "NHD", seed=31)
    torch.manual_seed(2)
    q8 = torch.randn(sum(qo_lens), hq, d).to(torch.float8_e4m3fn)
    sq, sk, sv = torch.rand(hq) + 0.5, torch.rand(hkv) + 0.5, torch.rand(hkv) + 0.5
    qo_indptr = torch.tensor([0] + list(torch.tensor(qo_lens).cumsum(0)), dtype=torch.int32)
    float_ws, int_ws, pinned = _workspaces()
    mod = compat.get_batch_prefill_module("fa3", torch.float8_e4m3fn, torch.float8_e4m3fn, torch.float16, torch.int32,
                                          d, d, 0, False, False, False)
    plan_info = mod.plan(float_ws, int_ws, pinned, qo_indptr, indptr, torch.tensor(kv_lens, dtype=torch.int32),
                         int(qo_indptr[-1]), 2, hq, hkv, ps, False, d, d, True, -1)
    cd, qd = cache.to(DEV), q8.to(DEV)
    out = torch.empty(qd.shape, dtype=torch.float16, device=DEV)
    mod.paged_run(float_ws, int_ws, plan_info, qd, cd[:, 0], cd[:, 1], qo_indptr.to(DEV), indptr.to(DEV),
                  indices.to(DEV), last.to(DEV), out, None, 1, 0, -1, False, sq.to(DEV), sk.to(DEV), sv.to(DEV),
                  1.0 / math.sqrt(d))
    off = 0
    for r in range(2):
        k, v = R.gather_paged_kv(cache, "NHD", indptr, indices, last, r)
        o_ref, _ = R.fp8_attention_ref(q8[off:off + qo_lens[r]], k, v, sq, sk, sv, causal=True)
        torch.testing.assert_close(out[off:off + qo_lens[r]].float().cpu(), o_ref.float(), rtol=5e-2, atol=5e-2)
        off += qo_lens[r]


def test_cascade_page_gemm_modules_reference_call_sequences():
    from flashinfer import compat

    # ---- cascade (flashinfer/cascade.py:87-100, 150-158, 205-216): outputs allocated by the caller ----
    torch.manual_seed(5)
    n, h, d = 37, 6, 128
    va, vb = torch.randn(n, h, d).half(), torch.randn(n, h, d).half()
    sa, sb = torch.randn(n, h) * 2, torch.randn(n, h) * 2
    mod = compat.get_cascade_module()
    vm = torch.empty(n, h, d, dtype=torch.float16, device=DEV)
    sm = torch.empty(n, h, dtype=torch.float32, device=DEV)
    mod.merge_state(va.to(DEV), sa.to(DEV), vb.to(DEV), sb.to(DEV), vm, sm)
    v_ref, s_ref = R.merge_state_ref(va.float(), sa, vb.float(), sb)
    torch.testing.assert_close(vm.float().cpu(), v_ref.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(sm.cpu(), s_ref.float(), rtol=1e-4, atol=1e-4)
    v_in, s_in = va.to(DEV).clone(), sa.to(DEV).clone()
    mask = torch.rand(n) < 0.5
    mod.merge_state_in_place(v_in, s_in, vb.to(DEV), sb.to(DEV), mask.to(DEV))
    want = torch.where(mask[:, None, None], v_ref.float(), va.float())
    torch.testing.assert_close(v_in.float().cpu(), want, rtol=1e-3, atol=1e-3)
    vs = torch.stack([va, vb, va], 1).contiguous()
    ss = torch.stack([sa, sb, sa - 1], 1).contiguous()
    vm3 = torch.empty(n, h, d, dtype=torch.float16, device=DEV)
    sm3 = torch.empty(n, h, dtype=torch.float32, device=DEV)
    mod.merge_states(vs.to(DEV), ss.to(DEV), vm3, sm3)
    v3, s3 = R.merge_states_ref(vs.float(), ss)
    torch.testing.assert_close(vm3.float().cpu(), v3.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(sm3.cpu(), s3.float(), rtol=1e-4, atol=1e-4)

    # ---- page append (flashinfer/page.py:411-424) then decode over the appended cache ----
    hkv, ps = 2, 16
    kv_lens = [40, 7, 100]
    pages = [-(-l // ps) for l in kv_lens]
    kv_indptr = torch.tensor([0] + list(torch.tensor(pages).cumsum(0)), dtype=torch.int32)
    kv_indices = torch.randperm(sum(pages)).to(torch.int32)
    kv_last = torch.tensor([(l - 1) % ps + 1 for l in kv_lens], dtype=torch.int32)
    cache = torch.zeros(sum(pages), 2, ps, hkv, d, dtype=torch.float16, device=DEV)
    nnz = sum(kv_lens)
    k_new, v_new = torch.randn(nnz, hkv, d).half(), torch.randn(nnz, hkv, d).half()
    batch_indices = torch.cat([torch.full((l,), i, dtype=torch.int32) for i, l in enumerate(kv_lens)])
    positions = torch.cat([torch.arange(l, dtype=torch.int32) for l in kv_lens])
    compat.get_page_module().append_paged_kv_cache(
        k_new.to(DEV), v_new.to(DEV), batch_indices.to(DEV), positions.to(DEV), cache[:, 0], cache[:, 1],
        kv_indices.to(DEV), kv_indptr.to(DEV), kv_last.to(DEV), 0)
    torch.cuda.synchronize()
    off = 0
    for r, l in enumerate(kv_lens):
        k, v = R.gather_paged_kv(cache.cpu(), "NHD", kv_indptr, kv_indices, kv_last, r)
        assert torch.equal(k, k_new[off:off + l]) and torch.equal(v, v_new[off:off + l])
        off += l

    # ---- grouped GEMM (flashinfer/gemm.py:2791-2806): workspaces first, out allocated by the caller ----
    groups, m, nn, k = 3, 132, 256, 256
    a = torch.randn(groups * m, k)
    bm = torch.randn(groups, nn, k) / math.sqrt(k)
    a8, a_s = G.quantize_fp8(a, (1, 128), "MN")
    b8, b_s = G.quantize_fp8(bm, (1, 128, 128), "MN")
    m_indptr = (torch.arange(groups + 1) * m).to(torch.int32)
    out = torch.empty(groups * m, nn, dtype=torch.bfloat16, device=DEV)
    gm = compat.get_gemm_sm100_module()
    int_ws = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    float_ws = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    gm.group_gemm_fp8_nt_groupwise(int_ws, float_ws, a8.to(DEV), b8.to(DEV), a_s.to(DEV), b_s.to(DEV), out,
                                   m_indptr.to(DEV), nn, k, 1, 128, 128, "MN", 1)
    ref = G.group_gemm_fp8_nt_groupwise_ref(a8, b8, a_s, b_s, m_indptr, "MN")
    torch.testing.assert_close(out.float().cpu(), ref.float(), rtol=1e-2, atol=1e-2)
    out2 = torch.empty(m, nn, dtype=torch.bfloat16, device=DEV)
    gm.gemm_fp8_nt_groupwise(float_ws, a8[:m].to(DEV), b8[0].to(DEV), a_s[:, :m].contiguous().to(DEV), b_s[0].to(DEV),
                             out2, 1, 128, 128, "MN", 1)
    torch.testing.assert_close(out2.float().cpu(), ref[:m].float(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("kv_len,qo_len,prefix_len,token_pos,tp_len,max_item_len",
                         [(54, 37, 17, list(range(17)) + list(range(19)) + [0], 100, 18),
                          (97, 81, 16, list(range(80)) + [0], 97, 79),
                          (700, 600, 100, [i % 53 for i in range(600)], 640, 52)])
@pytest.mark.parametrize("page_size", [1, 5, 16])
@pytest.mark.parametrize("hq", [4, 32])
@pytest.mark.parametrize("soft_cap", [0.0, 30.0])
def test_multi_item_scoring_matches_custom_mask(kv_len, qo_len, prefix_len, token_pos, tp_len, max_item_len,
                                                page_size, hq, soft_cap):
    """ref: tests/attention/test_batch_prefill_kernels.py:810-1010 -- multi-item scoring equals attention under the
    dense mask: a query at p >= prefix sees the prefix and the keys of its own item after its delimiter
    (token_pos == 0 marks a delimiter, which sees only the prefix); checked against the oracle with that mask."""
    import flashinfer

    hkv, d, b = 4, 128, 2
    kv_lens = [kv_len] * b
    cache, indptr, indices, last = make_paged(b, kv_lens, page_size, hkv, d, torch.float16, "NHD", seed=7)
    torch.manual_seed(8)
    q = torch.randn(b * qo_len, hq, d).half()
    qo_indptr = (torch.arange(b + 1) * qo_len).to(torch.int32)
    tp = torch.zeros(b, tp_len, dtype=torch.int32)
    tp[:, : len(token_pos)] = torch.tensor(token_pos)
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, "NHD")
    w.plan(qo_indptr.to(DEV), indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, page_size, causal=True,
           logits_soft_cap=soft_cap,
           prefix_len_ptr=torch.tensor([prefix_len] * b).to(torch.uint32).to(DEV),
           token_pos_in_items_ptr=tp.reshape(-1).to(torch.uint16).to(DEV), token_pos_in_items_len=tp_len,
           max_item_len_ptr=torch.tensor([max_item_len] * b).to(torch.uint16).to(DEV))
    o, lse = w.run(q.to(DEV), cache.to(DEV), return_lse=True)
    # dense mask of the same rule
    qi = torch.arange(qo_len)[:, None] + (kv_len - qo_len)  # position of each query on the kv axis
    kj = torch.arange(kv_len)[None, :]
    tpos = torch.tensor(token_pos)[:, None]
    in_prefix = qi < prefix_len
    mask = (kj <= qi) & (in_prefix | (kj < prefix_len) | (kj > qi - tpos))
    for r in range(b):
        k, v = R.gather_paged_kv(cache, "NHD", indptr, indices, last, r)
        o_ref, lse_ref = R.attention_ref(q[r * qo_len:(r + 1) * qo_len].float(), k.float(), v.float(), custom_mask=mask,
                                         logits_soft_cap=soft_cap)
        torch.testing.assert_close(o[r * qo_len:(r + 1) * qo_len].float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(lse[r * qo_len:(r + 1) * qo_len].cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)

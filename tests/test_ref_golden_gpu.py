"""GPU parity against vectors the REFERENCE's own pure-torch functions produced (tests/golden/*_ref_golden.npz,
see oracle/make_golden_ref.py): every kernel family consumes a fixture directly, no oracle in between.
Inputs are fp16-exact, so the kernels see precisely the values the reference functions saw.
Tolerances: the reference's bars -- rtol = atol = 1e-3 for 16-bit attention / merge
(tests/attention/test_batch_decode_kernels.py:144-184), 1e-2 for the fp8 groupwise GEMM
(tests/GEMM/test_groupwise_scaled_gemm_fp8.py:71)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def att():
    return np.load(os.path.join(HERE, "attention_ref_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def gem():
    return np.load(os.path.join(HERE, "gemm_ref_golden.npz"), allow_pickle=False)


def t(a):
    # C order: a transposed tensor is stored Fortran-ordered by numpy and would come back non-contiguous
    return torch.from_numpy(np.ascontiguousarray(a))


def paged_from_dense(k, v, page_size, seed):
    """[kv, H, D] rows scattered into shuffled pages of an NHD cache."""
    kv_len, h, d = k.shape
    n_pages = -(-kv_len // page_size)
    g = torch.Generator().manual_seed(seed)
    indices = torch.randperm(n_pages + 2, generator=g)[:n_pages].to(torch.int32)
    cache = torch.zeros(n_pages + 2, 2, page_size, h, d, dtype=k.dtype)
    for p in range(n_pages):
        rows = slice(p * page_size, min((p + 1) * page_size, kv_len))
        n = rows.stop - rows.start
        cache[int(indices[p]), 0, :n] = k[rows]
        cache[int(indices[p]), 1, :n] = v[rows]
    last = (kv_len - 1) % page_size + 1
    return cache, indices, last


@pytest.mark.parametrize("tag", ["gqa_a", "gqa_b", "dec", "gqa_c"])
@pytest.mark.parametrize("causal", [False, True])
def test_single_prefill_against_reference_vectors(att, tag, causal):
    import flashinfer

    q, k, v = (t(att[f"sp_{tag}_{n}"]).to(DEV) for n in "qkv")  # fp16, exact
    want = t(att[f"sp_{tag}_o_{'causal' if causal else 'full'}"])
    o = flashinfer.single_prefill_with_kv_cache(q, k, v, causal=causal)
    torch.testing.assert_close(o.float().cpu(), want, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("tag", ["gqa_a", "gqa_b", "gqa_c"])
@pytest.mark.parametrize("page_size", [1, 16])
def test_batch_prefill_paged_against_reference_vectors(att, tag, page_size):
    """The same vectors through the batch wrapper and a shuffled page table (two copies of the request plus
    a causal / non-causal plan each)."""
    import flashinfer

    q, k, v = (t(att[f"sp_{tag}_{n}"]) for n in "qkv")
    cache, indices, last = paged_from_dense(k, v, page_size, seed=3)
    n_pages = len(indices)
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, "NHD")
    qo_indptr = torch.tensor([0, q.shape[0], 2 * q.shape[0]], dtype=torch.int32, device=DEV)
    kv_indptr = torch.tensor([0, n_pages, 2 * n_pages], dtype=torch.int32, device=DEV)
    for causal in (False, True):
        w.plan(qo_indptr, kv_indptr, torch.cat([indices, indices]).to(DEV),
               torch.tensor([last, last], dtype=torch.int32, device=DEV), q.shape[1], k.shape[1], q.shape[2],
               page_size, causal=causal, q_data_type=torch.float16)
        o = w.run(torch.cat([q, q]).to(DEV), cache.to(DEV))
        want = t(att[f"sp_{tag}_o_{'causal' if causal else 'full'}"])
        torch.testing.assert_close(o[: q.shape[0]].float().cpu(), want, rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(o[q.shape[0]:].float().cpu(), want, rtol=1e-3, atol=1e-3)


def test_decode_against_reference_vectors(att):
    """qo_len = 1 vectors of the reference function: single decode, and batch paged decode (VALU kernel, G = 4)."""
    import flashinfer

    q, k, v = (t(att[f"sp_dec_{n}"]) for n in "qkv")
    want = t(att["sp_dec_o_full"])[0]
    o = flashinfer.single_decode_with_kv_cache(q[0].to(DEV), k.to(DEV), v.to(DEV))
    torch.testing.assert_close(o.float().cpu(), want, rtol=1e-3, atol=1e-3)
    for page_size in (1, 8, 16):
        cache, indices, last = paged_from_dense(k, v, page_size, seed=page_size)
        ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
        w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, "NHD")
        n_pages = len(indices)
        w.plan(torch.tensor([0, n_pages, 2 * n_pages], dtype=torch.int32, device=DEV),
               torch.cat([indices, indices]).to(DEV), torch.tensor([last, last], dtype=torch.int32, device=DEV),
               q.shape[1], k.shape[1], q.shape[2], page_size, q_data_type=torch.float16)
        o = w.run(torch.cat([q, q]).to(DEV), cache.to(DEV))
        torch.testing.assert_close(o[0].float().cpu(), want, rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(o[1].float().cpu(), want, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("causal", [False, True])
def test_batch_prefill_lse_against_reference_vectors(att, tag, causal):
    """o and base-2 lse of the reference's attention_ref (test_blackwell_fmha.py:11-54) for a batch of
    equal-length requests; tag c has qo_len = 1 (the decode wrapper is checked on it too)."""
    import flashinfer

    b, lq, lk = (int(x) for x in att[f"bw_{tag}_meta"])
    q, k, v = (t(att[f"bw_{tag}_{n}"]) for n in "qkv")
    c = "causal" if causal else "full"
    want_o, want_lse = t(att[f"bw_{tag}_o_{c}"]), t(att[f"bw_{tag}_lse_{c}"]).reshape(b * lq, -1)
    h, d = q.shape[1], q.shape[2]
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithRaggedKVCacheWrapper(ws, "NHD")
    w.plan((torch.arange(b + 1, dtype=torch.int32) * lq).to(DEV), (torch.arange(b + 1, dtype=torch.int32) * lk).to(DEV),
           h, h, d, causal=causal, q_data_type=torch.float16)
    o, lse = w.run(q.to(DEV), k.to(DEV), v.to(DEV), return_lse=True)
    torch.testing.assert_close(o.float().cpu(), want_o, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(lse.cpu(), want_lse, rtol=1e-3, atol=1e-3)
    if lq == 1:
        ps = 4
        pages_per = -(-lk // ps)
        cache = torch.zeros(b * pages_per, 2, ps, h, d, dtype=torch.float16)
        for i in range(b):
            kk = torch.zeros(pages_per * ps, h, d, dtype=torch.float16)
            vv = torch.zeros_like(kk)
            kk[:lk], vv[:lk] = k[i * lk:(i + 1) * lk], v[i * lk:(i + 1) * lk]
            cache[i * pages_per:(i + 1) * pages_per, 0] = kk.view(pages_per, ps, h, d)
            cache[i * pages_per:(i + 1) * pages_per, 1] = vv.view(pages_per, ps, h, d)
        wd = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, "NHD")
        wd.plan((torch.arange(b + 1, dtype=torch.int32) * pages_per).to(DEV),
                torch.arange(b * pages_per, dtype=torch.int32, device=DEV),
                torch.full((b,), (lk - 1) % ps + 1, dtype=torch.int32, device=DEV), h, h, d, ps,
                q_data_type=torch.float16)
        o, lse = wd.run(q.to(DEV), cache.to(DEV), return_lse=True)
        torch.testing.assert_close(o.float().cpu(), want_o, rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(lse.cpu(), want_lse, rtol=1e-3, atol=1e-3)


def test_merge_kernels_against_reference_vectors(att):
    """States over keys A and keys B (reference function) merged on the GPU = the reference's state over A|B.
    Covers merge_state, merge_state_in_place, merge_states and the exported ragged form
    fi_variable_length_merge_states with 16-bit partials (the reference's a6 contract: DTypeIn v, f32 s;
    cascade.cuh:686-736), called through the C ABI."""
    import flashinfer
    from flashinfer import _lib

    v_a, s_a, v_b, s_b = (t(att[f"merge_{n}"]) for n in ("o_a", "lse_a", "o_b", "lse_b"))
    want_v, want_s = t(att["merge_o_full"]), t(att["merge_lse_full"])
    for dt, tol in ((torch.float32, 1e-5), (torch.float16, 1e-3), (torch.bfloat16, 8e-3)):
        va, vb = v_a.to(dt).to(DEV), v_b.to(dt).to(DEV)
        sa, sb = s_a.to(DEV), s_b.to(DEV)
        v, s = flashinfer.merge_state(va, sa, vb, sb)
        torch.testing.assert_close(v.float().cpu(), want_v, rtol=tol, atol=tol)
        torch.testing.assert_close(s.cpu(), want_s, rtol=1e-4, atol=1e-4)
        v2, s2 = va.clone(), sa.clone()
        flashinfer.merge_state_in_place(v2, s2, vb, sb)
        torch.testing.assert_close(v2.float().cpu(), want_v, rtol=tol, atol=tol)
        v3, s3 = flashinfer.merge_states(torch.stack([va, vb], 1), torch.stack([sa, sb], 1))
        torch.testing.assert_close(v3.float().cpu(), want_v, rtol=tol, atol=tol)
        torch.testing.assert_close(s3.cpu(), want_s, rtol=1e-4, atol=1e-4)
        if dt == torch.float32:
            continue
        # ragged form: row r owns entries indptr[r]..indptr[r+1]; mix of 2-, 1- and 0-entry rows
        n, h, d = va.shape
        vi = torch.stack([va, vb], 1).reshape(2 * n, h, d).contiguous()
        si = torch.stack([sa, sb], 1).reshape(2 * n, h).contiguous()
        indptr = torch.cat([torch.arange(n + 1) * 2, torch.tensor([2 * n, 2 * n])]).to(torch.int32).to(DEV)
        rows = n + 2  # last two rows: empty
        vo = torch.full((rows, h, d), 7.0, dtype=dt, device=DEV)
        so = torch.full((rows, h), 7.0, dtype=torch.float32, device=DEV)
        _lib.check(_lib.lib().fi_variable_length_merge_states(
            vi.data_ptr(), si.data_ptr(), indptr.data_ptr(), vo.data_ptr(), so.data_ptr(), rows, h, d,
            _lib.fi_dtype(dt), _lib.fi_dtype(dt), _lib.current_stream(vi.device)), "variable_length_merge_states")
        torch.cuda.synchronize()
        torch.testing.assert_close(vo[:n].float().cpu(), want_v, rtol=tol, atol=tol)
        torch.testing.assert_close(so[:n].cpu(), want_s, rtol=1e-4, atol=1e-4)
        assert torch.all(vo[n:] == 0) and torch.all(so[n:] == _lib.FI_NEG_INF)  # cascade.cuh:397-405


def test_variable_length_merge_states_matches_oracle_ragged():
    """fi_variable_length_merge_states on ragged random states (1..9 partials per row, fp16 / bf16 partials)
    against oracle.variable_length_merge_states_ref."""
    from flashinfer import _lib
    from oracle import attention_ref as R

    g = torch.Generator().manual_seed(5)
    counts = [1, 9, 2, 0, 5, 3, 1, 7]
    indptr = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32)
    nnz, h, d = int(indptr[-1]), 6, 128
    for dt, tol in ((torch.float16, 1e-3), (torch.bfloat16, 8e-3)):
        v = torch.randn(nnz, h, d, generator=g).to(dt)
        s = torch.randn(nnz, h, generator=g) * 3
        # row 3 has no entry: the oracle's stack needs at least one, so compare it separately
        keep = [r for r, c in enumerate(counts) if c > 0]
        vo = torch.empty(len(counts), h, d, dtype=dt, device=DEV)
        so = torch.empty(len(counts), h, dtype=torch.float32, device=DEV)
        v_d, s_d, ip_d = v.to(DEV), s.to(DEV), indptr.to(DEV)  # kept alive across the asynchronous launch
        _lib.check(_lib.lib().fi_variable_length_merge_states(
            v_d.data_ptr(), s_d.data_ptr(), ip_d.data_ptr(), vo.data_ptr(), so.data_ptr(),
            len(counts), h, d, _lib.fi_dtype(dt), _lib.fi_dtype(dt), _lib.current_stream(vo.device)), "vlms")
        torch.cuda.synchronize()
        for r in keep:
            lo, hi = int(indptr[r]), int(indptr[r + 1])
            v_ref, s_ref = R.merge_states_ref(v[lo:hi].float()[None], s[lo:hi][None])
            torch.testing.assert_close(vo[r].float().cpu(), v_ref[0].float(), rtol=tol, atol=tol)
            torch.testing.assert_close(so[r].cpu(), s_ref[0].float(), rtol=1e-4, atol=1e-4)
        assert torch.all(vo[3] == 0) and torch.all(so[3] == _lib.FI_NEG_INF)


@pytest.mark.parametrize("mode", ["MN", "K"])
def test_gemm_against_reference_vectors(gem, mode):
    """fp8 operands and scales as the reference's quantize_fp8 made them; expected = the reference's
    dequantise -> einsum.  Plain, grouped (uniform) and grouped (ragged, with an empty group)."""
    import flashinfer

    def f8(name):
        return t(gem[name]).view(torch.float8_e4m3fn).to(DEV)

    c = flashinfer.gemm.gemm_fp8_nt_groupwise(f8(f"g2_{mode}_a8"), f8(f"g2_{mode}_b8"), t(gem[f"g2_{mode}_a_s"]).to(DEV),
                                              t(gem[f"g2_{mode}_b_s"]).to(DEV), scale_major_mode=mode,
                                              out_dtype=torch.bfloat16)
    torch.testing.assert_close(c.float().cpu(), t(gem[f"g2_{mode}_c"]), rtol=1e-2, atol=1e-2)
    a8, b8 = f8(f"g3_{mode}_a8"), f8(f"g3_{mode}_b8")
    a_s, b_s = t(gem[f"g3_{mode}_a_s"]).to(DEV), t(gem[f"g3_{mode}_b_s"]).to(DEV)
    groups, mg = b8.shape[0], a8.shape[0] // b8.shape[0]
    for indptr, want in (((torch.arange(groups + 1) * mg).to(torch.int32), f"g3_{mode}_c"),
                         (t(gem["g3_m_indptr_ragged"]), f"g3_{mode}_c_ragged")):
        out = flashinfer.gemm.group_gemm_fp8_nt_groupwise(a8, b8, a_s, b_s, indptr.to(DEV), scale_major_mode=mode,
                                                          out_dtype=torch.bfloat16)
        torch.testing.assert_close(out.float().cpu(), t(gem[want]), rtol=1e-2, atol=1e-2)
    # fp16 output: tighter (10-bit significand)
    out = flashinfer.gemm.group_gemm_fp8_nt_groupwise(a8, b8, a_s, b_s, (torch.arange(groups + 1) * mg).to(torch.int32).to(DEV),
                                                      scale_major_mode=mode, out_dtype=torch.float16)
    torch.testing.assert_close(out.float().cpu(), t(gem[f"g3_{mode}_c"]), rtol=2e-3, atol=2e-3)

"""GPU parity: HIP batch / single decode (through the C ABI) against the CPU oracle on the same seeded
inputs.  Grid modelled on the reference's tests/attention/test_batch_decode_kernels.py:58-70,
test_decode_prefill_lse.py:22-71 and test_non_contiguous_*.py.  Tolerance: rtol = atol = 1e-3 for
16-bit kv (the reference's own bar, test_batch_decode_kernels.py:144-184)."""
import math

import pytest
import torch

from oracle import attention_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def tol(dtype):
    """rtol = atol = 1e-3 (the reference's bar) for fp16 outputs.  A bf16 output cannot carry 1e-3: its
    own rounding is half an ulp = 2^-9 relative, so that is added on top for bf16."""
    if dtype == torch.bfloat16:
        return dict(rtol=1e-3 + 2.0 ** -8, atol=2e-3)
    return dict(rtol=1e-3, atol=1e-3)


def rope_rt(q_dtype, head_dim):
    """rope_round_dtype for the oracle: decode with fused RoPE runs on the matrix-core kernel for head_dim 64 / 128,
    which rounds the rotated q / k to the 16-bit type (as the reference's tensor-core decode = its prefill kernel
    does, prefill.cuh:465-612); other head dims stay on the VALU kernel, which rotates in f32 (decode.cuh:445-466)."""
    return q_dtype if head_dim in (64, 128) else None


def make_paged(batch, kv_lens, page_size, hkv, d, dtype, layout, seed, shuffle=True, extra_pages=3):
    g = torch.Generator().manual_seed(seed)
    pages = [max(0, -(-l // page_size)) for l in kv_lens]
    total = sum(pages)
    indptr = torch.tensor([0] + list(torch.tensor(pages).cumsum(0)), dtype=torch.int32)
    last = torch.tensor([(l - 1) % page_size + 1 if l > 0 else 0 for l in kv_lens], dtype=torch.int32)
    perm = torch.randperm(total + extra_pages, generator=g)[:total] if shuffle else torch.arange(total)
    indices = perm.to(torch.int32)
    shape = (total + extra_pages, 2, page_size, hkv, d) if layout == "NHD" else (total + extra_pages, 2, hkv, page_size, d)
    cache = torch.randn(shape, generator=g)
    if dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
        cache = cache.to(dtype)
    else:
        cache = cache.to(dtype)
    return cache, indptr, indices, last


def run_batch_decode(q, cache, layout, indptr, indices, last, hq, hkv, d, page_size, return_lse=True, **plan_kw):
    import flashinfer

    ws = torch.zeros(64 * 1024 * 1024, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, layout)
    w.plan(indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, page_size,
           q_data_type=q.dtype, kv_data_type=cache.dtype, **plan_kw)
    return w.run(q.to(DEV), cache.to(DEV), return_lse=return_lse), w


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("layout", ["NHD", "HND"])
@pytest.mark.parametrize("page_size", [1, 8, 16])
@pytest.mark.parametrize("hq,hkv", [(4, 4), (32, 4), (8, 4), (12, 4), (28, 4)])
def test_batch_decode_matches_oracle(dtype, layout, page_size, hq, hkv):
    d = 128
    kv_lens = [54, 97, 512, 1, 2048, 33, 16, 17]
    torch.manual_seed(7)
    cache, indptr, indices, last = make_paged(len(kv_lens), kv_lens, page_size, hkv, d, dtype, layout, seed=11)
    q = torch.randn(len(kv_lens), hq, d).to(dtype)
    (o, lse), _ = run_batch_decode(q, cache, layout, indptr, indices, last, hq, hkv, d, page_size)
    o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), layout, indptr, indices, last)
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), **tol(dtype))
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("page_size", [5, 16])
@pytest.mark.parametrize("hq,hkv", [(64, 8), (32, 2), (64, 2), (40, 8), (26, 2), (64, 1), (71, 1)])
def test_batch_decode_wide_groups_matrix_core_path(dtype, d, page_size, hq, hkv):
    """Groups of >= 5 query heads per kv head (32 per wave) run on the MFMA decode kernel (decode_mfma_kernel.h); the
    reference's counterpart is use_tensor_cores=True (tests/attention/test_tensor_cores_decode.py)."""
    kv_lens = [1, 31, 32, 33, 777, 4096, 64, 2500, 95]
    torch.manual_seed(17)
    layout = "NHD" if page_size == 16 else "HND"
    cache, indptr, indices, last = make_paged(len(kv_lens), kv_lens, page_size, hkv, d, dtype, layout, seed=23)
    q = torch.randn(len(kv_lens), hq, d).to(dtype)
    (o, lse), _ = run_batch_decode(q, cache, layout, indptr, indices, last, hq, hkv, d, page_size)
    o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), layout, indptr, indices, last)
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), **tol(dtype))
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("kv_dtype", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("qdtype", [torch.float16, torch.bfloat16])
def test_batch_decode_wide_groups_fp8_cache(kv_dtype, d, qdtype):
    hq, hkv, page_size = 32, 4, 16
    kv_lens = [1, 33, 500, 3000, 64, 129]
    cache, indptr, indices, last = make_paged(len(kv_lens), kv_lens, page_size, hkv, d, kv_dtype, "NHD", seed=41)
    torch.manual_seed(43)
    q = torch.randn(len(kv_lens), hq, d).to(qdtype)
    (o, lse), _ = run_batch_decode(q, cache, "NHD", indptr, indices, last, hq, hkv, d, page_size)
    o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), "NHD", indptr, indices, last)
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), **tol(qdtype))
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)


def test_batch_decode_wide_group_no_split_and_graph_padding():
    # enough requests that the planner does not split (one wave per request x kv head), then the padded
    # CUDA-graph style plan of the same shape
    hq, hkv, d, page_size = 64, 8, 128, 16
    kv_lens = [((37 * i) % 300) + 1 for i in range(320)]
    cache, indptr, indices, last = make_paged(len(kv_lens), kv_lens, page_size, hkv, d, torch.float16, "NHD", seed=29)
    torch.manual_seed(31)
    q = torch.randn(len(kv_lens), hq, d).half()
    o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), "NHD", indptr, indices, last)
    (o, lse), w = run_batch_decode(q, cache, "NHD", indptr, indices, last, hq, hkv, d, page_size)
    assert w._plan_info[9] == 0  # split_kv off
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("d", [64, 128, 256, 512])
@pytest.mark.parametrize("kv_dtype", [torch.float16, torch.float8_e4m3fn, torch.float8_e5m2])
def test_batch_decode_head_dims_and_fp8_kv(d, kv_dtype):
    hq, hkv, page_size = 8, 2, 16
    kv_lens = [200, 3, 1000, 77]
    torch.manual_seed(3)
    cache, indptr, indices, last = make_paged(len(kv_lens), kv_lens, page_size, hkv, d, kv_dtype, "NHD", seed=5)
    q = torch.randn(len(kv_lens), hq, d).half()
    (o, lse), _ = run_batch_decode(q, cache, "NHD", indptr, indices, last, hq, hkv, d, page_size)
    # the oracle sees the same (already quantised) cache values, so the 16-bit tolerance applies
    o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), "NHD", indptr, indices, last)
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("mode", ["ROPE_LLAMA", "ALIBI"])
@pytest.mark.parametrize("d", [64, 128])
def test_batch_decode_pos_encoding(mode, d):
    hq, hkv, page_size = 8, 2, 8
    kv_lens = [54, 700, 1, 2049]
    torch.manual_seed(5)
    cache, indptr, indices, last = make_paged(len(kv_lens), kv_lens, page_size, hkv, d, torch.float16, "NHD", seed=9)
    q = torch.randn(len(kv_lens), hq, d).half()
    (o, lse), _ = run_batch_decode(q, cache, "NHD", indptr, indices, last, hq, hkv, d, page_size,
                                   pos_encoding_mode=mode, rope_theta=1e4, rope_scale=1.0)
    o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), "NHD", indptr, indices, last,
                                        pos_encoding_mode=mode,
                                        rope_round_dtype=rope_rt(torch.float16, d) if mode == "ROPE_LLAMA" else None)
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_batch_decode_fused_rope_against_f32_rotation_oracle(dtype, d):
    """the matrix-core decode kernels round the rotated q / k to the 16-bit type (what the reference's
    tensor-core decode = its prefill kernel does, prefill.cuh:465-612); the reference's DEFAULT decode path rotates
    and multiplies in f32 (decode.cuh:445-466).  This test measures the distance to THAT arithmetic (oracle without
    rope_round_dtype) and holds it to a stated, looser bar: the rounding of q and k perturbs every logit by about
    2^-11 (fp16) / 2^-8 (bf16) of |q||k|/sqrt(d), which moves the output by up to ~2e-3 (fp16) / ~1.5e-2 (bf16) of the
    value scale (measured r3: 4.0e-4 fp16, 2.8e-3 bf16) -- recorded in INTEGRATION.md as a known difference from the reference's non-tensor-core decode."""
    hq, hkv, page_size = 8, 2, 16
    kv_lens = [54, 700, 1, 2049]
    torch.manual_seed(15)
    cache, indptr, indices, last = make_paged(len(kv_lens), kv_lens, page_size, hkv, d, dtype, "NHD", seed=19)
    q = torch.randn(len(kv_lens), hq, d).to(dtype)
    (o, lse), _ = run_batch_decode(q, cache, "NHD", indptr, indices, last, hq, hkv, d, page_size,
                                   pos_encoding_mode="ROPE_LLAMA", rope_theta=1e4, rope_scale=1.0)
    o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), "NHD", indptr, indices, last,
                                        pos_encoding_mode="ROPE_LLAMA")  # f32 rotation, no rounding of q / k
    err = (o.float().cpu() - o_ref.float()).abs().max().item()
    lerr = (lse.cpu() - lse_ref.float()).abs().max().item()
    print(f"decode fused RoPE vs f32-rotation oracle, {dtype} d={d}: max |o - ref| {err:.2e}, max |lse - ref| {lerr:.2e}")
    bar = 1e-3 if dtype == torch.float16 else 6e-3  # measured r3: 4.0e-4 / 2.8e-3
    assert err < bar and lerr < bar


@pytest.mark.parametrize("window_left,soft_cap", [(15, 0.0), (300, 0.0), (-1, 30.0), (64, 8.0)])
def test_batch_decode_window_and_soft_cap(window_left, soft_cap):
    hq, hkv, d, page_size = 8, 4, 128, 16
    kv_lens = [54, 700, 5, 1500]
    torch.manual_seed(6)
    cache, indptr, indices, last = make_paged(len(kv_lens), kv_lens, page_size, hkv, d, torch.bfloat16, "NHD", seed=2)
    q = (torch.randn(len(kv_lens), hq, d) * 3).bfloat16()
    (o, lse), _ = run_batch_decode(q, cache, "NHD", indptr, indices, last, hq, hkv, d, page_size,
                                   window_left=window_left, logits_soft_cap=soft_cap)
    o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), "NHD", indptr, indices, last,
                                        window_left=window_left, logits_soft_cap=soft_cap)
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), **tol(torch.bfloat16))
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=2e-3)


def test_empty_request_edge_case():
    # ref: tests/attention/test_decode_prefill_lse.py:24-26
    hq, hkv, d, page_size = 4, 2, 128, 16
    cache = torch.randn(9, 2, page_size, hkv, d).half()
    indptr = torch.tensor([0, 0, 9], dtype=torch.int32)
    indices = torch.arange(9, dtype=torch.int32)
    last = torch.tensor([0, 1], dtype=torch.int32)
    q = torch.randn(2, hq, d).half()
    (o, lse), _ = run_batch_decode(q, cache, "NHD", indptr, indices, last, hq, hkv, d, page_size)
    o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), "NHD", indptr, indices, last)
    assert torch.all(o[0] == 0) and torch.all(lse[0].cpu() == R.NEG_INF_SENTINEL)
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)


def test_tuple_cache_noncontiguous_q_and_out_args():
    hq, hkv, d, page_size = 8, 2, 128, 16
    kv_lens = [100, 260]
    cache, indptr, indices, last = make_paged(2, kv_lens, page_size, hkv, d, torch.float16, "NHD", seed=4)
    # non-contiguous q: a slice of a packed qkv tensor (ref: tests/attention/test_non_contiguous_decode.py)
    qkv = torch.randn(2, hq + 2 * hkv, d).half().to(DEV)
    q = qkv[:, :hq]
    import flashinfer

    ws = torch.zeros(32 * 1024 * 1024, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, "NHD")
    w.plan(indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, page_size, q_data_type=torch.float16, kv_data_type=torch.float16)
    cd = cache.to(DEV)
    out = torch.empty(2, hq, d, dtype=torch.float16, device=DEV)
    lse = torch.empty(2, hq, dtype=torch.float32, device=DEV)
    o2, l2 = w.run(q, (cd[:, 0], cd[:, 1]), out=out, lse=lse, return_lse=True)
    assert o2.data_ptr() == out.data_ptr()
    o_ref, lse_ref = R.batch_decode_ref(q.float().cpu(), cache.float(), "NHD", indptr, indices, last)
    torch.testing.assert_close(out.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)
    # dtype mismatch against the plan raises ValueError (ref: flashinfer/utils.py:239-251)
    with pytest.raises(ValueError):
        w.run(q.bfloat16(), cd)
    # int64 page table is rejected (ref: flashinfer/decode.py:900-908)
    with pytest.raises(ValueError):
        w.plan(indptr.long().to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, page_size)


def test_cuda_graph_wrapper_fixed_shape_and_replan():
    import flashinfer

    hq, hkv, d, page_size, b = 8, 2, 128, 16, 6
    max_pages = 400
    ws = torch.zeros(64 * 1024 * 1024, dtype=torch.uint8, device=DEV)
    w = flashinfer.CUDAGraphBatchDecodeWithPagedKVCacheWrapper(
        ws, torch.empty(b + 1, dtype=torch.int32, device=DEV),
        torch.empty(max_pages, dtype=torch.int32, device=DEV),
        torch.empty(b, dtype=torch.int32, device=DEV), "NHD")
    q = torch.randn(b, hq, d).half()
    for seed, kv_lens in [(1, [40, 900, 17, 1, 333, 64]), (2, [700, 5, 1200, 90, 16, 2])]:
        cache, indptr, indices, last = make_paged(b, kv_lens, page_size, hkv, d, torch.float16, "NHD", seed=seed)
        w.plan(indptr, indices, last, hq, hkv, d, page_size, q_data_type=torch.float16, kv_data_type=torch.float16)
        o, lse = w.run(q.to(DEV), cache.to(DEV), return_lse=True)
        o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), "NHD", indptr, indices, last)
        torch.testing.assert_close(o.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)
    with pytest.raises(ValueError):
        w.plan(indptr[:-1], indices, last[:-1], hq, hkv, d, page_size)


@pytest.mark.parametrize("kv_len", [1, 54, 97, 512, 2048, 5000])
@pytest.mark.parametrize("layout", ["NHD", "HND"])
@pytest.mark.parametrize("hq,hkv", [(32, 32), (32, 8), (32, 4), (48, 2)])
def test_single_decode_matches_oracle(kv_len, layout, hq, hkv):
    import flashinfer

    d = 128
    torch.manual_seed(0)
    q = torch.randn(hq, d).half()
    shape = (kv_len, hkv, d) if layout == "NHD" else (hkv, kv_len, d)
    k, v = torch.randn(shape).half(), torch.randn(shape).half()
    o, lse = flashinfer.single_decode_with_kv_cache(q.to(DEV), k.to(DEV), v.to(DEV), kv_layout=layout, return_lse=True)
    o_ref, lse_ref = R.single_decode_ref(q.float(), k.float(), v.float(), layout)
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)


def test_single_decode_c1_config_and_rope():
    # BASELINE config C1: fp16, 32/32 heads, d=128, kv_len=2048, seed 0
    import flashinfer

    torch.manual_seed(0)
    q = torch.randn(32, 128).half()
    k, v = torch.randn(2048, 32, 128).half(), torch.randn(2048, 32, 128).half()
    o = flashinfer.single_decode_with_kv_cache(q.to(DEV), k.to(DEV), v.to(DEV))
    o_ref, _ = R.single_decode_ref(q.float(), k.float(), v.float())
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
    o = flashinfer.single_decode_with_kv_cache(q.to(DEV), k.to(DEV), v.to(DEV), pos_encoding_mode="ROPE_LLAMA")
    o_ref, _ = R.single_decode_ref(q.float(), k.float(), v.float(), pos_encoding_mode="ROPE_LLAMA",
                                   rope_round_dtype=rope_rt(torch.float16, 128))
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)


def test_full_size_c2_split_invariance():
    """BASELINE config C2 at full size (bs=64, kv=8192, 32/8 heads, page 16): too big for the oracle, so
    check a size-independent property -- the split-KV result equals the unsplit one (merge associativity)
    -- plus the oracle on a sample of requests."""
    import flashinfer

    b, hq, hkv, d, ps, L = 64, 32, 8, 128, 16, 8192
    g = torch.Generator(device=DEV).manual_seed(0)
    npages = b * L // ps
    cache = torch.randn(npages, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16, generator=g)
    q = torch.randn(b, hq, d, device=DEV, dtype=torch.bfloat16, generator=g)
    indptr = (torch.arange(b + 1, dtype=torch.int32) * (L // ps)).to(DEV)
    indices = torch.randperm(npages, device=DEV, generator=g).to(torch.int32)
    last = torch.full((b,), ps, dtype=torch.int32, device=DEV)
    ws = torch.zeros(128 * 1024 * 1024, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, "NHD")
    w.plan(indptr, indices, last, hq, hkv, d, ps, q_data_type=torch.bfloat16, kv_data_type=torch.bfloat16)
    o_split, lse_split = w.run(q, cache, return_lse=True)
    assert w._plan_info[9] == 1  # split-kv on a 256-CU chip
    w.plan(indptr, indices, last, hq, hkv, d, ps, q_data_type=torch.bfloat16, kv_data_type=torch.bfloat16, disable_split_kv=True)
    assert w._plan_info[9] == 0
    o_one, lse_one = w.run(q, cache, return_lse=True)
    torch.testing.assert_close(o_split.float(), o_one.float(), **tol(torch.bfloat16))
    torch.testing.assert_close(lse_split, lse_one, rtol=1e-4, atol=1e-4)
    for r in (0, 37, 63):
        pages = indices[indptr[r]:indptr[r + 1]].long()
        sub = cache[pages].float().cpu()  # this request's pages, in order
        o_ref, lse_ref = R.batch_decode_ref(
            q[r:r + 1].float().cpu(), sub, "NHD", torch.tensor([0, len(pages)], dtype=torch.int32),
            torch.arange(len(pages), dtype=torch.int32), last[r:r + 1].cpu())
        torch.testing.assert_close(o_split[r].float().cpu(), o_ref[0].float(), **tol(torch.bfloat16))
        torch.testing.assert_close(lse_split[r].cpu(), lse_ref[0].float(), rtol=1e-3, atol=1e-3)


def test_run_is_graph_capturable():
    """run() issues only kernel launches on the current stream, so it can be captured into a hipGraph and
    replayed (ref: CUDA-graph mode, tests/attention/test_batch_decode_kernels.py:503)."""
    import flashinfer

    hq, hkv, d, ps, b = 8, 2, 128, 16, 4
    kv_lens = [300, 40, 1000, 77]
    cache, indptr, indices, last = make_paged(b, kv_lens, ps, hkv, d, torch.float16, "NHD", seed=12)
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.CUDAGraphBatchDecodeWithPagedKVCacheWrapper(
        ws, torch.empty(b + 1, dtype=torch.int32, device=DEV), torch.empty(512, dtype=torch.int32, device=DEV),
        torch.empty(b, dtype=torch.int32, device=DEV), "NHD")
    w.plan(indptr, indices, last, hq, hkv, d, ps, q_data_type=torch.float16)
    q = torch.randn(b, hq, d).half().to(DEV)
    cd = cache.to(DEV)
    out = torch.empty_like(q)
    w.run(q, cd, out=out)  # warm-up outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        w.run(q, cd, out=out)
    # new query values in the same buffers, then replay
    q.copy_(torch.randn(b, hq, d).half())
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    o_ref, _ = R.batch_decode_ref(q.float().cpu(), cache.float(), "NHD", indptr, indices, last)
    torch.testing.assert_close(out.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)


def test_fast_decode_plan_matches_plan():
    """ref: flashinfer/decode.py:2416-2579 -- same plan, no buffer copies, host indptr supplied by the caller."""
    import functools

    import flashinfer

    hq, hkv, d, page_size = 32, 8, 128, 16
    kv_lens = [100, 2000, 17, 512]
    cache, indptr, indices, last = make_paged(len(kv_lens), kv_lens, page_size, hkv, d, torch.float16, "NHD", seed=8)
    torch.manual_seed(9)
    q = torch.randn(len(kv_lens), hq, d).half().to(DEV)
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, "NHD")
    w.plan(indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, page_size, q_data_type=torch.float16)
    o_ref = w.run(q, cache.to(DEV))
    # graph-mode wrapper whose buffers the caller fills in place
    bufs = (torch.zeros(len(kv_lens) + 1, dtype=torch.int32, device=DEV),
            torch.zeros(len(indices) + 8, dtype=torch.int32, device=DEV),
            torch.zeros(len(kv_lens), dtype=torch.int32, device=DEV))
    wg = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, "NHD", use_cuda_graph=True, paged_kv_indptr_buffer=bufs[0],
                                                       paged_kv_indices_buffer=bufs[1], paged_kv_last_page_len_buffer=bufs[2])
    wg.begin_forward = functools.partial(flashinfer.fast_decode_plan, wg)
    bufs[0].copy_(indptr)
    bufs[1][: len(indices)].copy_(indices)
    bufs[2].copy_(last)
    wg.begin_forward(bufs[0], bufs[1][: len(indices)], bufs[2], hq, hkv, d, page_size, q_data_type=torch.float16,
                     global_override_indptr_cpu=indptr)
    o = wg.run(q, cache.to(DEV))
    assert torch.equal(o, o_ref) or torch.allclose(o.float(), o_ref.float(), rtol=1e-3, atol=1e-3)
    o_ora, _ = R.batch_decode_ref(q.float().cpu(), cache.float(), "NHD", indptr, indices, last)
    torch.testing.assert_close(o.float().cpu(), o_ora.float(), rtol=1e-3, atol=1e-3)


def test_single_decode_head_dim_512():
    import flashinfer

    torch.manual_seed(2)
    q = torch.randn(8, 512).half()
    k, v = torch.randn(700, 2, 512).half(), torch.randn(700, 2, 512).half()
    o, lse = flashinfer.single_decode_with_kv_cache(q.to(DEV), k.to(DEV), v.to(DEV), return_lse=True)
    o_ref, lse_ref = R.single_decode_ref(q.float(), k.float(), v.float())
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)


def test_decode_suite_through_r1_kernel_choice():
    """FI_DECODE_MFMA16=0: groups of <= 4 heads back on the VALU kernel, wider ones on the 32x32x16 matrix-core
    kernel (the default sends every group of <= 16 heads to decode_mfma16_kernel.h).  The switch is read once per
    process, so the suite runs in a child."""
    import os
    import subprocess
    import sys

    if os.environ.get("FI_DECODE_MFMA16") == "0":
        pytest.skip("already the child")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FI_DECODE_MFMA16="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_decode_gpu.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout

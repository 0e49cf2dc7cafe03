"""Seeded random sweep of the batch decode / prefill wrappers against the CPU oracle: shapes, group sizes,
page sizes, layouts, variants and plan modes drawn at random (fixed seeds, so failures reproduce).  The grid
tests elsewhere pin named cases; this one looks for interactions between features (ragged last pages, split
plans, masks, GQA packing, dtype mixes) the grids do not enumerate."""
import os
import random

import pytest
import torch

from oracle import attention_ref as R
from test_decode_gpu import make_paged, tol
from test_prefill_gpu import ptol
from test_decode_gpu import rope_rt

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SEEDS = range(int(os.environ.get("FI_FUZZ_SEEDS", "24")))  # FI_FUZZ_SEEDS=300 for a longer hunt


def _lens(rng, n, hi):
    return [rng.choice([1, 2, rng.randint(1, 70), rng.randint(1, hi)]) for _ in range(n)]


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_batch_decode(seed):
    import flashinfer

    rng = random.Random(1000 + seed)
    d = rng.choice([64, 128, 128, 256])
    hkv = rng.choice([1, 2, 4, 8])
    group = rng.choice([1, 2, 3, 4, 5, 7, 8, 16])
    hq = hkv * group
    ps = rng.choice([1, 2, 4, 5, 8, 16, 32])
    layout = rng.choice(["NHD", "HND"])
    qdt = rng.choice([torch.float16, torch.bfloat16])
    kvdt = rng.choice([qdt, qdt, torch.float8_e4m3fn, torch.float8_e5m2])
    batch = rng.randint(1, 12)
    kv_lens = _lens(rng, batch, 3000)
    if rng.random() < 0.2:
        kv_lens[rng.randrange(batch)] = 0
    variant = rng.choice(["plain", "plain", "rope", "alibi", "window", "softcap"])
    kw, okw = {}, {}
    if variant == "rope":
        kw, okw = dict(pos_encoding_mode="ROPE_LLAMA"), dict(pos_encoding_mode="ROPE_LLAMA", rope_round_dtype=rope_rt(qdt, d))
    elif variant == "alibi":
        kw, okw = dict(pos_encoding_mode="ALIBI"), dict(pos_encoding_mode="ALIBI")
    elif variant == "window":
        wl = rng.randint(1, 400)
        kw, okw = dict(window_left=wl), dict(window_left=wl)
    elif variant == "softcap":
        kw, okw = dict(logits_soft_cap=20.0), dict(logits_soft_cap=20.0)
    graph = rng.random() < 0.3
    cache, indptr, indices, last = make_paged(batch, kv_lens, ps, hkv, d, kvdt, layout, seed=seed)
    torch.manual_seed(seed)
    q = torch.randn(batch, hq, d).to(qdt)
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
    if graph:
        w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(
            ws, layout, use_cuda_graph=True,
            paged_kv_indptr_buffer=torch.zeros(batch + 1, dtype=torch.int32, device=DEV),
            paged_kv_indices_buffer=torch.zeros(len(indices) + 4, dtype=torch.int32, device=DEV),
            paged_kv_last_page_len_buffer=torch.zeros(batch, dtype=torch.int32, device=DEV))
    else:
        w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, layout)
    w.plan(indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, ps, q_data_type=qdt, kv_data_type=kvdt, **kw)
    o, lse = w.run(q.to(DEV), cache.to(DEV), return_lse=True)
    o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), layout, indptr, indices, last, **okw)
    t = tol(qdt)
    if variant == "rope":  # rotation re-rounds through f32 sin/cos of large positions
        t = dict(rtol=t["rtol"] * 2, atol=t["atol"] * 2)
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), **t)
    # fused RoPE re-rounds the rotated q / k to the 16-bit type (as the reference does in shared memory,
    # prefill.cuh:465-612): with one or two visible keys the lse IS a single logit and carries that rounding
    lt = (2e-2 if qdt == torch.bfloat16 else 4e-3) if variant == "rope" else 2e-3
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=lt, atol=lt)


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_batch_prefill(seed):
    import flashinfer

    rng = random.Random(2000 + seed)
    d = rng.choice([64, 128, 128, 256])
    hkv = rng.choice([1, 2, 4])
    group = rng.choice([1, 2, 4, 7, 8])
    hq = hkv * group
    ps = rng.choice([1, 3, 4, 8, 16])
    layout = rng.choice(["NHD", "HND"])
    qdt = rng.choice([torch.float16, torch.bfloat16])
    kvdt = rng.choice([qdt, qdt, torch.float8_e4m3fn]) if d != 256 else qdt
    batch = rng.randint(1, 5)
    kv_lens = _lens(rng, batch, 2500)
    qo_lens = [rng.choice([0, 1, rng.randint(1, 40), rng.randint(1, max(1, k))]) if rng.random() < 0.8
               else rng.randint(1, k + 50) for k in kv_lens]
    if sum(qo_lens) == 0:
        qo_lens[0] = 1
    causal = rng.random() < 0.6
    variant = rng.choice(["plain", "plain", "rope", "alibi", "window", "softcap", "mask"])
    kw, okw = {}, {}
    if variant == "rope":
        kw, okw = dict(pos_encoding_mode="ROPE_LLAMA"), dict(pos_encoding_mode="ROPE_LLAMA", rope_round_dtype=qdt)
    elif variant == "alibi":
        kw, okw = dict(pos_encoding_mode="ALIBI"), dict(pos_encoding_mode="ALIBI")
    elif variant == "window":
        wl = rng.randint(1, 300)
        kw, okw = dict(window_left=wl), dict(window_left=wl)
    elif variant == "softcap":
        kw, okw = dict(logits_soft_cap=15.0), dict(logits_soft_cap=15.0)
    plan_mode = rng.choice(["auto", "auto", "disable", "fixed"])
    if plan_mode == "disable":
        kw["disable_split_kv"] = True
    elif plan_mode == "fixed":
        kw["fixed_split_size"] = rng.choice([128, 256, 1000])
    cache, indptr, indices, last = make_paged(batch, kv_lens, ps, hkv, d, kvdt, layout, seed=seed)
    torch.manual_seed(seed)
    q = torch.randn(sum(qo_lens), hq, d).to(qdt)
    qo_indptr = torch.tensor([0] + list(torch.tensor(qo_lens).cumsum(0)), dtype=torch.int32)
    mask = None
    if variant == "mask":
        mask = torch.cat([(torch.rand(a, b) < 0.7).view(-1) for a, b in zip(qo_lens, kv_lens)])
        kw["custom_mask"] = mask.to(DEV)
    ws = torch.zeros(256 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, layout)
    w.plan(qo_indptr.to(DEV), indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, ps, causal=causal,
           q_data_type=qdt, kv_data_type=kvdt, **kw)
    o, lse = w.run(q.to(DEV), cache.to(DEV), return_lse=True)
    o_ref, lse_ref = R.batch_prefill_ref(q.float(), qo_indptr, cache.float(), layout, indptr, indices, last,
                                         causal=causal, custom_mask=mask, **okw)
    t = ptol(qdt)
    if variant == "rope":
        # the oracle rounds the rotated q / k to the 16-bit type like the kernel; what is left is the occasional
        # element whose rounding flips with the sin / cos precision
        t = dict(rtol=t["rtol"] * 2, atol=t["atol"] * 2)
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), **t)
    # fused RoPE re-rounds the rotated q / k to the 16-bit type (as the reference does in shared memory,
    # prefill.cuh:465-612): with one or two visible keys the lse IS a single logit and carries that rounding
    lt = (2e-2 if qdt == torch.bfloat16 else 4e-3) if variant == "rope" else 2e-3
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=lt, atol=lt)


FP8_SEEDS = range(int(os.environ.get("FI_FUZZ_FP8_SEEDS", "24")))


@pytest.mark.parametrize("seed", FP8_SEEDS)
def test_fuzz_fp8_attention_prefill(seed):
    """Random fp8 attention (q, k, v of one fp8 type, per-head scales) through the paged wrapper: head_dim 64 / 128
    (fp8-native kernel: 4- or 8-wave form, e4m3 / e5m2, one head per wave or mixed) and 256 (upcast kernel), GQA
    groups that do and do not divide a wave, any page size, both layouts, causal or not, automatic / disabled /
    fixed kv split, rows without visible keys -- against the oracle's restatement of the reference's FA3 arithmetic
    (hopper/variants.cuh:71-90) at the fp8 bars of tests/test_prefill_gpu.py."""
    import flashinfer

    rng = random.Random(7000 + seed)
    f8 = rng.choice([torch.float8_e4m3fn, torch.float8_e4m3fn, torch.float8_e5m2])
    d = rng.choice([64, 128, 128, 256])
    hkv = rng.choice([1, 2, 4])
    group = rng.choice([1, 2, 4, 7, 8])
    hq = hkv * group
    ps = rng.choice([1, 4, 8, 16, 32])
    layout = rng.choice(["NHD", "HND"])
    batch = rng.randint(1, 4)
    kv_lens = [rng.choice([1, 63, 64, 65, rng.randint(1, 1500)]) for _ in range(batch)]
    qo_lens = [rng.choice([1, rng.randint(1, 40), rng.randint(1, k + 20)]) for k in kv_lens]
    causal = rng.random() < 0.6
    o_dtype = rng.choice([torch.float16, torch.bfloat16])
    kw = {}
    plan_mode = rng.choice(["auto", "auto", "disable", "fixed"])
    if plan_mode == "disable":
        kw["disable_split_kv"] = True
    elif plan_mode == "fixed":
        kw["fixed_split_size"] = rng.choice([128, 256])
    torch.manual_seed(seed)
    q16 = torch.randn(sum(qo_lens), hq, d).half()
    k16 = torch.randn(sum(kv_lens), hkv, d).half()
    v16 = torch.randn(sum(kv_lens), hkv, d).half()
    q8, sq = R.per_head_symmetric_quant(q16, f8)
    k8, sk = R.per_head_symmetric_quant(k16, f8)
    v8, sv = R.per_head_symmetric_quant(v16, f8)
    pages = [-(-l // ps) for l in kv_lens]
    total = sum(pages) + 2
    indptr = torch.tensor([0] + list(torch.tensor(pages).cumsum(0)), dtype=torch.int32)
    indices = torch.randperm(total)[: sum(pages)].to(torch.int32)
    last = torch.tensor([(l - 1) % ps + 1 for l in kv_lens], dtype=torch.int32)
    cache_f = torch.zeros(total, 2, ps, hkv, d)  # NHD mirror in f32 (fp8 values are exact in it)
    off = 0
    for b, l in enumerate(kv_lens):
        for t in range(l):
            pg = int(indices[int(indptr[b]) + t // ps])
            cache_f[pg, 0, t % ps] = k8[off + t].float()
            cache_f[pg, 1, t % ps] = v8[off + t].float()
        off += l
    cache = (cache_f if layout == "NHD" else cache_f.transpose(2, 3).contiguous()).to(f8)
    qo_indptr = torch.tensor([0] + list(torch.tensor(qo_lens).cumsum(0)), dtype=torch.int32)
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, layout)
    w.plan(qo_indptr.to(DEV), indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, ps, causal=causal,
           q_data_type=f8, kv_data_type=f8, o_data_type=o_dtype, **kw)
    o, lse = w.run(q8.to(DEV), cache.to(DEV), return_lse=True, scale_q=sq.to(DEV), scale_k=sk.to(DEV),
                   scale_v=sv.to(DEV))
    assert o.dtype == o_dtype
    tol8 = 5e-2 if f8 == torch.float8_e4m3fn else 1e-1
    off = 0
    for b in range(batch):
        rows = slice(int(qo_indptr[b]), int(qo_indptr[b + 1]))
        kb, vb = k8[off:off + kv_lens[b]], v8[off:off + kv_lens[b]]
        off += kv_lens[b]
        if causal and qo_lens[b] > kv_lens[b]:
            # rows that see no key: o = 0, lse = -5e4; the oracle is asked for the rows that see at least one
            n_dead = qo_lens[b] - kv_lens[b]
            dead = slice(rows.start, rows.start + n_dead)
            assert float(o[dead].float().abs().max()) == 0.0
            assert bool((lse[dead] < -4.9e4).all())
            rows = slice(rows.start + n_dead, rows.stop)
        o_ref, lse_ref = R.fp8_attention_ref(q8[rows], kb, vb, sq, sk, sv, causal=causal)
        torch.testing.assert_close(o[rows].float().cpu(), o_ref.float(), rtol=tol8, atol=tol8)
        torch.testing.assert_close(lse[rows].cpu(), lse_ref.float(), rtol=1e-3, atol=2e-3)

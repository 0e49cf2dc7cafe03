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

"""Error census of bf16 batch prefill over the seeded fuzz configurations (tests/test_fuzz_gpu.py draws): for each
bf16 case the largest normalised error max |o - o_ref| / (atol + rtol |o_ref|) under a candidate bar, with the
position of the worst element (request, row, head, dim), its query position, the number of visible keys and the
tile it falls in.  Usage: python tools/bf16_error_scan.py <seeds> [rtol atol]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import flashinfer
from oracle import attention_ref as R
from test_decode_gpu import make_paged
from test_fuzz_gpu import _lens

DEV = "cuda:0"
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rtol = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0 ** -7
atol = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-3
worst = []
for seed in range(n_seeds):
    rng = random.Random(2000 + seed)
    d = rng.choice([64, 128, 128, 256]); hkv = rng.choice([1, 2, 4]); group = rng.choice([1, 2, 4, 7, 8]); hq = hkv * group
    ps = rng.choice([1, 3, 4, 8, 16]); layout = rng.choice(["NHD", "HND"]); qdt = rng.choice([torch.float16, torch.bfloat16])
    kvdt = rng.choice([qdt, qdt, torch.float8_e4m3fn]) if d != 256 else qdt
    batch = rng.randint(1, 5); kv_lens = _lens(rng, batch, 2500)
    qo_lens = [rng.choice([0, 1, rng.randint(1, 40), rng.randint(1, max(1, k))]) if rng.random() < 0.8 else rng.randint(1, k + 50) for k in kv_lens]
    if sum(qo_lens) == 0: qo_lens[0] = 1
    causal = rng.random() < 0.6
    variant = rng.choice(["plain", "plain", "rope", "alibi", "window", "softcap", "mask"])
    kw, okw = {}, {}
    if variant == "rope": kw, okw = dict(pos_encoding_mode="ROPE_LLAMA"), dict(pos_encoding_mode="ROPE_LLAMA")
    elif variant == "alibi": kw, okw = dict(pos_encoding_mode="ALIBI"), dict(pos_encoding_mode="ALIBI")
    elif variant == "window":
        wl = rng.randint(1, 300); kw, okw = dict(window_left=wl), dict(window_left=wl)
    elif variant == "softcap": kw, okw = dict(logits_soft_cap=15.0), dict(logits_soft_cap=15.0)
    plan_mode = rng.choice(["auto", "auto", "disable", "fixed"])
    if plan_mode == "disable": kw["disable_split_kv"] = True
    elif plan_mode == "fixed": kw["fixed_split_size"] = rng.choice([128, 256, 1000])
    if qdt != torch.bfloat16 or variant == "rope":
        continue
    cache, indptr, indices, last = make_paged(batch, kv_lens, ps, hkv, d, kvdt, layout, seed=seed)
    torch.manual_seed(seed)
    q = torch.randn(sum(qo_lens), hq, d).to(qdt)
    qo_indptr = torch.tensor([0] + list(torch.tensor(qo_lens).cumsum(0)), dtype=torch.int32)
    mask = None
    if variant == "mask":
        mask = torch.cat([(torch.rand(a, b) < 0.7).view(-1) for a, b in zip(qo_lens, kv_lens)]); kw["custom_mask"] = mask.to(DEV)
    ws = torch.zeros(256 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, layout)
    w.plan(qo_indptr.to(DEV), indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, ps, causal=causal, q_data_type=qdt, kv_data_type=kvdt, **kw)
    o = w.run(q.to(DEV), cache.to(DEV)).float().cpu()
    o_ref, _ = R.batch_prefill_ref(q.float(), qo_indptr, cache.float(), layout, indptr, indices, last, causal=causal, custom_mask=mask, **okw)
    o_ref = o_ref.float()
    err = (o - o_ref).abs() / (atol + rtol * o_ref.abs())
    m = float(err.max()) if err.numel() else 0.0
    idx = int(err.argmax()) if err.numel() else 0
    row, head, dim = idx // (hq * d), (idx // d) % hq, idx % d
    req = int((qo_indptr[1:] > row).nonzero()[0]) if err.numel() else 0
    qi = row - int(qo_indptr[req])
    vis = kv_lens[req] - qo_lens[req] + qi + 1 if causal else kv_lens[req]
    worst.append((m, seed, variant, plan_mode, d, group, kvdt, req, qi, head, dim, vis, float(o.view(-1)[idx]) if err.numel() else 0, float(o_ref.view(-1)[idx]) if err.numel() else 0,
                  float((err > 1).float().mean()) if err.numel() else 0.0))
worst.sort(key=lambda x: -x[0])
print(f"bf16 cases: {len(worst)}; bar rtol={rtol:.5f} atol={atol}; cases over the bar: {sum(1 for x in worst if x[0] > 1)}")
for x in worst[:12]:
    print("norm_err %.3f seed %d %s plan=%s d=%d G=%d kv=%s req %d q_row %d head %d dim %d visible_keys %d got %.6f ref %.6f frac_over %.2e" % x)

#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json config C2 -- BatchDecodeWithPagedKVCacheWrapper, bf16, GQA 32/8,
head_dim 128, page_size 16, batch 64, kv_len 8192 -- on N MI355X GPUs of one node.

A "step" is one pass of the hot path (wrapper.run = decode kernel + split-KV merge kernel) over one batch
of synthetic paged KV that is already resident in HBM.  Work is per-request data-parallel: every rank owns
its own batch of 64 requests (weak scaling) and there is no data-path collective.

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = algorithmic bytes (q + k + v + o, the reference's formula,
flashinfer/testing/utils.py:476-480) of all ranks / the timed wall time (max over ranks).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "flashinfer-ai_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

C2 = dict(batch=64, kv_len=8192, num_qo_heads=32, num_kv_heads=8, head_dim=128, page_size=16)


def algorithmic_bytes_flops(cfg, esize=2):
    b, L, hq, hkv, d = cfg["batch"], cfg["kv_len"], cfg["num_qo_heads"], cfg["num_kv_heads"], cfg["head_dim"]
    q = b * hq * d * esize
    kv = 2 * b * L * hkv * d * esize
    o = b * hq * d * esize
    flops = 2 * b * 1 * L * hq * (d + d)  # ref: flashinfer/testing/utils.py:280-297 (non-causal)
    return q + kv + o, flops


def build_inputs(cfg, device, seed, permute=True):
    b, L, hq, hkv, d, ps = (cfg[k] for k in ("batch", "kv_len", "num_qo_heads", "num_kv_heads", "head_dim", "page_size"))
    g = torch.Generator(device=device).manual_seed(seed)
    pages_per_req = L // ps
    npages = b * pages_per_req
    cache = torch.randn(npages, 2, ps, hkv, d, device=device, dtype=torch.bfloat16, generator=g)
    q = torch.randn(b, hq, d, device=device, dtype=torch.bfloat16, generator=g)
    indptr = (torch.arange(b + 1, dtype=torch.int32) * pages_per_req).to(device)
    if permute:  # random page permutation is the headline case
        indices = torch.randperm(npages, device=device, generator=g).to(torch.int32)
    else:
        indices = torch.arange(npages, device=device, dtype=torch.int32)
    last = torch.full((b,), ps, dtype=torch.int32, device=device)
    return q, cache, indptr, indices, last


def cpu_baseline(cfg, sample_requests=8, iters=3):
    """The reference's CPU path for this op ("torch-CPU SDPA"): pages gathered
    into dense fp32 [B, Hkv, L, d], then F.scaled_dot_product_attention(enable_gqa=True) on the host cores.
    Bounded sample: `sample_requests` requests of the same shape; gather time excluded."""
    import torch.nn.functional as F

    from oracle import attention_ref as R  # checker-side code, allowed here (cpu_baseline leg)

    sub = dict(cfg, batch=sample_requests)
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    q, cache, indptr, indices, last = build_inputs(sub, "cpu", seed=0)
    ks, vs = [], []
    for r in range(sample_requests):
        k, v = R.gather_paged_kv(cache, "NHD", indptr, indices, last, r)
        ks.append(k.float().transpose(0, 1))
        vs.append(v.float().transpose(0, 1))
    k = torch.stack(ks)  # [B, Hkv, L, d]
    v = torch.stack(vs)
    qq = q.float()[:, :, None, :]  # [B, Hq, 1, d]
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        F.scaled_dot_product_attention(qq, k, v, enable_gqa=True)
        times.append(time.perf_counter() - t0)
    t = sorted(times)[len(times) // 2]
    nbytes, _ = algorithmic_bytes_flops(sub)
    return {
        "value": nbytes / t / 1e9,
        "unit": "GB/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{sample_requests} of 64 requests (kv_len 8192, 32/8 heads, d128), fp32 torch-CPU SDPA, "
                  f"median of {iters}, gather excluded, {t * 1e3:.1f} ms",
    }


_L2_FLUSH = {}


def _time_ms(fn, iters, warm, flush_l2=True):
    """median of `iters` event-bracketed launches; the 256 MB buffer written before each one evicts L2 /
    Infinity Cache as the reference's bench does (flashinfer/testing/utils.py:544-545)."""
    dev = torch.cuda.current_device()
    if flush_l2 and dev not in _L2_FLUSH:
        _L2_FLUSH[dev] = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        if flush_l2:
            _L2_FLUSH[dev].zero_()
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2]


def cpu_baseline_c3(b, qo, kv, hq, hkv, d, sample_heads=8, iters=1):
    """torch-CPU SDPA (fp32, explicit bottom-right causal mask) on ONE request and `sample_heads` of the q heads
    of C3's shape; FLOPs counted with the same causal formula."""
    import torch.nn.functional as F

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    group = hq // hkv
    q = torch.randn(1, sample_heads, qo, d, generator=g)
    k = torch.randn(1, sample_heads // group, kv, d, generator=g)
    v = torch.randn(1, sample_heads // group, kv, d, generator=g)
    mask = torch.ones(qo, kv, dtype=torch.bool).tril(kv - qo)
    ts = []
    for _ in range(iters + 1):  # first run warms the allocator / thread pool
        t0 = time.perf_counter()
        F.scaled_dot_product_attention(q, k, v, attn_mask=mask, enable_gqa=True)
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    flops = (2 * kv - qo) * qo * sample_heads * 2 * d
    return {"value": flops / t / 1e12, "unit": "TFLOP/s", "cores": cores, "kind": "port",
            "sample": f"1 of {b} requests, {sample_heads} of {hq} q heads (qo {qo}, kv {kv}, d {d}, causal), fp32 "
                      f"torch-CPU SDPA on dequantised values, {t * 1e3:.0f} ms"}


def cpu_baseline_c4(G, m, n, k, sample_rows=512):
    """dequantise -> fp32 torch.matmul on the host for `sample_rows` rows of ONE group of C4."""
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    a = torch.randn(sample_rows, k, generator=g).to(torch.float8_e4m3fn)
    bm = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.float8_e4m3fn)
    sa = torch.rand(k // 128, sample_rows, generator=g) + 0.5
    sb = torch.rand(k // 128, n // 128, generator=g) + 0.5
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        ad = a.float() * sa.t().repeat_interleave(128, 1)
        bd = bm.float() * sb.t().repeat_interleave(128, 0).repeat_interleave(128, 1)
        torch.matmul(ad, bd.t())
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    flops = 2 * sample_rows * n * k
    return {"value": flops / t / 1e12, "unit": "TFLOP/s", "cores": cores, "kind": "port",
            "sample": f"{sample_rows} rows of 1 of {G} groups (n {n}, k {k}), dequantise + fp32 torch.matmul, "
                      f"{t * 1e3:.0f} ms"}


def _cached_pmc(name):
    """traffic / clock / matrix-pipe occupancy of a secondary kernel from the committed rocprofv3 PMC passes of the
    same config and harness (profiles/r03_<name>_pmc.json, written by tools/prof_secondary.sh): NOT measured by this
    run, labelled as such."""
    path = os.path.join(ROOT, "profiles", f"r03_{name}_pmc.json")
    if not os.path.exists(path):
        return {}
    try:
        d = json.load(open(path))
    except Exception:
        return {}
    keep = {k: d[k] for k in ("traffic", "traffic_note", "clock_ghz", "mfma_busy", "valu_per_mfma", "l2_hit_rate") if k in d}
    keep["pmc_source"] = f"profiles/r03_{name}_pmc.json (rocprofv3 --pmc of this config with the same L2 flush; not this run)"
    return keep


def secondary_workloads(device):
    """BASELINE.json configs C3 (fp8 causal batch prefill) and C4 (fp8 groupwise grouped GEMM), timed on this
    GPU after the headline run: reported next to it, never part of `value`.  FLOP formulas are the
    reference's (flashinfer/testing/utils.py:280-297; benchmarks/bench_groupwise_grouped_gemm_fp8_blackwell.py:51);
    peak = 5 PFLOP/s dense fp8 (MI355X_MICROARCH.md).

    Inputs are SURVEY.md 8(d)'s: quantised with the reference's own schemes.  Both kernels' wall time depends on the
    DATA at an unchanged instruction stream (the chip lowers its clock under matrix-pipe load; values that fill the
    e4m3 range toggle more than N(0, 1) cast to e4m3, zeros toggle nothing: DESIGN.md 3.3), so the r1 / r2 input
    style and all-zero inputs are timed beside the headline entry and labelled."""
    import flashinfer

    out = []
    g = torch.Generator(device=device).manual_seed(1)
    # C3: fp8_e4m3 causal, qo_len 2048, kv_len 8192, bs 16, head_dim 128 (32/8 heads as C2), page 16
    b, qo, kv, hq, hkv, d, ps = 16, 2048, 8192, 32, 8, 128, 16
    npages = b * kv // ps

    # 16-bit randn q / cache -> per-head symmetric quantisation to e4m3, scale = amax / 448 (clamp 1e-6),
    # the reference's helper restated (tests/attention/test_hopper_fp8_attention.py:12-41); the scales go to run()
    def quant_per_head(x, head_axis):
        dims = [i for i in range(x.dim()) if i != head_axis]
        scale = (x.float().abs().amax(dim=dims, keepdim=True) / 448.0).clamp(min=1e-6)
        return (x.float() / scale).to(torch.float8_e4m3fn), scale.flatten().contiguous()

    cache16 = torch.randn(npages, 2, ps, hkv, d, device=device, dtype=torch.float16, generator=g)
    q16 = torch.randn(b * qo, hq, d, device=device, dtype=torch.float16, generator=g)
    k8, scale_k = quant_per_head(cache16[:, 0], 2)
    v8, scale_v = quant_per_head(cache16[:, 1], 2)
    cache = torch.stack([k8, v8], dim=1).contiguous()
    del k8, v8
    q, scale_q = quant_per_head(q16, 1)
    qo_indptr = (torch.arange(b + 1, dtype=torch.int32) * qo).to(device)
    indptr = (torch.arange(b + 1, dtype=torch.int32) * (kv // ps)).to(device)
    indices = torch.randperm(npages, device=device, generator=g).to(torch.int32)
    last = torch.full((b,), ps, dtype=torch.int32, device=device)
    ws = torch.zeros(128 << 20, dtype=torch.uint8, device=device)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, "NHD")
    w.plan(qo_indptr, indptr, indices, last, hq, hkv, d, ps, causal=True, q_data_type=torch.float8_e4m3fn,
           kv_data_type=torch.float8_e4m3fn, o_data_type=torch.bfloat16)
    o = torch.empty(b * qo, hq, d, device=device, dtype=torch.bfloat16)
    flops = b * (2 * kv - qo) * qo * hq * 2 * d
    ms = _time_ms(lambda: w.run(q, cache, out=o, scale_q=scale_q, scale_k=scale_k, scale_v=scale_v), iters=10, warm=3)
    # the same launch on other data (same plan, same kernel)
    cache_r2, q_r2 = cache16.to(torch.float8_e4m3fn), q16.to(torch.float8_e4m3fn)  # r1 / r2 bench inputs: N(0, 1) cast
    del cache16, q16
    ms_r2 = _time_ms(lambda: w.run(q_r2, cache_r2, out=o), iters=10, warm=3)
    cache_r2.zero_()
    q_r2.zero_()
    ms_zero = _time_ms(lambda: w.run(q_r2, cache_r2, out=o), iters=10, warm=3)
    del cache_r2, q_r2
    roof = {"bound": "mfma", "achieved": flops / ms / 1e9, "peak": 5000.0, "unit": "TFLOP/s",
            "frac": flops / ms / 1e9 / 5000.0, "kernel": "fi::batch_prefill_fp8_kernel"}
    roof.update(_cached_pmc("c3"))
    out.append({"workload": "C3: BatchPrefillWithPagedKVCacheWrapper fp8_e4m3 causal qo_len=2048 kv_len=8192 bs=16 "
                            "head_dim=128 GQA 32/8 page_size=16, per-head quantised q/k/v with their scales (SURVEY 8d)",
                "ms": ms, "value": flops / ms / 1e9, "unit": "TFLOP/s", "dtype": "fp8_e4m3", "roofline": roof,
                "same_launch_other_data": {
                    "note": "identical plan and kernel; only the tensor VALUES differ (clock under load is data dependent)",
                    "randn_cast_to_e4m3_no_scales (the r1/r2 bench inputs)": {"ms": ms_r2, "TFLOP/s": flops / ms_r2 / 1e9},
                    "all_zero": {"ms": ms_zero, "TFLOP/s": flops / ms_zero / 1e9}},
                "l2_flush_between_iters": True, "cpu_baseline": cpu_baseline_c3(b, qo, kv, hq, hkv, d)})
    del cache, q, o, w, ws
    torch.cuda.empty_cache()
    # C4: 8 experts, M=4096 per expert, N=14336, K=4096, 128-wide block scales; operands quantised with the
    # reference's block quantiser (flashinfer/testing/utils.py:66-161 restated: scale = amax.clamp(1e-4) / 448 rounded
    # UP to a power of two, tiles (1, 128) for a and (128, 128) for b, "MN"-major scales)
    G, m, n, k = 8, 4096, 14336, 4096

    def quant_block(x, tr, tk):
        gg, rows, kk = x.shape
        xt = x.float().reshape(gg, rows // tr, tr, kk // tk, tk)
        amax = xt.abs().amax(dim=(2, 4)).clamp(1e-4)
        scale = torch.pow(2.0, torch.ceil(torch.log2(amax / 448.0)))
        x8 = (xt / (scale[:, :, None, :, None] + 1e-8)).reshape(gg, rows, kk).to(torch.float8_e4m3fn)
        return x8, scale.transpose(1, 2).contiguous()

    a = torch.empty(G * m, k, device=device, dtype=torch.float8_e4m3fn)
    sa = torch.empty(k // 128, G * m, device=device)
    bm = torch.empty(G, n, k, device=device, dtype=torch.float8_e4m3fn)
    sb = torch.empty(G, k // 128, n // 128, device=device)
    for i in range(G):  # group by group: the f32 staging copies stay small
        q8, s8 = quant_block(torch.randn(1, m, k, device=device, generator=g), 1, 128)
        a[i * m:(i + 1) * m], sa[:, i * m:(i + 1) * m] = q8[0], s8[0]
        q8, s8 = quant_block(torch.randn(1, n, k, device=device, generator=g) / k ** 0.5, 128, 128)
        bm[i], sb[i] = q8[0], s8[0]
    del q8, s8
    m_indptr = (torch.arange(G + 1, dtype=torch.int32) * m).to(device)
    dout = torch.empty(G * m, n, device=device, dtype=torch.bfloat16)
    flops = 2 * G * m * n * k
    ms = _time_ms(lambda: flashinfer.group_gemm_fp8_nt_groupwise(a, bm, sa, sb, m_indptr, out=dout), iters=10, warm=3)
    # r1 / r2 bench inputs: N(0, 1) cast to e4m3, uniform random (not power-of-two) scales -> the general fold path
    a2 = torch.randn(G * m, k, device=device, generator=g).to(torch.float8_e4m3fn)
    b2 = (torch.randn(G, n, k, device=device, generator=g) / k ** 0.5).to(torch.float8_e4m3fn)
    sa2 = torch.rand(k // 128, G * m, device=device, generator=g) + 0.5
    sb2 = torch.rand(G, k // 128, n // 128, device=device, generator=g) + 0.5
    ms_r2 = _time_ms(lambda: flashinfer.group_gemm_fp8_nt_groupwise(a2, b2, sa2, sb2, m_indptr, out=dout), iters=10, warm=3)
    roof = {"bound": "mfma", "achieved": flops / ms / 1e9, "peak": 5000.0, "unit": "TFLOP/s",
            "frac": flops / ms / 1e9 / 5000.0,
            "kernel": "fi::group_gemm_fp8_big_kernel<hardware block scales> (+ scales_pow2_check_kernel)"}
    roof.update(_cached_pmc("c4"))
    out.append({"workload": "C4: group_gemm_fp8_nt_groupwise 8 experts M=4096 N=14336 K=4096 block=128, operands from the "
                            "reference's block quantiser (power-of-two scales; SURVEY 8d)", "ms": ms,
                "value": flops / ms / 1e9, "unit": "TFLOP/s", "dtype": "fp8_e4m3", "roofline": roof,
                "same_shape_other_data": {
                    "randn_cast_to_e4m3, uniform random scales (the r1/r2 bench inputs; general fold path)":
                        {"ms": ms_r2, "TFLOP/s": flops / ms_r2 / 1e9}},
                "l2_flush_between_iters": True, "cpu_baseline": cpu_baseline_c4(G, m, n, k)})
    return out


def c5_cascade(device, world, rank, dist, iters=200, warm=30):
    """BASELINE config C5 on the ranks of this job: cascade shared-prefix batch decode, 64 requests per GPU
    (512 on 8), bf16 GQA 32/8 d128 page 16, shared prefix 8192 tokens + 128 unique tokens per request
    (SURVEY.md 8d).  Two arrangements, both timed with a barrier + synchronize on either side, max over ranks:
      sharded    : prefix pages split over the ranks; every rank attends ALL queries over its shard (prefill
                   kernel, one request of world * 64 rows), states exchanged by ONE RCCL all_to_all_single and
                   merged locally (flashinfer/distributed.py);
      replicated : every rank holds the whole prefix and attends its own 64 queries, no communication (the
                   scaling upper bound SURVEY.md 8e asks for beside it)."""
    import flashinfer
    from flashinfer import distributed as fdist

    B, HQ, HKV, D, PS, PREFIX, UNIQUE = 64, 32, 8, 128, 16, 8192, 128
    g = torch.Generator(device=device).manual_seed(100 + rank)
    u_pages = UNIQUE // PS
    cache_u = torch.randn(B * u_pages, 2, PS, HKV, D, device=device, dtype=torch.bfloat16, generator=g)
    q_local = torch.randn(B, HQ, D, device=device, dtype=torch.bfloat16, generator=g)
    dw = flashinfer.BatchDecodeWithPagedKVCacheWrapper(torch.zeros(64 << 20, dtype=torch.uint8, device=device), "NHD")
    dw.plan((torch.arange(B + 1, dtype=torch.int32) * u_pages).to(device),
            torch.randperm(B * u_pages, device=device, generator=g).to(torch.int32),
            torch.full((B,), PS, dtype=torch.int32, device=device), HQ, HKV, D, PS, q_data_type=torch.bfloat16)

    def prefix_wrapper(tokens, rows):
        pages = tokens // PS
        cache = torch.randn(pages, 2, PS, HKV, D, device=device, dtype=torch.bfloat16, generator=g)
        w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(torch.zeros(128 << 20, dtype=torch.uint8, device=device), "NHD")
        w.plan(torch.tensor([0, rows], dtype=torch.int32, device=device),
               torch.tensor([0, pages], dtype=torch.int32, device=device),
               torch.arange(pages, dtype=torch.int32, device=device),
               torch.tensor([PS], dtype=torch.int32, device=device), HQ, HKV, D, PS, causal=False,
               q_data_type=torch.bfloat16)
        return w, cache

    ex = fdist.SharedPrefixExchange(HQ, D, torch.bfloat16, device, B, always_collective=True)
    pw_s, cache_s = prefix_wrapper(PREFIX // world, B * world)
    pw_r, cache_r = prefix_wrapper(PREFIX, B)

    def sharded():
        return fdist.sharded_shared_prefix_decode(
            q_local, lambda qa: pw_s.run(qa, cache_s, return_lse=True), lambda ql: dw.run(ql, cache_u, return_lse=True),
            flashinfer.merge_states, flashinfer.merge_state, exchange=ex)

    # the one-collective form: q already on every rank (replicated upstream), no all-gather in the step
    q_rep = torch.zeros(B * world, HQ, D, device=device, dtype=torch.bfloat16)
    q_rep[rank * B:(rank + 1) * B] = q_local
    if dist.get_world_size() > 1:
        dist.all_reduce(q_rep)  # setup, outside the timed loops: every rank's rows (the others are zero here)

    def sharded_q_replicated():
        return fdist.sharded_shared_prefix_decode(
            q_local, lambda qa: pw_s.run(qa, cache_s, return_lse=True), lambda ql: dw.run(ql, cache_u, return_lse=True),
            flashinfer.merge_states, flashinfer.merge_state, exchange=ex, q_all=q_rep)

    def replicated():
        v_p, s_p = pw_r.run(q_local, cache_r, return_lse=True)
        v_u, s_u = dw.run(q_local, cache_u, return_lse=True)
        return flashinfer.merge_state(v_p, s_p, v_u, s_u)[0]

    res = {}
    for name, fn in (("sharded", sharded), ("sharded_q_replicated", sharded_q_replicated), ("replicated", replicated)):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res[name] = float(t.item()) / iters
    flat_bytes = 2 * B * (PREFIX + UNIQUE) * HKV * D * 2  # what a flat (non-cascade) decode reads per GPU
    return {
        "workload": f"C5: cascade shared-prefix batch decode bs={B * world} over {world} GPU(s) (64/GPU), bf16 GQA 32/8 "
                    f"d128 page 16, prefix {PREFIX} + {UNIQUE} unique tokens",
        "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
        "collectives_per_step": "sharded: 1 all_gather_into_tensor (q) + 1 all_to_all_single (packed v|lse states); "
                                "sharded_q_replicated: the all_to_all_single alone (q replicated upstream)",
        "us_per_step_sharded_prefix": res["sharded"] * 1e6,
        "us_per_step_sharded_prefix_q_replicated": res["sharded_q_replicated"] * 1e6,  # ONE collective (all_to_all)
        "us_per_step_replicated_prefix": res["replicated"] * 1e6,
        "requests_per_s_sharded": B * world / res["sharded"], "requests_per_s_replicated": B * world / res["replicated"],
        "state_bytes_sent_per_rank_per_step": ex.bytes_sent_per_step,
        "q_bytes_gathered_per_rank_per_step": (world - 1) * B * HQ * D * 2,
        "flat_decode_equivalent_TBps_per_gpu": flat_bytes / res["sharded"] / 1e12,
        "iters": iters,
    }


def main():
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner to stdout
    # when its communicator comes up), so file descriptor 1 is pointed at stderr for the whole run and the result
    # line goes to a private duplicate of the real stdout.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--no-permute", action="store_true", help="arange page table instead of a random permutation")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C3 / C4 side measurements (N=1 only)")
    ap.add_argument("--no-c5", action="store_true", help="N=1: skip the cascade exchange step on a one-rank RCCL group")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FI_BENCH_FORCE_DIST=1 exercises the RCCL code path (init, barrier, all-reduce of the timing) with one rank
    distributed = world > 1 or os.environ.get("FI_BENCH_FORCE_DIST") == "1"
    if args.gpus != world and distributed:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")  # single-rank FI_BENCH_FORCE_DIST run without a launcher
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        import datetime

        dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(minutes=10))

    import flashinfer

    cfg = C2
    # every rank owns its own 64-request batch slice (disjoint pages, own page table): weak scaling
    q, cache, indptr, indices, last = build_inputs(cfg, device, seed=rank, permute=not args.no_permute)
    ws = torch.zeros(128 * 1024 * 1024, dtype=torch.uint8, device=device)
    wrapper = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, "NHD")
    wrapper.plan(indptr, indices, last, cfg["num_qo_heads"], cfg["num_kv_heads"], cfg["head_dim"],
                 cfg["page_size"], pos_encoding_mode="NONE", q_data_type=torch.bfloat16, kv_data_type=torch.bfloat16)
    out = torch.empty_like(q)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # Setup, before the W warm-up steps: the first few dozen launches after an idle period run at a lower
    # clock (power-state ramp, measured ~40 launches); run them here so that a small --warmup does not
    # leave that transient inside the timed region.  The timed region below is exactly K steps.
    for _ in range(64):
        wrapper.run(q, cache, out=out)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        wrapper.run(q, cache, out=out)
    barrier()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record()  # kernels are launched on torch's current stream, where these events live
        wrapper.run(q, cache, out=out)
        ends[i].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per_step = [s.elapsed_time(e) for s, e in zip(starts, ends)]
    kernel_ms = sum(per_step) / args.steps

    nbytes, flops = algorithmic_bytes_flops(cfg)
    c5 = None
    if not distributed and not args.no_c5:
        # N = 1 from the driver: the exchange step still runs through RCCL, on a one-rank group (the collectives
        # degenerate to local copies inside the library, but it is the same code path the N > 1 runs take)
        try:
            import datetime

            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device,
                                    timeout=datetime.timedelta(minutes=5))
            distributed = True
        except Exception as exc:
            c5 = {"error": "one-rank RCCL group: " + repr(exc)}
    if distributed and c5 is None:  # the one exchange step of the path (RCCL all-to-all of shared-prefix states)
        try:
            c5 = c5_cascade(device, world, rank, dist)
        except Exception as exc:
            c5 = {"error": repr(exc)}
    if rank == 0:
        achieved = nbytes / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes (2 * FETCH_SIZE + WRITE_SIZE, gfx950 rule): NOT measured by
        # this run -- a cached rocprofv3 result of the same config, labelled as such; dropped when the config differs
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "r03_c2_decode_traffic.json")
        if os.path.exists(pmc) and not args.no_permute:
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                traffic_source = "profiles/r03_c2_decode_traffic.json (rocprofv3 --pmc of this config and kernel, tools/prof_r03.sh; not this run)"
            except Exception:
                traffic = None
        line = {
            "metric": "paged_decode_hbm_bandwidth_bs64_kv8192_hd128",
            "value": world * nbytes * args.steps / elapsed / 1e9,
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "setup_launches": 64,  # clock-ramp launches before the W warm-up steps (see main)
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": "C2: BatchDecodeWithPagedKVCacheWrapper bf16 GQA 32/8 head_dim=128 page_size=16 "
                            "bs=64/GPU kv_len=8192, random page permutation" + (" (arange)" if args.no_permute else ""),
                "batch_per_gpu": cfg["batch"], "kv_len": cfg["kv_len"], "num_qo_heads": cfg["num_qo_heads"],
                "num_kv_heads": cfg["num_kv_heads"], "head_dim": cfg["head_dim"], "page_size": cfg["page_size"],
                "split_kv": bool(wrapper._plan_info[9]), "kv_chunk_size": int(wrapper._plan_info[10]),
                "parallelism": f"batch-shard x{world} (no data-path collective)",
            },
            "tflops": world * flops * args.steps / elapsed / 1e12,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": "fi::decode_mfma16_kernel (+ merge_n_kernel)",
                "kernel_ms": kernel_ms,
                "kernel_ms_median": sorted(per_step)[len(per_step) // 2],
                "kernel_ms_min": min(per_step),
                "kernel_ms_max": max(per_step),
                "algorithmic_bytes_per_launch": nbytes,
            },
        }
        if c5 is not None:
            line["c5"] = c5
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        if world == 1 and not args.no_secondary:
            del cache, q, out
            torch.cuda.empty_cache()
            try:
                line["secondary"] = secondary_workloads(device)
            except Exception as exc:  # the headline line must survive a failure of the side measurements
                line["secondary"] = {"error": repr(exc)}
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

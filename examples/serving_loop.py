#!/usr/bin/env python3
"""A minimal serving loop on the MI355X FlashInfer path: chunked prefill into a paged KV cache, then
graph-captured batch decode steps (append one token per request, attend, repeat).

    PYTHONPATH=flashinfer-ai_amd python examples/serving_loop.py

Everything below is the reference's public API (flashinfer.page / prefill / decode); nothing is specific to
this build except that it runs on gfx950.  Weights are random: the point is the data flow.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd"))

import torch  # noqa: E402

import flashinfer  # noqa: E402


def main(batch=8, prompt_len=700, new_tokens=16, hq=32, hkv=8, d=128, page_size=16, dtype=torch.bfloat16):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    max_len = prompt_len + new_tokens
    pages_per_req = (max_len + page_size - 1) // page_size
    cache = torch.zeros(batch * pages_per_req, 2, page_size, hkv, d, dtype=dtype, device=dev)
    # every request owns a fixed set of pages (a real allocator would hand them out on demand)
    page_table = torch.randperm(batch * pages_per_req, device=dev).to(torch.int32).view(batch, pages_per_req)

    def table(lens):
        """CSR page table for the given per-request lengths."""
        n_pages = [(l + page_size - 1) // page_size for l in lens]
        indptr = torch.tensor([0] + list(torch.tensor(n_pages).cumsum(0)), dtype=torch.int32, device=dev)
        indices = torch.cat([page_table[i, :n] for i, n in enumerate(n_pages)])
        last = torch.tensor([(l - 1) % page_size + 1 for l in lens], dtype=torch.int32, device=dev)
        return indptr, indices, last

    ws = torch.zeros(128 << 20, dtype=torch.uint8, device=dev)

    # ---- prefill: append the prompt's K/V (RoPE applied once, at append time), then causal attention ----
    lens = [prompt_len] * batch
    indptr, indices, last = table(lens)
    qo_indptr = (torch.arange(batch + 1, dtype=torch.int32) * prompt_len).to(dev)
    q = torch.randn(batch * prompt_len, hq, d, dtype=dtype, device=dev)
    k = torch.randn(batch * prompt_len, hkv, d, dtype=dtype, device=dev)
    v = torch.randn_like(k)
    pos = torch.arange(prompt_len, dtype=torch.int32, device=dev).repeat(batch)
    flashinfer.apply_rope_pos_ids_inplace(q, k, pos)
    bi, bp = flashinfer.get_batch_indices_positions(qo_indptr, flashinfer.get_seq_lens(indptr, last, page_size),
                                                   batch * prompt_len)
    flashinfer.append_paged_kv_cache(k, v, bi, bp, cache, indices, indptr, last)
    prefill = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, "NHD")
    prefill.plan(qo_indptr, indptr, indices, last, hq, hkv, d, page_size, causal=True, q_data_type=dtype)
    out = prefill.run(q, cache)
    print("prefill:", tuple(out.shape), "finite:", bool(torch.isfinite(out.float()).all()))

    # ---- decode: fixed-shape graph-mode wrapper; plan() per step rewrites the work list, run() is replayed ----
    max_pages = batch * pages_per_req
    decode = flashinfer.BatchDecodeWithPagedKVCacheWrapper(
        ws, "NHD", use_cuda_graph=True,
        paged_kv_indptr_buffer=torch.zeros(batch + 1, dtype=torch.int32, device=dev),
        paged_kv_indices_buffer=torch.zeros(max_pages, dtype=torch.int32, device=dev),
        paged_kv_last_page_len_buffer=torch.zeros(batch, dtype=torch.int32, device=dev))
    q1 = torch.zeros(batch, hq, d, dtype=dtype, device=dev)
    o1 = torch.zeros_like(q1)
    graph = None
    one = (torch.arange(batch + 1, dtype=torch.int32)).to(dev)
    for step in range(new_tokens):
        lens = [prompt_len + step + 1] * batch
        indptr, indices, last = table(lens)
        # the new token's q / k / v (a model would produce them); rotate at its position and append
        qn = torch.randn(batch, hq, d, dtype=dtype, device=dev)
        kn = torch.randn(batch, hkv, d, dtype=dtype, device=dev)
        vn = torch.randn_like(kn)
        pos1 = torch.full((batch,), prompt_len + step, dtype=torch.int32, device=dev)
        flashinfer.apply_rope_pos_ids_inplace(qn, kn, pos1)
        bi, bp = flashinfer.get_batch_indices_positions(one, flashinfer.get_seq_lens(indptr, last, page_size), batch)
        flashinfer.append_paged_kv_cache(kn, vn, bi, bp, cache, indices, indptr, last)
        decode.plan(indptr, indices, last, hq, hkv, d, page_size, q_data_type=dtype)
        q1.copy_(qn)
        if graph is None:
            decode.run(q1, cache, out=o1)  # warm-up, then capture once
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                decode.run(q1, cache, out=o1)
        graph.replay()
    torch.cuda.synchronize()
    print("decode :", tuple(o1.shape), "finite:", bool(torch.isfinite(o1.float()).all()), "steps:", new_tokens)
    return out, o1


if __name__ == "__main__":
    main()

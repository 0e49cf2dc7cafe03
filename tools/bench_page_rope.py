"""HBM-bound helpers next to the attention path: page append (prefill-sized and decode-sized) and RoPE."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")
hkv, hq, d, ps = 8, 32, 128, 16

def append(b, n_new, L):
    pages_per = (L + ps - 1) // ps
    cache = torch.zeros(b * pages_per, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16)
    k = torch.randn(b * n_new, hkv, d, device=DEV, dtype=torch.bfloat16); v = torch.randn_like(k)
    indptr = (torch.arange(b + 1, dtype=torch.int32) * pages_per).to(DEV)
    indices = torch.randperm(b * pages_per, device=DEV).to(torch.int32)
    last = torch.full((b,), (L - 1) % ps + 1, dtype=torch.int32, device=DEV)
    append_indptr = (torch.arange(b + 1, dtype=torch.int32) * n_new).to(DEV)
    seq_lens = flashinfer.get_seq_lens(indptr, last, ps)
    bi, pos = flashinfer.get_batch_indices_positions(append_indptr, seq_lens, b * n_new)
    med, _ = bench(lambda: flashinfer.append_paged_kv_cache(k, v, bi, pos, cache, indices, indptr, last))
    nbytes = 2 * 2 * k.numel() * 2  # read + write, K and V
    print(f"append bs={b} new={n_new}/req: {med*1e3:7.1f} us  {nbytes/med/1e6:8.1f} GB/s (read+write)", flush=True)

def rope(n):
    q = torch.randn(n, hq, d, device=DEV, dtype=torch.bfloat16); k = torch.randn(n, hkv, d, device=DEV, dtype=torch.bfloat16)
    pos = torch.arange(n, device=DEV, dtype=torch.int32)
    med, _ = bench(lambda: flashinfer.apply_rope_pos_ids_inplace(q, k, pos))
    nbytes = 2 * (q.numel() + k.numel()) * 2
    print(f"rope in place n={n}: {med*1e3:7.1f} us  {nbytes/med/1e6:8.1f} GB/s (read+write)", flush=True)

def rope_append(b, n_new, L):
    """RoPE on q, k then append k, v: two kernels vs the fused one; bytes = q read + write, k and v read + write"""
    pages_per = (L + ps - 1) // ps
    cache = torch.zeros(b * pages_per, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16)
    q = torch.randn(b * n_new, hq, d, device=DEV, dtype=torch.bfloat16)
    k = torch.randn(b * n_new, hkv, d, device=DEV, dtype=torch.bfloat16); v = torch.randn_like(k)
    indptr = (torch.arange(b + 1, dtype=torch.int32) * pages_per).to(DEV)
    indices = torch.randperm(b * pages_per, device=DEV).to(torch.int32)
    last = torch.full((b,), (L - 1) % ps + 1, dtype=torch.int32, device=DEV)
    append_indptr = (torch.arange(b + 1, dtype=torch.int32) * n_new).to(DEV)
    seq_lens = flashinfer.get_seq_lens(indptr, last, ps)
    bi, pos = flashinfer.get_batch_indices_positions(append_indptr, seq_lens, b * n_new)
    q_out, k_out = torch.empty_like(q), torch.empty_like(k)
    def two():
        flashinfer.rope._run(q, k, q_out, k_out, pos, None, False, 1.0, 1e4)
        flashinfer.append_paged_kv_cache(k_out, v, bi, pos, cache, indices, indptr, last)
    med2, _ = bench(two)
    med1, _ = bench(lambda: flashinfer.apply_rope_append_paged_kv_cache(q, k, v, bi, pos, cache, indices, indptr, last, q_out=q_out))
    nbytes = 2 * (q.numel() + 2 * k.numel()) * 2
    print(f"rope + append bs={b} new={n_new}/req: two kernels {med2*1e3:7.1f} us, fused {med1*1e3:7.1f} us  "
          f"{nbytes/med1/1e6:8.1f} GB/s (algorithmic read+write)", flush=True)

append(16, 2048, 8192); append(64, 1, 8192); rope(32768); rope(64)
rope_append(16, 2048, 8192); rope_append(64, 1, 8192)

"""Two of the reference's own benchmark grids on this library (synthetic inputs generated the way those files do):
  decode   benchmarks/bench_batch_decode.py:82-96   BatchDecodeWithPagedKVCacheWrapper(use_tensor_cores=True), q bf16,
           kv bf16 / fp8_e4m3, batch 1..512 x seq 512..16384, 32 / 4 heads, head_dim 128, page 16, identity page order;
           GB/s = (q + kv cache bytes) / median time (the file's formula, MiB-based "GB")
  prefill  benchmarks/bench_hopper_fp8_attention.py:20-75  single_prefill_with_kv_cache_return_lse, f16 and fp8 e4m3 q / k /
           v, seq 4096 / 8192 / 16384, 24 / 32 heads (MHA), causal and not, head_dim 64 / 128 / 256; TFLOP/s by the file's
           formula (causal counted as half)
  mixed    benchmarks/bench_batch_attention.py (see mixed_grid)
  hopper   benchmarks/bench_hopper_attention.py (see hopper_grid)
Usage: python tools/bench_ref_grids.py decode|prefill|mixed|hopper"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import flashinfer
from bench_decode_sweep import bench
DEV = torch.device("cuda:0")


def decode_grid():
    page, hkv, hq, d = 16, 4, 32, 128
    ws = torch.empty(128 << 20, dtype=torch.uint8, device=DEV)
    print("kv_dtype  batch  seq_len     ms      GB/s (reference formula)   TB/s (10^12)", flush=True)
    for kv_dtype in (torch.bfloat16, torch.float8_e4m3fn):
        for bs in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
            for seq in (512, 1024, 2048, 4096, 8192, 16384):
                blocks = bs * ((seq + page - 1) // page)
                kv_indptr = (torch.arange(bs + 1, dtype=torch.int32) * (blocks // bs)).to(DEV)
                last = torch.full((bs,), seq - (blocks // bs - 1) * page, dtype=torch.int32, device=DEV)
                q = torch.rand(bs, hq, d, dtype=torch.bfloat16, device=DEV)
                kv = torch.empty(blocks, 2, page, hkv, d, device=DEV, dtype=torch.bfloat16).normal_().to(kv_dtype)
                w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, kv_layout="NHD", use_tensor_cores=True)
                w.plan(kv_indptr, torch.arange(blocks, dtype=torch.int32, device=DEV), last, hq, hkv, d, page,
                       data_type=kv_dtype, q_data_type=torch.bfloat16)
                med, _ = bench(lambda: w.run(q, kv), iters=15, warm=3)
                io = q.numel() * q.element_size() + kv.numel() * kv.element_size()
                print(f"{str(kv_dtype).replace('torch.', ''):14s} {bs:5d} {seq:7d} {med:8.4f} {io / med / 1024 / 1024:10.1f} {io / med / 1e9:10.3f}", flush=True)
                del kv, w


def prefill_grid():
    print("seq_len heads causal head_dim   f16 TFLOP/s   fp8-e4m3 TFLOP/s", flush=True)
    for seq in (4096, 8192, 16384):
        for heads in (24, 32):
            for causal in (True, False):
                for d in (64, 128, 256):
                    q = torch.randn(seq, heads, d, dtype=torch.half, device=DEV)
                    k = torch.randn(seq, heads, d, dtype=torch.half, device=DEV)
                    v = torch.randn(seq, heads, d, dtype=torch.half, device=DEV)
                    m16, _ = bench(lambda: flashinfer.single_prefill_with_kv_cache_return_lse(q, k, v, causal=causal), iters=7, warm=2)
                    q8, k8, v8 = (t.to(torch.float8_e4m3fn) for t in (q, k, v))
                    m8, _ = bench(lambda: flashinfer.single_prefill_with_kv_cache_return_lse(
                        q8, k8, v8, causal=causal, backend="fa3", o_dtype=torch.half), iters=7, warm=2)
                    fl = seq * seq * heads * d * (2 if causal else 4)
                    print(f"{seq:7d} {heads:5d} {int(causal):6d} {d:8d} {fl / m16 / 1e9:12.1f} {fl / m8 / 1e9:14.1f}", flush=True)




def mixed_grid():
    """benchmarks/bench_batch_attention.py:96-152: BatchPrefillWithPagedKVCacheWrapper on six (kv_len, qo_len) mixes
    (decode-only, prefill-only, two hybrids, two random), page 1 / 8 / 16, head_dim 64 / 128, 28 / 4 heads, bf16, causal;
    GB/s = (q + cache bytes) / time with the file's 1024-based units, plus TFLOP/s by the reference's causal formula."""
    import numpy as np
    np.random.seed(42)
    torch.random.manual_seed(42)
    cfgs = [[(8192, 1)] * 128, [(4096, 128)] * 4, [(600, 1)] * 122 + [(10_000, 17)] * 8, [(8192, 1)] * 127 * 2 + [(8192, 4096)] * 1]

    def rand_case(bsz, lo, hi):
        full = np.random.randint(lo, hi, size=bsz)
        return [(int(kv), 17) if i % 16 == 0 else (int(kv * 0.05), 1) for i, kv in enumerate(full)]
    cfgs.append(rand_case(256, 1000, 8192))
    cfgs.append(rand_case(128, 2000, 16_000))
    names = ["decode-only 128 x kv 8192", "prefill-only 4 x (4096, 128)", "hybrid 122 x (600, 1) + 8 x (10000, 17)",
             "chunked prefill 254 x (8192, 1) + (8192, 4096)", "random 256", "random 128"]
    hq, hkv = 28, 4
    ws = torch.empty(128 << 20, dtype=torch.uint8, device=DEV)
    print("config                                            page head_dim     ms    GiB/s   TFLOP/s", flush=True)
    for name, pairs in zip(names, cfgs):
        for page in (1, 8, 16):
            for d in (64, 128):
                kv_lens = torch.tensor([p[0] for p in pairs], dtype=torch.int32)
                q_lens = torch.tensor([p[1] for p in pairs], dtype=torch.int32)
                blocks_per = torch.ceil(kv_lens / page).int()
                q_indptr = torch.cat([torch.tensor([0]), torch.cumsum(q_lens, 0)]).int()
                kv_indptr = torch.cat([torch.tensor([0]), torch.cumsum(blocks_per, 0)]).int()
                nb = int(kv_indptr[-1])
                q = torch.rand(int(q_indptr[-1]), hq, d, dtype=torch.bfloat16, device=DEV)
                kv = torch.randn(nb, 2, page, hkv, d, dtype=torch.bfloat16, device=DEV)
                w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, kv_layout="NHD", backend="fa2")
                w.plan(q_indptr.to(DEV), kv_indptr.to(DEV), torch.arange(nb, dtype=torch.int32, device=DEV),
                       ((kv_lens - 1) % page + 1).to(DEV), hq, hkv, d, page, causal=True,
                       q_data_type=torch.bfloat16, kv_data_type=torch.bfloat16)
                med, _ = bench(lambda: w.run(q, kv), iters=9, warm=2)
                byt = q.numel() * 2 + kv.numel() * 2
                fl = sum((2 * kv_ - q_) * q_ * hq * 2 * d for kv_, q_ in pairs)
                print(f"{name:48s} {page:5d} {d:8d} {med:8.4f} {byt / med / 1e-3 / 1024 ** 3:8.1f} {fl / med / 1e9:8.1f}", flush=True)
                del kv, w


def hopper_grid():
    """benchmarks/bench_hopper_attention.py:196-210: f16, 32 / 32 heads, head_dim 128, causal, batch x seq_len = 131 072
    tokens; paged prefill with separate K / V page tensors at page_size 1 and 16, and ragged prefill.  TFLOP/s by the
    file's formula (causal counted as half)."""
    hq = hkv = 32; d = 128
    print("kind    page  batch  seq_len      ms   TFLOP/s", flush=True)
    for page in (1, 16):
        for bs, seq in ((128, 1024), (64, 2048), (32, 4096), (16, 8192), (1, 32768)):
            q = torch.randn(bs * seq, hq, d, dtype=torch.half, device=DEV)
            k = torch.randn(bs * seq // page, page, hkv, d, dtype=torch.half, device=DEV)
            v = torch.randn(bs * seq // page, page, hkv, d, dtype=torch.half, device=DEV)
            w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(torch.empty(256 << 20, dtype=torch.uint8, device=DEV), kv_layout="NHD", backend="fa2")
            qo_indptr = torch.arange(0, bs * seq + 1, seq).int()
            kv_indptr = torch.arange(0, bs * seq // page + 1, seq // page).int()
            w.plan(qo_indptr.to(DEV), kv_indptr.to(DEV), torch.arange(0, bs * seq // page, dtype=torch.int32, device=DEV),
                   torch.ones(bs, dtype=torch.int32, device=DEV) * page, hq, hkv, d, page, causal=True)
            med, _ = bench(lambda: w.run(q, (k, v)), iters=7, warm=2)
            print(f"paged  {page:5d} {bs:6d} {seq:8d} {med:8.3f} {bs * seq * seq * hq * d * 2 / med / 1e9:9.1f}", flush=True)
            del k, v, w
    for bs, seq in ((128, 1024), (64, 2048), (32, 4096), (16, 8192), (1, 32768)):
        q = torch.randn(bs * seq, hq, d, dtype=torch.half, device=DEV)
        k = torch.randn(bs * seq, hkv, d, dtype=torch.half, device=DEV)
        v = torch.randn(bs * seq, hkv, d, dtype=torch.half, device=DEV)
        w = flashinfer.BatchPrefillWithRaggedKVCacheWrapper(torch.empty(256 << 20, dtype=torch.uint8, device=DEV), kv_layout="NHD", backend="fa2")
        ind = torch.arange(0, bs * seq + 1, seq).int().to(DEV)
        w.plan(ind, ind, hq, hkv, d, causal=True)
        med, _ = bench(lambda: w.run(q, k, v), iters=7, warm=2)
        print(f"ragged     - {bs:6d} {seq:8d} {med:8.3f} {bs * seq * seq * hq * d * 2 / med / 1e9:9.1f}", flush=True)


if __name__ == "__main__":
    {"decode": decode_grid, "prefill": prefill_grid, "mixed": mixed_grid, "hopper": hopper_grid}[sys.argv[1]]()

"""Two of the reference's own benchmark grids on this library (synthetic inputs generated the way those files do):
  decode   benchmarks/bench_batch_decode.py:82-96   BatchDecodeWithPagedKVCacheWrapper(use_tensor_cores=True), q bf16,
           kv bf16 / fp8_e4m3, batch 1..512 x seq 512..16384, 32 / 4 heads, head_dim 128, page 16, identity page order;
           GB/s = (q + kv cache bytes) / median time (the file's formula, MiB-based "GB")
  prefill  benchmarks/bench_hopper_fp8_attention.py:20-75  single_prefill_with_kv_cache_return_lse, f16 and fp8 e4m3 q / k /
           v, seq 4096 / 8192 / 16384, 24 / 32 heads (MHA), causal and not, head_dim 64 / 128 / 256; TFLOP/s by the file's
           formula (causal counted as half)
Usage: python tools/bench_ref_grids.py decode|prefill"""
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


if __name__ == "__main__":
    (decode_grid if sys.argv[1] == "decode" else prefill_grid)()

"""Host-side cost of one run() call (Python wrapper + C ABI validation + launches), measured as the CPU time
of back-to-back calls while the GPU queue absorbs them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd"))
import torch, flashinfer
DEV = torch.device("cuda:0")
hq, hkv, d, ps, b = 32, 8, 128, 16, 8
cache = torch.randn(b * 8, 2, ps, hkv, d, device=DEV, dtype=torch.bfloat16)
indptr = (torch.arange(b + 1, dtype=torch.int32) * 8).to(DEV); indices = torch.arange(b * 8, dtype=torch.int32, device=DEV)
last = torch.full((b,), ps, dtype=torch.int32, device=DEV)
ws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t = (time.perf_counter() - t0) / n; torch.cuda.synchronize(); return t * 1e6
dw = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, "NHD"); dw.plan(indptr, indices, last, hq, hkv, d, ps, q_data_type=torch.bfloat16)
q = torch.randn(b, hq, d, device=DEV, dtype=torch.bfloat16); out = torch.empty_like(q)
print(f"decode run(out=):        {timeit(lambda: dw.run(q, cache, out=out)):6.1f} us")
print(f"decode run():            {timeit(lambda: dw.run(q, cache)):6.1f} us")
pw = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, "NHD")
qo_indptr = (torch.arange(b + 1, dtype=torch.int32) * 4).to(DEV)
pw.plan(qo_indptr, indptr, indices, last, hq, hkv, d, ps, causal=True, q_data_type=torch.bfloat16)
qp = torch.randn(b * 4, hq, d, device=DEV, dtype=torch.bfloat16); outp = torch.empty_like(qp)
print(f"prefill run(out=):       {timeit(lambda: pw.run(qp, cache, out=outp)):6.1f} us")
print(f"prefill run():           {timeit(lambda: pw.run(qp, cache)):6.1f} us")
t0 = time.perf_counter()
for _ in range(50): dw.plan(indptr, indices, last, hq, hkv, d, ps, q_data_type=torch.bfloat16)
print(f"decode plan():           {(time.perf_counter()-t0)/50*1e6:6.1f} us")
t0 = time.perf_counter()
for _ in range(50): pw.plan(qo_indptr, indptr, indices, last, hq, hkv, d, ps, causal=True, q_data_type=torch.bfloat16)
print(f"prefill plan():          {(time.perf_counter()-t0)/50*1e6:6.1f} us")
k = torch.randn(128, hkv, d, device=DEV, dtype=torch.bfloat16); v = torch.randn_like(k); q1 = torch.randn(hq, d, device=DEV, dtype=torch.bfloat16)
print(f"single_decode:           {timeit(lambda: flashinfer.single_decode_with_kv_cache(q1, k, v)):6.1f} us")
a, bb = torch.randn(b, hq, d, device=DEV, dtype=torch.bfloat16), torch.randn(b, hq, device=DEV)
print(f"merge_state:             {timeit(lambda: flashinfer.merge_state(a, bb, a, bb)):6.1f} us")

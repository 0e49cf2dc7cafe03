"""One process for rocprofv3: C3 or C4 exactly as bench.py's secondary_workloads() times them (SURVEY 8d inputs, the
256 MB L2 flush before every launch), N launches.  usage: python tools/prof_secondary_once.py c3|c4 [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd"))
import torch
import flashinfer

which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
device = torch.device("cuda", 0)
g = torch.Generator(device=device).manual_seed(1)
flush = torch.empty(256 << 20, dtype=torch.uint8, device=device)
if which == "c3":
    b, qo, kv, hq, hkv, d, ps = 16, 2048, 8192, 32, 8, 128, 16
    npages = b * kv // ps

    def quant_per_head(x, head_axis):
        dims = [i for i in range(x.dim()) if i != head_axis]
        scale = (x.float().abs().amax(dim=dims, keepdim=True) / 448.0).clamp(min=1e-6)
        return (x.float() / scale).to(torch.float8_e4m3fn), scale.flatten().contiguous()

    cache16 = torch.randn(npages, 2, ps, hkv, d, device=device, dtype=torch.float16, generator=g)
    q16 = torch.randn(b * qo, hq, d, device=device, dtype=torch.float16, generator=g)
    k8, scale_k = quant_per_head(cache16[:, 0], 2)
    v8, scale_v = quant_per_head(cache16[:, 1], 2)
    cache = torch.stack([k8, v8], dim=1).contiguous()
    q, scale_q = quant_per_head(q16, 1)
    del cache16, q16, k8, v8
    qo_indptr = (torch.arange(b + 1, dtype=torch.int32) * qo).to(device)
    indptr = (torch.arange(b + 1, dtype=torch.int32) * (kv // ps)).to(device)
    indices = torch.randperm(npages, device=device, generator=g).to(torch.int32)
    last = torch.full((b,), ps, dtype=torch.int32, device=device)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(torch.zeros(128 << 20, dtype=torch.uint8, device=device), "NHD")
    w.plan(qo_indptr, indptr, indices, last, hq, hkv, d, ps, causal=True, q_data_type=torch.float8_e4m3fn,
           kv_data_type=torch.float8_e4m3fn, o_data_type=torch.bfloat16)
    o = torch.empty(b * qo, hq, d, device=device, dtype=torch.bfloat16)
    fn = lambda: w.run(q, cache, out=o, scale_q=scale_q, scale_k=scale_k, scale_v=scale_v)
else:
    G, m, n_, k = 8, 4096, 14336, 4096

    def quant_block(x, tr, tk):
        gg, rows, kk = x.shape
        xt = x.float().reshape(gg, rows // tr, tr, kk // tk, tk)
        amax = xt.abs().amax(dim=(2, 4)).clamp(1e-4)
        scale = torch.pow(2.0, torch.ceil(torch.log2(amax / 448.0)))
        return (xt / (scale[:, :, None, :, None] + 1e-8)).reshape(gg, rows, kk).to(torch.float8_e4m3fn), scale.transpose(1, 2).contiguous()

    a = torch.empty(G * m, k, device=device, dtype=torch.float8_e4m3fn)
    sa = torch.empty(k // 128, G * m, device=device)
    bm = torch.empty(G, n_, k, device=device, dtype=torch.float8_e4m3fn)
    sb = torch.empty(G, k // 128, n_ // 128, device=device)
    for i in range(G):
        q8, s8 = quant_block(torch.randn(1, m, k, device=device, generator=g), 1, 128)
        a[i * m:(i + 1) * m], sa[:, i * m:(i + 1) * m] = q8[0], s8[0]
        q8, s8 = quant_block(torch.randn(1, n_, k, device=device, generator=g) / k ** 0.5, 128, 128)
        bm[i], sb[i] = q8[0], s8[0]
    m_indptr = (torch.arange(G + 1, dtype=torch.int32) * m).to(device)
    dout = torch.empty(G * m, n_, device=device, dtype=torch.bfloat16)
    fn = lambda: flashinfer.group_gemm_fp8_nt_groupwise(a, bm, sa, sb, m_indptr, out=dout)
for _ in range(3):
    fn()
torch.cuda.synchronize()
for _ in range(n):
    flush.zero_()
    fn()
torch.cuda.synchronize()
print("done", which, n)

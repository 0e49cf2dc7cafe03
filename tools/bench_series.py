import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd"))
import torch, flashinfer, bench as B
dev = torch.device("cuda:0")
q, cache, indptr, indices, last = B.build_inputs(B.C2, dev, seed=0)
ws = torch.zeros(128 << 20, dtype=torch.uint8, device=dev)
w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, "NHD")
w.plan(indptr, indices, last, 32, 8, 128, 16, q_data_type=torch.bfloat16, kv_data_type=torch.bfloat16)
out = torch.empty_like(q)
n = 300
s = [torch.cuda.Event(enable_timing=True) for _ in range(n)]; e = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
for i in range(n):
    s[i].record(); w.run(q, cache, out=out); e[i].record()
torch.cuda.synchronize()
t = [a.elapsed_time(b) * 1e3 for a, b in zip(s, e)]
print("first 20:", [round(x) for x in t[:20]])
for lo in range(0, n, 50):
    seg = sorted(t[lo:lo + 50]); print(lo, "min", round(seg[0]), "med", round(seg[25]), "p90", round(seg[45]), "max", round(seg[-1]), "mean", round(sum(seg) / 50))

"""Compare a forced GEMM kernel (FI_GEMM_WS_MIN_TILES=0 with FI_GEMM_DMA / FI_GEMM_DMA_TM) with the exact
small-integer result rounded to fp16; prints where they differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd"))
import torch
import flashinfer
DEV = "cuda:0"
for (m, n, k) in [(4, 1032, 256), (4, 1024, 256), (4, 256, 256), (300, 1032, 384), (600, 1032, 128)]:
    torch.manual_seed(1)
    a = torch.randint(-3, 4, (m, k)).float()
    b = torch.randint(-3, 4, (n, k)).float()
    sa = torch.pow(2.0, torch.randint(-1, 2, (k // 128, m)).float())
    sb = torch.pow(2.0, torch.randint(-1, 2, (k // 128, -(-n // 128))).float())
    out = flashinfer.gemm_fp8_nt_groupwise(a.to(torch.float8_e4m3fn).to(DEV), b.to(torch.float8_e4m3fn).to(DEV), sa.to(DEV),
                                           sb.to(DEV), scale_major_mode="MN", out_dtype=torch.float16).float().cpu()
    ref = torch.zeros(m, n, dtype=torch.float64)
    for kb in range(k // 128):
        part = a[:, kb * 128:(kb + 1) * 128].double() @ b[:, kb * 128:(kb + 1) * 128].double().T
        ref += part * sa[kb, :, None].double() * sb[kb].double().repeat_interleave(128)[:n][None]
    bad = (out != ref.float().half().float())
    print((m, n, k), "mismatches", int(bad.sum()), "of", bad.numel())
    if bad.any():
        rows = bad.any(1).nonzero().flatten().tolist()
        cols = bad.any(0).nonzero().flatten().tolist()
        print("  rows", rows[:10], "...", rows[-3:], " cols", cols[:10], "...", cols[-5:], " n tiles", sorted(set(c // 128 for c in cols)))

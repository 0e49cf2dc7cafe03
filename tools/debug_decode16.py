"""Error structure of the 16x16x32 decode kernel on small cases (run with FI_DECODE_MFMA16=1 FI_DECODE_MFMA_MIN_GROUP=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import flashinfer
from oracle import attention_ref as R
from test_decode_gpu import make_paged, run_batch_decode
torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
for d in (128, 64):
    for hq, hkv in ((1, 1), (4, 1), (4, 4)):
        for mode in ("NONE", "ROPE_LLAMA"):
            for kv_lens in ([1], [16], [33], [100]):
                ps = 8
                cache, indptr, indices, last = make_paged(len(kv_lens), kv_lens, ps, hkv, d, torch.float16, "NHD", seed=1)
                torch.manual_seed(0)
                q = torch.randn(len(kv_lens), hq, d).half()
                (o, lse), _ = run_batch_decode(q, cache, "NHD", indptr, indices, last, hq, hkv, d, ps, pos_encoding_mode=mode)
                o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), "NHD", indptr, indices, last, pos_encoding_mode=mode,
                                                    rope_round_dtype=torch.float16 if mode != "NONE" else None)
                err = (o.float().cpu() - o_ref.float()).abs()
                bad = err > 3e-3
                print(f"d={d} hq={hq} hkv={hkv} {mode:10s} kv={kv_lens[0]:4d}: max err {float(err.max()):.4f} bad {int(bad.sum())}/{bad.numel()} lse err {float((lse.cpu()-lse_ref).abs().max()):.4f}", flush=True)
                if bad.any() and kv_lens[0] <= 16 and hq == 1:
                    print("   got ", o[0, 0, :16].float().cpu())
                    print("   want", o_ref[0, 0, :16].float())
                    bd = bad[0, 0].nonzero().flatten()
                    print("   bad dims", bd[:40].tolist())

"""Debug helper: run several decode configurations and print error statistics against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import flashinfer
from oracle import attention_ref as R
from test_decode_gpu import make_paged

DEV = "cuda:0"

def run(tag, kv_lens, hq, hkv, d, ps, dtype=torch.float16, layout="NHD", disable_split=False, shuffle=True, **kw):
    torch.manual_seed(1)
    cache, indptr, indices, last = make_paged(len(kv_lens), kv_lens, ps, hkv, d, dtype, layout, seed=3, shuffle=shuffle)
    q = torch.randn(len(kv_lens), hq, d).to(dtype)
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchDecodeWithPagedKVCacheWrapper(ws, layout)
    w.plan(indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, ps, q_data_type=dtype, kv_data_type=dtype, disable_split_kv=disable_split, **kw)
    o, lse = w.run(q.to(DEV), cache.to(DEV), return_lse=True)
    torch.cuda.synchronize()
    o_ref, lse_ref = R.batch_decode_ref(q.float(), cache.float(), layout, indptr, indices, last, **{k: v for k, v in kw.items() if k in ("pos_encoding_mode", "window_left", "logits_soft_cap")})
    eo = (o.float().cpu() - o_ref.float()).abs()
    el = (lse.cpu() - lse_ref.float()).abs()
    print(f"{tag:40s} split={w._plan_info[9]} chunk={w._plan_info[10]} max|do|={eo.max():.4g} max|dlse|={el.max():.4g} per-req do={[round(float(x),4) for x in eo.amax(dim=(1,2))]}")
    if eo.max() > 1e-2:
        print("   o[0,0,:8]   =", o[0, 0, :8].float().cpu().tolist())
        print("   ref[0,0,:8] =", o_ref[0, 0, :8].float().tolist())
        print("   lse[0,:4]", lse[0, :4].cpu().tolist(), "ref", lse_ref[0, :4].tolist())

run("1req kv16 ps16 G1 nosplit", [16], 4, 4, 128, 16, disable_split=True, shuffle=False)
run("1req kv4 ps16 G1 nosplit", [4], 4, 4, 128, 16, disable_split=True, shuffle=False)
run("1req kv64 ps16 G1 nosplit", [64], 4, 4, 128, 16, disable_split=True, shuffle=False)
run("1req kv54 ps16 G1 nosplit", [54], 4, 4, 128, 16, disable_split=True)
run("1req kv54 ps1 G1 nosplit", [54], 4, 4, 128, 1, disable_split=True)
run("1req kv2048 ps16 G1 split", [2048], 4, 4, 128, 16)
run("8req ps16 G1 split", [54, 97, 512, 1, 2048, 33, 16, 17], 4, 4, 128, 16)
run("8req ps1 G1 split", [54, 97, 512, 1, 2048, 33, 16, 17], 4, 4, 128, 1)
run("8req ps16 G4", [54, 97, 512, 1, 2048, 33, 16, 17], 16, 4, 128, 16)
run("8req ps16 G8 bf16", [54, 97, 512, 1, 2048, 33, 16, 17], 32, 4, 128, 16, dtype=torch.bfloat16)
run("8req ps16 G4 HND", [54, 97, 512, 1, 2048, 33, 16, 17], 16, 4, 128, 16, layout="HND")
run("rope", [54, 700, 1, 2049], 8, 2, 128, 8, pos_encoding_mode="ROPE_LLAMA")
run("alibi", [54, 700, 1, 2049], 8, 2, 128, 8, pos_encoding_mode="ALIBI")
run("window", [54, 700, 5, 1500], 8, 4, 128, 16, window_left=15)
run("softcap", [54, 700, 5, 1500], 8, 4, 128, 16, logits_soft_cap=30.0)
run("d64", [54, 700, 5, 1500], 8, 4, 64, 16)
run("d256", [54, 700, 5, 1500], 8, 4, 256, 16)

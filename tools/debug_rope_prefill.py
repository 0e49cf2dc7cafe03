import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd"))
import flashinfer
from oracle import attention_ref as R
torch.manual_seed(4)
d = 128
for kv_len in [1, 2, 3, 5, 17, 33, 65]:
    q = torch.randn(1, 1, d).half(); k = torch.randn(kv_len, 1, d).half(); v = torch.randn(kv_len, 1, d).half()
    o, lse = flashinfer.single_prefill_with_kv_cache(q.cuda(), k.cuda(), v.cuda(), causal=True, pos_encoding_mode="ROPE_LLAMA", return_lse=True)
    o_ref, lse_ref = R.attention_ref(q.float(), k.float(), v.float(), causal=True, pos_encoding_mode="ROPE_LLAMA")
    print(kv_len, "max err o", (o.float().cpu() - o_ref.float()).abs().max().item(), "lse", lse.item(), lse_ref.item())
# batch wrapper on the same data

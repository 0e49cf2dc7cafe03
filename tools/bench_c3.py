"""C3 only (fp8 e4m3 causal batch prefill, bs 16 x qo 2048 x kv 8192, 32/8 heads, d128, page 16): median / min of
N event-timed launches (FI_MI355_LIB selects a library variant built by tools/build_ko.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd"))
import torch
import flashinfer
from bench_prefill import run

if __name__ == "__main__":
    tag = "C3 fp8"
    if os.environ.get("FI_MI355_LIB"):
        tag += " " + os.path.basename(os.environ["FI_MI355_LIB"]).replace("libfi_prefill_fp8_inst_", "KO=").replace(".so", "")
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
        run(torch.float8_e4m3fn, tag=tag)
    if len(sys.argv) > 2 and sys.argv[2] == "quick":
        sys.exit(0)
    run(torch.float8_e4m3fn, causal=False, tag=tag + " non-causal")
    run(torch.float8_e4m3fn, b=4, qo=8192, kv=8192, tag=tag + " bs4 8k/8k")

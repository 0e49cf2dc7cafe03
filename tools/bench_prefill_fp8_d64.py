"""fp8 (e4m3) causal batch prefill at head_dim 64, C3-like shape (bs 16 x qo 2048 x kv 8192, GQA 32/8): the fp8-native
kernel (default) against the upcast kernel (FI_PREFILL_FP8_NATIVE_D64=0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_prefill import run
run(torch.float8_e4m3fn, d=64, tag="fp8 d64 " + os.environ.get("FI_PREFILL_FP8_NATIVE_D64", "native"))
run(torch.float8_e5m2, d=128, tag="e5m2 d128")

"""Prefill throughput across shapes callers actually use (page sizes, head dims, group sizes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_prefill import run
run(torch.bfloat16, tag="C3-shape bf16 page 16")
run(torch.bfloat16, ps=1, tag="page_size 1")
run(torch.bfloat16, hq=64, hkv=8, tag="G=8 (64/8)")
run(torch.bfloat16, hq=32, hkv=32, b=8, tag="MHA 32/32 bs8")
run(torch.bfloat16, d=64, tag="head_dim 64")
run(torch.bfloat16, d=256, hq=16, hkv=4, tag="head_dim 256 (16/4)")
run(torch.float8_e4m3fn, ps=1, tag="fp8 page_size 1")
run(torch.bfloat16, b=1, qo=8192, kv=8192, tag="bs1 8k/8k")
run(torch.bfloat16, b=1, qo=32768, kv=32768, hq=8, hkv=2, tag="bs1 32k/32k 8/2")

import sys, os
sys.path.insert(0, "tools")
import torch
from bench_prefill import run
run(torch.bfloat16, b=16, qo=2048, kv=8192, hq=16, hkv=8, d=256, tag="bf16 d256 G=2")
run(torch.bfloat16, b=16, qo=2048, kv=8192, hq=32, hkv=8, d=128, tag="bf16 d128 G=4")
run(torch.bfloat16, b=16, qo=2048, kv=8192, hq=32, hkv=8, d=64, tag="bf16 d64 G=4")
run(torch.float8_e4m3fn, b=16, qo=2048, kv=8192, hq=16, hkv=8, d=256, tag="fp8 d256 G=2")
run(torch.float8_e4m3fn, b=16, qo=2048, kv=8192, hq=32, hkv=8, d=256, tag="fp8 d256 G=4")

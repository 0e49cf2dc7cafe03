import sys; sys.argv=['x']
sys.path.insert(0,'tools')
import torch
from bench_prefill import run
run(torch.bfloat16, tag="C3-shape bf16")
run(torch.bfloat16, causal=False, tag="C3-shape bf16 non-causal")
run(torch.bfloat16, b=32, qo=1024, kv=1024, hq=64, hkv=8, tag="ref sample shape")

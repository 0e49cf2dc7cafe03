"""Short-sequence prefill shapes (prologue / epilogue dominated): A/B against another build via FI_MI355_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_prefill import run
for _ in range(2):
    run(torch.bfloat16, b=32, qo=1024, kv=1024, hq=64, hkv=8, tag="bs32 1k/1k 64/8")
    run(torch.bfloat16, b=64, qo=512, kv=512, hq=32, hkv=8, tag="bs64 512/512 32/8")
    run(torch.bfloat16, b=1, qo=128, kv=32768, hq=32, hkv=8, tag="bs1 append 128 on 32k")
    run(torch.bfloat16, b=4, qo=16, kv=16384, hq=32, hkv=8, tag="bs4 append 16 on 16k")

import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_prefill import run
dt = {"bf16": torch.bfloat16, "fp8": torch.float8_e4m3fn, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
run(dt, tag="prof")

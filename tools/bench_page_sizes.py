"""Decode / prefill throughput against the page size (page_size = 1 is what token-granular allocators use)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_decode_sweep import run
for ps in (1, 2, 4, 8, 16, 32):
    run(wpc=8, ps=ps, tag=f"C2 page_size={ps}")
run(wpc=8, ps=1, hq=64, tag="G=8 page_size=1")
run(wpc=8, ps=1, dtype=torch.float8_e4m3fn, tag="fp8 kv page_size=1")

import sys, os
sys.path.insert(0, "tools")
import torch
from bench_decode_sweep import run
run(d=256, hq=32, hkv=8, L=4096, tag="d256 G=4 kv4096")
run(d=256, hq=16, hkv=16, L=4096, tag="d256 G=1")
run(d=256, hq=64, hkv=8, L=4096, tag="d256 G=8")
run(d=128, hq=32, hkv=8, L=8192, tag="d128 G=4 (ref)")

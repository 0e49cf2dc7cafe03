"""Plain batch decode on the VALU kernel vs the 16x16x32 matrix-core kernel over shapes (run twice: default, and
with FI_DECODE_MFMA16=1 FI_DECODE_MFMA_MIN_GROUP=1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_decode_sweep import run
run(tag="C2 bf16 G=4")
run(hq=8, tag="G=1")
run(hq=16, tag="G=2")
run(hq=64, tag="G=8")
run(hq=128, tag="G=16")
run(b=256, L=2048, tag="bs256 kv2048")
run(b=16, L=32768, tag="bs16 kv32768")
run(b=1, L=131072, tag="bs1 kv131072")
run(b=8, L=1024, tag="bs8 kv1024")
run(ps=1, tag="page 1")
run(d=64, hq=32, hkv=8, tag="d64")
run(dtype=torch.float8_e4m3fn, tag="fp8 kv")
run(b=64, L=512, tag="bs64 kv512")

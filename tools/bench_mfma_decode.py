"""Wide-group decode: matrix-core kernel vs the VALU kernel (FI_DECODE_MFMA_MIN_GROUP=0 disables MFMA)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_decode_sweep import run
tag = "mfma" if os.environ.get("FI_DECODE_MFMA_MIN_GROUP", "5") != "0" else "valu"
run(wpc=8, hq=64, tag=f"G=8 hq=64 bs64 kv8192 [{tag}]")
run(wpc=8, hq=128, tag=f"G=16 hq=128 bs64 kv8192 [{tag}]")
run(wpc=8, hq=64, hkv=4, tag=f"G=16 hq=64/4 bs64 kv8192 [{tag}]")
run(wpc=8, hq=40, hkv=8, tag=f"G=5 hq=40 bs64 kv8192 [{tag}]")
run(wpc=8, hq=64, b=4, L=65536, tag=f"G=8 bs4 kv65536 [{tag}]")
run(wpc=8, hq=64, b=256, L=2048, tag=f"G=8 bs256 kv2048 [{tag}]")
run(wpc=8, hq=64, layout="HND", tag=f"G=8 HND [{tag}]")

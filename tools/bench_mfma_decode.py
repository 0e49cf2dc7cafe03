"""Wide-group decode: matrix-core kernel vs the VALU kernel.
FI_DECODE_MFMA_MIN_GROUP / FI_DECODE_MFMA_MIN_GROUP_FP8 move the crossover (0 disables the MFMA path)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_decode_sweep import run
tag = "mg=%s mg8=%s" % (os.environ.get("FI_DECODE_MFMA_MIN_GROUP", "-"), os.environ.get("FI_DECODE_MFMA_MIN_GROUP_FP8", "-"))
which = sys.argv[1] if len(sys.argv) > 1 else "wide"
if which == "wide":
    run(wpc=8, hq=64, tag=f"G=8 hq=64 bs64 kv8192 [{tag}]")
    run(wpc=8, hq=128, tag=f"G=16 hq=128 bs64 kv8192 [{tag}]")
    run(wpc=8, hq=64, hkv=4, tag=f"G=16 hq=64/4 bs64 kv8192 [{tag}]")
    run(wpc=8, hq=40, hkv=8, tag=f"G=5 hq=40 bs64 kv8192 [{tag}]")
    run(wpc=8, hq=64, b=4, L=65536, tag=f"G=8 bs4 kv65536 [{tag}]")
    run(wpc=8, hq=64, b=256, L=2048, tag=f"G=8 bs256 kv2048 [{tag}]")
    run(wpc=8, hq=64, layout="HND", tag=f"G=8 HND [{tag}]")
    run(wpc=8, hq=64, dtype=torch.float8_e4m3fn, tag=f"G=8 fp8 kv [{tag}]")
else:  # narrow groups: where is the crossover?
    run(wpc=8, tag=f"C2 G=4 [{tag}]")
    run(wpc=8, hq=16, tag=f"G=2 [{tag}]")
    run(wpc=8, hq=8, tag=f"G=1 [{tag}]")
    run(wpc=8, dtype=torch.float8_e4m3fn, tag=f"C2 fp8 kv G=4 [{tag}]")
    run(wpc=8, hq=8, dtype=torch.float8_e4m3fn, tag=f"fp8 kv G=1 [{tag}]")
    run(wpc=8, b=1, L=131072, tag=f"bs1 kv131072 [{tag}]")

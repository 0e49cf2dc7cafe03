import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_decode_sweep import run
tile = os.environ.get("FI_DECODE_MAX_HEAD_TILE", "8")
run(wpc=8, hq=64, tag=f"G=8 (hq=64) max_tile={tile}")
run(wpc=16, hq=64, tag=f"G=8 (hq=64) max_tile={tile}")
run(wpc=8, b=1, L=131072, tag="bs1 kv131072")
run(wpc=8, b=4, L=65536, tag="bs4 kv65536")

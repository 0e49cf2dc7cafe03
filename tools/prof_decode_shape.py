import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from bench_decode_sweep import run
b, L, hq, hkv = (int(x) for x in sys.argv[1:5])
run(wpc=8, hq=hq, hkv=hkv, b=b, L=L, tag=f"prof {hq}/{hkv} bs{b} kv{L}")

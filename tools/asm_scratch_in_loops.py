#!/usr/bin/env python3
"""For every loop (backward branch) of the kernels in a `hipcc -S` file whose name contains <substr>: number of
MFMAs, global/buffer loads and scratch (spill) operations inside it.  A scratch_load inside a loop that also issues
global loads or LDS-DMA drags a vmcnt(0) wait behind those loads.  usage: asm_scratch_in_loops.py k.s [substr]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for k0 in starts:
    name = lines[k0].split(":")[0]
    if want not in name:
        continue
    k1 = next(i for i in range(k0, len(lines)) if "s_endpgm" in lines[i])
    seg = lines[k0:k1]
    lab = {}
    for i, l in enumerate(seg):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            lab[m.group(1)] = i
    print(name[:110])
    for i, l in enumerate(seg):
        m = re.search(r"s_cbranch\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if not m:
            continue
        t = m.group(1) or m.group(2)
        if t in lab and lab[t] < i:
            body = seg[lab[t]:i]
            nm = sum("v_mfma" in x for x in body)
            if nm == 0:
                continue
            sc = sum("scratch_" in x for x in body)
            vm = sum(bool(re.search(r"\b(global|buffer)_load", x)) for x in body)
            print(f"  loop {t:12s} {len(body):5d} lines  mfma {nm:3d}  vmem loads {vm:3d}  scratch ops {sc}")

#!/usr/bin/env python3
"""Static checks of the built gfx950 ISA for the kernels that schedule by hand (no GPU needed).

The fp8 prefill kernel and the DMA GEMM kernels issue MFMAs, LDS reads and waits from `asm volatile` statements.  For
an asm statement hipcc pads no hazard and counts no memory operation (cdna_hip_programming.md 5.7), so the rules the
source relies on are re-checked here on the compiler's output, after every build:

  R1  result latency of an asm MFMA: the first instruction that reads or writes its destination registers -- other
      than an MFMA that takes them whole as its C operand (accumulate chain) -- must be at least passes + 2 wait
      states later on EVERY path (one wait state per instruction, N + 1 for `s_nop N`, P for an intervening MFMA of
      P passes: it cannot issue before the pipe is free).
  R2  operands of an asm MFMA: no vector instruction in the two wait states in front of it writes one of its
      source registers (VALU write -> MFMA read needs 2 states; the asm statements open with `s_nop 1` for it).
  R3  every `s_barrier` directly follows an `s_waitcnt` (the pair is ONE asm statement: the barrier builtin alone is
      no fence for LDS reads, which hipcc hoisted above it in r2), the wait of a hand-written pair names `vmcnt`
      with one of the counts documented for the kernel, and it drains the LDS counter (`lgkmcnt(0)`) wherever LDS
      reads of the finished step may be in flight.
  R4  no scratch access inside a loop that issues MFMAs (a `scratch_load` beside LDS-DMA draws `vmcnt(0)`).

    python tools/isa_guard.py            # compiles the three units to build/isa/*.s and checks them
"""
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flashinfer-ai_amd", "csrc")
OUT = os.path.join(ROOT, "build", "isa")

# unit -> (extra flags, {kernel-name substring: allowed vmcnt counts of hand-written wait + barrier pairs})
UNITS = {
    "prefill_fp8_inst": (["-fno-slp-vectorize"], {"batch_prefill_fp8_kernel": {0, 2, 3, 4, 6, 8, 12}}),
    "gemm": ([], {"group_gemm_fp8_dma_kernel": {0, 8}}),
    "gemm_big": ([], {"group_gemm_fp8_big_kernel": {0}}),
    # 16-bit prefill at head_dim 256: asm MFMAs with O / Q in accumulator registers (no hand-written barriers)
    "prefill_inst:bf16_256": (["-DFI_PF_T16=1", "-DFI_PF_KVS=1", "-DFI_PF_QS=1", "-DFI_PF_D=256"], {"batch_prefill_kernel": None}),
    "prefill_inst:f16_256": (["-DFI_PF_T16=0", "-DFI_PF_KVS=0", "-DFI_PF_QS=0", "-DFI_PF_D=256"], {"batch_prefill_kernel": None}),
}


def mfma_passes(op):
    if "f8f6f4" in op:
        return 16 if "32x32" in op else 8
    if "32x32" in op:
        return 8 if ("x16" in op or "x32" in op) else 16
    return 4 if ("x32" in op or "x64" in op or "x128" in op) else 8


def compile_unit(unit, force=False):
    """hipcc -S of one translation unit; cached on the hash of every source and header it can include."""
    flags, _ = UNITS[unit]
    src_unit = unit.split(":")[0]  # "file:variant" = the same source with the variant's -D flags
    os.makedirs(OUT, exist_ok=True)
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".h", ".hip")) and (f.endswith(".h") or f == src_unit + ".hip"):
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "fi_mi355.h"), "rb").read())
    h.update(" ".join(flags).encode())
    path = os.path.join(OUT, f"{unit.replace(':', '_')}.s")
    stamp = path + ".sha"
    if not force and os.path.exists(path) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return path
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{ROOT}/include", f"-I{CSRC}", "-fno-gpu-rdc",
           "-DFI_BUILDING_LIB", "--cuda-device-only", "-S", "-o", path] + flags + [os.path.join(CSRC, src_unit + ".hip")]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    open(stamp, "w").write(h.hexdigest())
    return path


REG = re.compile(r"\b([vas])(\d+)\b|\b([vas])\[(\d+):(\d+)\]")


def regs_of(text, kinds="va"):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            if m.group(1) in kinds:
                out.add((m.group(1), int(m.group(2))))
        elif m.group(3) in kinds:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
    return out


class Inst:
    __slots__ = ("op", "args", "in_asm", "line", "defs", "uses", "label")

    def __init__(self, op, args, in_asm, line):
        self.op, self.args, self.in_asm, self.line = op, args, in_asm, line
        ops = [a.strip() for a in args.split(",")] if args else []
        nd = 1
        if op.startswith(("s_cmp", "s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch", "s_endpgm", "ds_write",
                          "global_store", "buffer_store", "scratch_store", "global_load_lds", "s_setprio", "s_sleep",
                          "v_cmpx")):
            nd = 0
        if op.startswith("v_cmp") and not op.startswith("v_cmpx"):
            nd = 1
        if op.startswith("buffer_load") and args.rstrip().endswith("lds"):
            nd = 0
        self.defs = regs_of(",".join(ops[:nd]))
        self.uses = regs_of(",".join(ops[nd:]))
        if op.startswith("v_mfma") and len(ops) >= 4:
            # D = A x B + C: when D and C are the same registers they are read as well
            self.uses |= regs_of(ops[3])
        if op.startswith(("v_fmac", "v_mac", "v_dot2c", "v_pk_fmac")) or "op_sel" in args and op.startswith("v_cvt_pk_fp8"):
            self.uses |= self.defs  # read-modify-write destinations
        self.label = None


def parse(path):
    """-> {kernel name: (list of Inst, {label: index})}, metadata text"""
    kernels = {}
    cur, name, in_asm = None, None, False
    text = open(path).read()
    for ln_no, ln in enumerate(text.split("\n"), 1):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name = m.group(1)
            cur = ([], {})
            kernels[name] = cur
            in_asm = False
            continue
        if cur is None:
            continue
        s = ln.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            cur[1][m.group(1)] = len(cur[0])
            continue
        if not s or s.startswith((";", ".", "//")):
            if s.startswith(".section") or s.startswith(".end_amdhsa_kernel"):
                cur = None
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        parts = s.split(None, 1)
        cur[0].append(Inst(parts[0], parts[1] if len(parts) > 1 else "", in_asm, ln_no))
        if parts[0] == "s_endpgm":
            pass
    return {k: v for k, v in kernels.items() if any(i.op == "s_endpgm" for i in v[0])}, text


def successors(insts, labels, i):
    ins = insts[i]
    if ins.op == "s_endpgm":
        return []
    if ins.op == "s_branch":
        return [labels[ins.args.strip()]] if ins.args.strip() in labels else []
    if ins.op.startswith("s_cbranch"):
        t = ins.args.strip()
        return [i + 1] + ([labels[t]] if t in labels else [])
    return [i + 1] if i + 1 < len(insts) else []


def states_of(ins):
    if ins.op == "s_nop":
        return int(ins.args.strip(), 0) + 1
    if ins.op.startswith("v_mfma"):
        return mfma_passes(ins.op)
    return 1


def check_r1(insts, labels, errors, kname):
    for i, m in enumerate(insts):
        if not (m.in_asm and m.op.startswith("v_mfma")):
            continue
        need = mfma_passes(m.op) + 2
        dst = m.defs
        # DFS over paths: (index, states so far)
        stack = [(s, 0) for s in successors(insts, labels, i)]
        seen = {}
        while stack:
            j, st = stack.pop()
            if st >= need or j >= len(insts):
                continue
            if seen.get(j, 1 << 30) <= st:
                continue
            seen[j] = st
            x = insts[j]
            touches = (x.defs | x.uses) & dst
            if touches:
                chain = x.op.startswith("v_mfma") and x.defs == dst and regs_of(x.args.split(",")[3]) == dst
                if not chain:
                    errors.append(f"R1 {kname}: line {x.line} `{x.op} {x.args}` touches the result of the asm MFMA at "
                                  f"line {m.line} after {st} wait states (needs {need})")
                continue  # the chain MFMA re-defines the registers: its own check covers what follows
            for s in successors(insts, labels, j):
                stack.append((s, st + states_of(x)))


def check_r2(insts, errors, kname):
    for i, m in enumerate(insts):
        if not (m.in_asm and m.op.startswith("v_mfma")):
            continue
        st, j = 0, i - 1
        while j >= 0 and st < 2:
            x = insts[j]
            if x.op.startswith("v_") and not x.op.startswith("v_mfma") and (x.defs & m.uses):
                errors.append(f"R2 {kname}: line {x.line} `{x.op} {x.args}` writes an operand of the asm MFMA at line "
                              f"{m.line}, {st} wait states in front of it (needs 2)")
            st += states_of(x) if not x.op.startswith("v_mfma") else 1
            j -= 1


def check_r3(insts, errors, kname, allowed):
    n = 0
    for i, b in enumerate(insts):
        if b.op != "s_barrier" or not b.in_asm:
            continue  # __syncthreads() is the compiler's own (wait + barrier, nothing in flight that it cannot see)
        n += 1
        p = insts[i - 1] if i else None
        m = re.search(r"vmcnt\((\d+)\)", p.args) if p is not None and p.op == "s_waitcnt" and p.in_asm else None
        if m is None:
            errors.append(f"R3 {kname}: line {b.line} hand-written s_barrier without its vmcnt wait directly in front "
                          f"(previous: `{p.op if p else None} {p.args if p else ''}`)")
        elif allowed is not None and int(m.group(1)) not in allowed:
            errors.append(f"R3 {kname}: line {p.line} vmcnt({m.group(1)}) is not one of the documented counts {sorted(allowed)}")
    return n


def check_r4(insts, labels, errors, kname):
    """no scratch access in the same innermost loop as an MFMA (a scratch_load beside LDS-DMA draws vmcnt(0));
    spills in an outer (per-tile) loop around the k loop are allowed"""
    loops = []
    for i, x in enumerate(insts):
        if x.op.startswith("s_cbranch") or x.op == "s_branch":
            t = labels.get(x.args.strip())
            if t is not None and t <= i:
                loops.append((t, i))

    def innermost(j):
        best = None
        for t, i in loops:
            if t <= j <= i and (best is None or i - t < best[1] - best[0]):
                best = (t, i)
        return best

    mfma_loops = {innermost(j) for j, x in enumerate(insts) if x.op.startswith("v_mfma")}
    mfma_loops.discard(None)
    for j, y in enumerate(insts):
        if y.op.startswith("scratch_"):
            lp = innermost(j)
            if lp in mfma_loops:
                errors.append(f"R4 {kname}: line {y.line} `{y.op} {y.args}` in the innermost loop of an MFMA (lines "
                              f"{insts[lp[0]].line}-{insts[lp[1]].line})")


def check_unit(unit, force=False):
    path = compile_unit(unit, force)
    kernels, text = parse(path)
    _, patterns = UNITS[unit]
    errors, stats = [], {}
    for kname, (insts, labels) in kernels.items():
        pat = next((p for p in patterns if p in kname), None)
        if pat is None:
            continue
        n_asm_mfma = sum(1 for i in insts if i.in_asm and i.op.startswith("v_mfma"))
        check_r1(insts, labels, errors, kname)
        check_r2(insts, errors, kname)
        n_bar = check_r3(insts, errors, kname, patterns[pat])
        stats[kname] = (n_asm_mfma, n_bar)
        # fused-RoPE prefill at head_dim 256 is a coverage path that spills inside its loop (it did with the builtin
        # MFMAs too: 56-92 scratch accesses in r2, 47-95 now); every other instantiation is held to R4
        if not (unit.startswith("prefill_inst") and "ELi256ELb1E" in kname):
            check_r4(insts, labels, errors, kname)
    return errors, stats


def main():
    bad = 0
    for unit in UNITS:
        errors, stats = check_unit(unit, force="--force" in sys.argv)
        print(f"{unit}: {len(stats)} kernels checked, "
              f"{sum(a for a, _ in stats.values())} asm MFMAs, {sum(b for _, b in stats.values())} hand-written barriers, "
              f"{len(errors)} violations")
        for e in errors[:40]:
            print("  ", e)
        bad += len(errors)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

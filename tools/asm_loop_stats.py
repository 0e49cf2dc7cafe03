#!/usr/bin/env python3
"""Instruction mix of the loops of a gfx950 kernel, read from `hipcc -S` output (no GPU needed).

    hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S -o k.s kernel.hip
    python tools/asm_loop_stats.py k.s [kernel-name-substring]

For every kernel: register / LDS / spill figures from the metadata, then for every backward branch (a loop
body = the lines between the branch target label and the branch) the count of MFMA, transcendental, other
vector, LDS, global-memory, scalar and wait instructions, and an issue-cycle estimate from the measured costs of
MI355X_MICROARCH.md (vector 4 cycles, transcendental 8, MFMA holds the issue port 8 of its 32/64 pipe cycles).
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32",
              "v_exp_f16", "v_log_f16", "v_rcp_f16"):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_rd"
    if op.startswith("ds_"):
        return "lds_wr"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vm_ld"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
        return "vm_st"
    if op == "s_waitcnt":
        return "wait"
    if op == "s_barrier":
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def mfma_cycles(op):
    if "32x32x64" in op or "16x16x128" in op:
        return 64 if "32x32" in op else 32
    if "32x32" in op:
        return 32
    return 16


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path).read().split("\n")
    kern = None
    body = {}
    for ln in lines:
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", ln)
        if m and not ln.startswith(".") and not ln.startswith("\t"):
            if m.group(1).startswith("_Z") or kern is None:
                kern = m.group(1)
                body[kern] = []
                continue
        if kern:
            body[kern].append(ln)
    for k, b in body.items():
        if want not in k or not any("s_endpgm" in x for x in b):
            continue
        print("=" * 100)
        print(k)
        meta = "\n".join(lines)
        for key in (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".group_segment_fixed_size",
                    ".private_segment_fixed_size"):
            m = re.search(r"\.name:\s+%s.*?%s:\s+(\d+)" % (re.escape(k), re.escape(key)), meta, re.S)
            m2 = None
            for mm in re.finditer(r"%s:\s+(\d+)" % re.escape(key), meta):
                pass
            if m:
                print("  %s = %s" % (key, m.group(1)))
        labels = {}
        insts = []  # (index, op, text)
        for ln in b:
            s = ln.strip()
            lm = re.match(r"^(\.LBB\w+):", s)
            if lm:
                labels[lm.group(1)] = len(insts)
                continue
            if not s or s.startswith(";") or s.startswith("."):
                continue
            op = s.split()[0]
            insts.append((op, s))
        total = Counter(classify(op) for op, _ in insts)
        print("  whole kernel:", dict(total))
        for i, (op, s) in enumerate(insts):
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = s.split()[-1]
                if tgt in labels and labels[tgt] <= i:
                    seg = insts[labels[tgt]: i + 1]
                    c = Counter(classify(o) for o, _ in seg)
                    mf = sum(mfma_cycles(o) for o, _ in seg if o.startswith("v_mfma"))
                    issue = 4 * c["valu"] + 8 * c["trans"] + 8 * c["mfma"] + 4 * c["lds_rd"] + 4 * c["lds_wr"] + \
                        4 * c["vm_ld"] + 4 * c["vm_st"] + 4 * c["salu"] + 4 * c["nop"]
                    pk = sum(1 for o, _ in seg if o.startswith("v_pk_"))
                    print("  loop %s (%d insts): %s" % (tgt, len(seg), dict(c)))
                    print("      mfma pipe cycles %d, issue-cycle estimate %d (VALU+trans per MFMA: %.1f, v_pk_*: %d)" %
                          (mf, issue, (c["valu"] + c["trans"]) / max(c["mfma"], 1), pk))
                    top = Counter(o for o, _ in seg if classify(o) in ("valu", "trans"))
                    print("      top vector ops:", top.most_common(14))


if __name__ == "__main__":
    main()

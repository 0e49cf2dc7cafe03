"""CPU test: the hand-scheduled kernels' built ISA obeys the rules their source relies on (tools/isa_guard.py).

hipcc neither pads hazards nor counts memory operations for `asm volatile` statements; r2's last commit fixed an LDS
read that had been hoisted above a barrier builtin while the whole GPU suite passed.  This test compiles
prefill_fp8_inst.hip, gemm.hip and gemm_big.hip to gfx950 assembly (no GPU needed; cached under build/isa on the
hash of the sources) and fails when a source edit or another compiler breaks result-read distance of an asm MFMA,
operand-write distance, wait + barrier pairing with the documented vmcnt counts, or puts a scratch access into an
MFMA loop."""
import os
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_guard as G  # noqa: E402

SEEDED = textwrap.dedent("""\
    _Z6seededv:
    .LBB0_1:
    \tv_cvt_pk_fp8_f32 v40, v101, v102
    \t;;#ASMSTART
    \tv_mfma_scale_f32_32x32x64_f8f6f4 v[0:15], v[16:23], v[40:47], v[0:15], v99, v99 op_sel_hi:[0,0,0]
    \t;;#ASMEND
    \tv_add_f32_e32 v50, v51, v52
    \tv_mul_f32_e32 v60, v3, v61
    \tscratch_load_dword v70, off, off offset:4
    \tds_read_b128 v[80:83], v90
    \t;;#ASMSTART
    \ts_waitcnt vmcnt(5)
    \ts_barrier
    \t;;#ASMEND
    \t;;#ASMSTART
    \ts_barrier
    \t;;#ASMEND
    \ts_cbranch_scc1 .LBB0_1
    \ts_endpgm
    """)


def test_guard_reports_seeded_violations(tmp_path):
    """one violation of every rule, so that a parser regression cannot turn the guard into a no-op"""
    f = tmp_path / "seeded.s"
    f.write_text(SEEDED)
    kernels, _ = G.parse(str(f))
    insts, labels = kernels["_Z6seededv"]
    errors = []
    G.check_r1(insts, labels, errors, "seeded")
    G.check_r2(insts, errors, "seeded")
    G.check_r3(insts, errors, "seeded", {0, 6})
    G.check_r4(insts, labels, errors, "seeded")
    text = "\n".join(errors)
    assert "R1" in text and "v_mul_f32_e32" in text          # reads v3 of the MFMA result 1 wait state later
    assert "R2" in text and "v_cvt_pk_fp8_f32" in text       # writes operand v40 right in front of the MFMA
    assert "vmcnt(5) is not one of the documented counts" in text
    assert "without its vmcnt wait" in text                  # the bare barrier
    assert "R4" in text and "scratch_load_dword" in text


def test_guard_accepts_the_legal_forms(tmp_path):
    legal = SEEDED.replace("v_cvt_pk_fp8_f32 v40, v101, v102", "v_cvt_pk_fp8_f32 v140, v101, v102") \
                  .replace("v_mul_f32_e32 v60, v3, v61", "v_mul_f32_e32 v60, v103, v61") \
                  .replace("\tscratch_load_dword v70, off, off offset:4\n", "") \
                  .replace("vmcnt(5)", "vmcnt(6) lgkmcnt(0)") \
                  .replace("\t;;#ASMSTART\n\ts_barrier\n\t;;#ASMEND\n", "")
    f = tmp_path / "legal.s"
    f.write_text(legal)
    kernels, _ = G.parse(str(f))
    insts, labels = kernels["_Z6seededv"]
    errors = []
    G.check_r1(insts, labels, errors, "legal")
    G.check_r2(insts, errors, "legal")
    assert G.check_r3(insts, errors, "legal", {0, 6}) == 1
    G.check_r4(insts, labels, errors, "legal")
    assert errors == []


@pytest.mark.parametrize("unit", sorted(G.UNITS))
def test_built_isa_obeys_the_hand_scheduling_rules(unit):
    errors, stats = G.check_unit(unit)
    assert stats, f"no kernel of {unit} matched {list(G.UNITS[unit][1])}"
    if not unit.startswith("prefill_inst"):
        assert sum(b for _, b in stats.values()) > 0, "no hand-written wait + barrier pair found: parser out of date?"
    if unit.startswith("prefill"):
        assert sum(a for a, _ in stats.values()) > 0, "no asm MFMA found in the prefill kernels"
    assert not errors, "\n".join(errors[:20])

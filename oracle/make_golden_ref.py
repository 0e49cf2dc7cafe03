"""Generates tests/golden/{attention,gemm}_ref_golden.npz from the reference's OWN pure-torch statements of the
hot path's arithmetic.

TEST INFRASTRUCTURE; run ONLY in the build container (needs /root/reference).  The outputs are data (seeded
inputs + the reference functions' outputs) and are committed; no reference source travels or is stored.

The functions live in files whose module top does `import flashinfer` (CUDA-only, not importable here:
ModuleNotFoundError tvm_ffi), but the functions themselves are plain torch (+ einops, installed).  They are
therefore loaded one by one: the file is parsed with `ast`, the wanted FunctionDef nodes are compiled into
a namespace that holds only torch / math / einops, and executed on the CPU (the two `.to("cuda:0")` of
build_causal_mask are redirected to the CPU).

  tests/attention/test_single_prefill.py:9-53      build_causal_mask, _repeat_kv, single_prefill_with_kv_cache_ref
  tests/attention/test_blackwell_fmha.py:11-54     attention_ref  (o and BASE-2 lse)
  tests/attention/test_hopper_fp8_attention.py:12-41  per_head_symmetric_quant
  flashinfer/testing/utils.py:66-216               quantize_fp8, dequantize_fp8
  tests/GEMM/test_groupwise_scaled_gemm_fp8.py:86-192   (the dequantise -> einsum flows, restated in this script
                                                        with the reference's own quantiser / dequantiser)

Usage: python oracle/make_golden_ref.py
"""
import ast
import math
import os

import builtins

import einops
import numpy as np
import torch

REF = "/root/reference"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_functions(rel_path, names, patch=None):
    """Compile the named top-level functions of a reference file into a fresh namespace."""
    src = open(os.path.join(REF, rel_path)).read()
    tree = ast.parse(src)
    # The function bodies come from an untrusted tree: they get torch / math / einops and a short whitelist of
    # builtins -- no open, no __import__, no eval / exec / getattr -- so a changed reference file cannot touch the
    # build container through this script.
    safe_builtins = {k: getattr(builtins, k) for k in (
        "range", "len", "int", "float", "bool", "tuple", "list", "dict", "min", "max", "abs", "sum", "zip",
        "enumerate", "isinstance", "slice", "print", "round", "sorted", "reversed", "map", "any", "all", "str",
        "ValueError", "AssertionError", "RuntimeError", "TypeError", "NotImplementedError", "Exception",
        "True", "False", "None")
        if hasattr(builtins, k)}
    def _import(name, globals=None, locals=None, fromlist=(), level=0):
        if level != 0 or name.split(".")[0] not in ("math", "torch", "einops"):
            raise ImportError(f"import of {name!r} is not allowed in extracted reference functions")
        return builtins.__import__(name, globals, locals, fromlist, level)

    safe_builtins["__import__"] = _import
    ns = {"__builtins__": safe_builtins, "torch": torch, "math": math, "Tuple": tuple, "rearrange": einops.rearrange,
          "reduce": einops.reduce, "repeat": einops.repeat, "einsum": einops.einsum}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            seg = ast.get_source_segment(src, node)
            if patch:
                seg = patch(seg)
            exec(compile(seg, rel_path, "exec"), ns)  # noqa: S102 -- reference test helper, torch only
    missing = [n for n in names if n not in ns]
    assert not missing, missing
    return ns


def f16_exact(*shape, g, scale=1.0):
    """float32 values that are exactly representable in fp16 AND bf16-safe inputs for GPU tests: the GPU
    test converts them to 16 bit without rounding (fp16) and the reference function sees the same values."""
    return (torch.randn(*shape, generator=g) * scale).half().float()


def attention(g):
    data = {}
    sp = load_functions("tests/attention/test_single_prefill.py",
                        ["build_causal_mask", "_repeat_kv", "single_prefill_with_kv_cache_ref"],
                        patch=lambda s: s.replace('"cuda:0"', '"cpu"'))
    # (qo_len, kv_len, Hq, Hkv, D): GQA with Hq != Hkv, append (qo < kv), decode (qo = 1), odd group (7 / 1)
    for tag, (lq, lk, hq, hkv, d) in {"gqa_a": (13, 37, 8, 2, 64), "gqa_b": (40, 40, 7, 1, 128),
                                      "dec": (1, 100, 8, 2, 128), "gqa_c": (70, 130, 12, 4, 64)}.items():
        q, k, v = f16_exact(lq, hq, d, g=g), f16_exact(lk, hkv, d, g=g), f16_exact(lk, hkv, d, g=g)
        # inputs are fp16-exact, so they are stored as fp16 (half the bytes, no information lost)
        data[f"sp_{tag}_q"], data[f"sp_{tag}_k"], data[f"sp_{tag}_v"] = (x.half().numpy() for x in (q, k, v))
        for causal in (False, True):
            o = sp["single_prefill_with_kv_cache_ref"](q, k, v, causal=causal)
            data[f"sp_{tag}_o_{'causal' if causal else 'full'}"] = o.numpy()
    # o + base-2 lse (this statement needs Hq == Hkv: GQA is pinned by the function above)
    bw = load_functions("tests/attention/test_blackwell_fmha.py", ["attention_ref"])
    for tag, (b, lq, lk, h, d) in {"a": (2, 9, 17, 4, 64), "b": (1, 33, 33, 2, 128), "c": (3, 1, 70, 4, 128)}.items():
        q, k, v = f16_exact(b * lq, h, d, g=g), f16_exact(b * lk, h, d, g=g), f16_exact(b * lk, h, d, g=g)
        sm_scale = 1.0 / math.sqrt(d)
        data[f"bw_{tag}_q"], data[f"bw_{tag}_k"], data[f"bw_{tag}_v"] = (x.half().numpy() for x in (q, k, v))
        data[f"bw_{tag}_meta"] = np.array([b, lq, lk], dtype=np.int64)
        for causal in (False, True):
            o, lse = bw["attention_ref"](b, q, k, v, causal, sm_scale)
            c = "causal" if causal else "full"
            data[f"bw_{tag}_o_{c}"], data[f"bw_{tag}_lse_{c}"] = o.numpy(), lse.numpy()  # lse [b, lq, h]
    # merge operator, pinned through split invariance of the reference's own (o, lse): attention over the keys
    # [A | B] must equal merge(state over A, state over B)  (docs/tutorials/recursive_attention.rst:38-52)
    b, lq, lk, h, d = 1, 11, 48, 4, 64
    q, k, v = f16_exact(lq, h, d, g=g), f16_exact(lk, h, d, g=g), f16_exact(lk, h, d, g=g)
    sm_scale = 1.0 / math.sqrt(d)
    cut = 19
    o_full, lse_full = bw["attention_ref"](1, q, k, v, False, sm_scale)
    o_a, lse_a = bw["attention_ref"](1, q, k[:cut], v[:cut], False, sm_scale)
    o_b, lse_b = bw["attention_ref"](1, q, k[cut:], v[cut:], False, sm_scale)
    for name, val in (("o_full", o_full), ("lse_full", lse_full[0]), ("o_a", o_a), ("lse_a", lse_a[0]),
                      ("o_b", o_b), ("lse_b", lse_b[0])):
        data[f"merge_{name}"] = val.numpy()
    # per-head symmetric quantisation (the fp8 attention tests' input maker)
    ph = load_functions("tests/attention/test_hopper_fp8_attention.py", ["per_head_symmetric_quant"])
    x = f16_exact(33, 4, 64, g=g, scale=3.0)
    data["phq_x"] = x.numpy()
    for tag, dt in (("e4m3", torch.float8_e4m3fn), ("e5m2", torch.float8_e5m2)):
        xq, s = ph["per_head_symmetric_quant"](x.half(), dt)
        data[f"phq_{tag}_bytes"], data[f"phq_{tag}_scale"] = xq.view(torch.uint8).numpy(), s.numpy()
    path = os.path.join(OUT_DIR, "attention_ref_golden.npz")
    np.savez_compressed(path, **data)
    print("wrote", os.path.abspath(path), os.path.getsize(path), "bytes")


def gemm(g):
    data = {}
    tu = load_functions("flashinfer/testing/utils.py", ["quantize_fp8", "dequantize_fp8"])
    quantize_fp8, dequantize_fp8 = tu["quantize_fp8"], tu["dequantize_fp8"]
    t = 128
    # ---- 2-D: gemm_fp8_nt_groupwise operands (ref test :86-132), both scale layouts, both A granularities ----
    m, n, k = 12, 128, 256
    a = torch.randn(m, k, generator=g)
    a[3] *= 40.0  # rows of very different magnitude: per-row scales differ
    a[5] *= 1e-6  # below the 1e-4 amax clamp
    bmat = torch.randn(n, k, generator=g) / math.sqrt(k)
    data["g2_a"], data["g2_b"] = a.numpy(), bmat.numpy()
    for mode in ("MN", "K"):
        a_ss, b_ss = ((m, k // t), (n // t, k // t)) if mode == "K" else ((k // t, m), (k // t, n // t))
        a8, a_s = quantize_fp8(a, a_ss, (1, t), mode)
        b8, b_s = quantize_fp8(bmat, b_ss, (t, t), mode)
        a_d, b_d = dequantize_fp8(a8, a_s, mode), dequantize_fp8(b8, b_s, mode)
        c = einops.einsum(a_d, b_d, "m k, n k -> m n")
        for name, val in (("a8", a8.view(torch.uint8)), ("a_s", a_s), ("b8", b8.view(torch.uint8)), ("b_s", b_s),
                          ("a_d", a_d), ("b_d", b_d), ("c", c)):
            data[f"g2_{mode}_{name}"] = val.numpy()
    # (128, 128) granularity for A as well (ref test :35-71, gemm_fp8_nt_blockscaled)
    m2 = 128
    a2 = torch.randn(m2, k, generator=g)
    data["g2b_a"] = a2.numpy()
    for mode in ("MN", "K"):
        a_ss = (m2 // t, k // t) if mode == "K" else (k // t, m2 // t)
        a8, a_s = quantize_fp8(a2, a_ss, (t, t), mode)
        data[f"g2b_{mode}_a8"], data[f"g2b_{mode}_a_s"] = a8.view(torch.uint8).numpy(), a_s.numpy()
        data[f"g2b_{mode}_a_d"] = dequantize_fp8(a8, a_s, mode).numpy()
    # ---- 3-D: grouped GEMM (ref test :135-192) ----
    G, mg, n, k = 3, 8, 128, 256
    a = torch.randn(G * mg, k, generator=g)
    bmat = torch.randn(G, n, k, generator=g) / math.sqrt(k)
    data["g3_a"], data["g3_b"] = a.numpy(), bmat.numpy()
    for mode in ("MN", "K"):
        a_ss = (G * mg, k // t) if mode == "K" else (k // t, mg * G)
        b_ss = (G, n // t, k // t) if mode == "K" else (G, k // t, n // t)
        a8, a_s = quantize_fp8(a, a_ss, (1, t), mode)
        b8, b_s = quantize_fp8(bmat, b_ss, (1, t, t), mode)
        a_d, b_d = dequantize_fp8(a8, a_s, mode), dequantize_fp8(b8, b_s, mode)
        c = einops.einsum(a_d.view((G, mg, k)), b_d, "b m k, b n k -> b m n").view((G * mg, n))
        # ragged groups over the same operands (m_indptr multiples of 4, one empty group): the reference's flow
        # applied group by group
        m_indptr = [0, 4, 4, 24]
        c_ragged = torch.cat([einops.einsum(a_d[m_indptr[i]:m_indptr[i + 1]], b_d[i], "m k, n k -> m n")
                              for i in range(G)])
        for name, val in (("a8", a8.view(torch.uint8)), ("a_s", a_s), ("b8", b8.view(torch.uint8)), ("b_s", b_s),
                          ("a_d", a_d), ("b_d0", b_d[0]), ("c", c), ("c_ragged", c_ragged)):
            data[f"g3_{mode}_{name}"] = val.numpy()
    data["g3_m_indptr_ragged"] = np.array([0, 4, 4, 24], dtype=np.int32)
    path = os.path.join(OUT_DIR, "gemm_ref_golden.npz")
    np.savez_compressed(path, **data)
    print("wrote", os.path.abspath(path), os.path.getsize(path), "bytes")


if __name__ == "__main__":
    gen = torch.Generator().manual_seed(20261004)
    attention(gen)
    gemm(gen)

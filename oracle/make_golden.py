"""Generates tests/golden/rope_alibi_golden.npz from the reference's own importable pure-torch helpers.

Run ONLY in the build container (needs /root/reference); the output is data (inputs + expected outputs)
and is committed, the reference files never travel.  Usage: python oracle/make_golden.py
Helpers used (they import nothing from `flashinfer`):
  /root/reference/tests/test_helpers/rope_reference.py  : generate_cos_sin_f32_cache, apply_rotary_pos_emb
  /root/reference/tests/test_helpers/alibi_reference.py : alibi_attention, get_slopes
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/tests/test_helpers"
sys.path.insert(0, REF)
import alibi_reference  # noqa: E402
import rope_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden",
                   "rope_alibi_golden.npz")


def main():
    g = torch.Generator().manual_seed(20260101)
    data = {}
    # ---- RoPE (non-interleaved / "rotate_half" form == ROPE_LLAMA in-kernel semantics) ----
    for tag, (n, hq, hkv, d, theta) in {"a": (37, 4, 2, 64, 1e4), "b": (70, 2, 1, 128, 5e5)}.items():
        q = torch.randn(n, hq, d, generator=g)
        k = torch.randn(n, hkv, d, generator=g)
        cos, sin = rope_reference.generate_cos_sin_f32_cache(n, d, theta=theta, device="cpu")
        q_ref, k_ref = rope_reference.apply_rotary_pos_emb(q, k, cos, sin)
        data[f"rope_{tag}_q"] = q.numpy()
        data[f"rope_{tag}_k"] = k.numpy()
        data[f"rope_{tag}_theta"] = np.float64(theta)
        data[f"rope_{tag}_q_out"] = q_ref.numpy()
        data[f"rope_{tag}_k_out"] = k_ref.numpy()
    # ---- ALiBi attention (bias = slope * kv_idx, causal mask) ----
    for tag, (qo, kv, h, d) in {"a": (1, 33, 8, 64), "b": (9, 40, 6, 128)}.items():
        q = torch.randn(qo, h, d, generator=g)
        k = torch.randn(kv, h, d, generator=g)
        v = torch.randn(kv, h, d, generator=g)
        mask = torch.tril(torch.ones(qo, kv), diagonal=kv - qo)
        o = alibi_reference.alibi_attention(q, k, v, mask)
        data[f"alibi_{tag}_q"] = q.numpy()
        data[f"alibi_{tag}_k"] = k.numpy()
        data[f"alibi_{tag}_v"] = v.numpy()
        data[f"alibi_{tag}_o"] = o.numpy()
        data[f"alibi_{tag}_slopes"] = alibi_reference.get_slopes(h).numpy()
    np.savez_compressed(OUT, **data)
    print("wrote", os.path.abspath(OUT), {k: v.shape for k, v in data.items()})


if __name__ == "__main__":
    main()

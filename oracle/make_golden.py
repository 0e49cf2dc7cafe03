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
    rope_family(g)


def rope_family(g):
    """tests/golden/rope_family_golden.npz: the standalone RoPE family against rope_reference.py
    (the flows of tests/attention/test_rope.py:33-360, run on the CPU)."""
    out = os.path.join(os.path.dirname(OUT), "rope_family_golden.npz")
    data = {}
    b, n, hq, hk, d, offset = 2, 9, 3, 2, 64, 5
    for tag, rot, llama31 in (("plain_full", 64, False), ("plain_partial", 32, False), ("llama31_full", 64, True)):
        q = torch.randn(b * n, hq, d, generator=g)
        k = torch.randn(b * n, hk, d, generator=g)
        theta = 5e5 if llama31 else 1e4
        freqs_cis = rope_reference.precompute_freqs_cis(rot, n + offset, theta, use_scaled=llama31, device="cpu")
        q_rot, k_rot = rope_reference.apply_rotary_emb(q.reshape(b, n, hq, d)[..., :rot], k.reshape(b, n, hk, d)[..., :rot],
                                                       freqs_cis[offset: offset + n])
        q_out = torch.cat([q_rot, q.reshape(b, n, hq, d)[..., rot:]], -1).reshape(b * n, hq, d)
        k_out = torch.cat([k_rot, k.reshape(b, n, hk, d)[..., rot:]], -1).reshape(b * n, hk, d)
        data[f"{tag}_q"], data[f"{tag}_k"] = q.numpy(), k.numpy()
        data[f"{tag}_q_out"], data[f"{tag}_k_out"] = q_out.numpy(), k_out.numpy()
        data[f"{tag}_meta"] = np.array([b, n, offset, rot, theta, float(llama31)], dtype=np.float64)
    # cos/sin cache form (vLLM-style module, neox and gpt-j pairings, partial rotary)
    for tag, neox, rot in (("cache_neox", True, 64), ("cache_gptj", False, 32)):
        emb = rope_reference.RotaryEmbedding(d, rot, 64, 10000, neox, torch.float32, device="cpu")
        pos = torch.randint(0, 64, (11,), generator=g)
        q = torch.randn(11, hq * d, generator=g)
        k = torch.randn(11, hk * d, generator=g)
        q_out, k_out = emb.forward_native(pos, q.clone(), k.clone())
        data[f"{tag}_q"], data[f"{tag}_k"], data[f"{tag}_pos"] = q.numpy(), k.numpy(), pos.numpy()
        data[f"{tag}_cache"] = emb.cos_sin_cache.numpy()
        data[f"{tag}_q_out"], data[f"{tag}_k_out"] = q_out.numpy(), k_out.numpy()
    np.savez_compressed(out, **data)
    print("wrote", os.path.abspath(out))


if __name__ == "__main__":
    main()

"""CPU restatement of the standalone RoPE family.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Follows include/flashinfer/pos_enc.cuh:476-493 (frequencies + llama-3.1 smoothing), :78-101 / :134-150
(half-split / interleaved rotation) and flashinfer/rope.py:321-1150 (argument meaning).  Pinned against the
reference's own importable helpers (tests/test_helpers/rope_reference.py: apply_rotary_emb,
apply_rotary_pos_emb, apply_scaling, RotaryEmbedding.forward_native) through tests/golden/rope_family_golden.npz.
"""
from __future__ import annotations

import math

import torch


def rope_freqs(rotary_dim: int, interleave: bool, rope_scale: float, rope_theta: float,
               smooth_a: float = 0.0, smooth_b: float = 0.0) -> torch.Tensor:
    """Per-element frequency [rotary_dim] (f64).  ref: pos_enc.cuh:481-493."""
    i = torch.arange(rotary_dim, dtype=torch.float64)
    j = torch.floor(i / 2) if interleave else (i % (rotary_dim // 2))
    freq = (1.0 / rope_theta) ** (2.0 * j / rotary_dim)
    smooth = torch.clamp(freq * smooth_a + smooth_b, 0.0, 1.0)
    return (1 - smooth) * (freq / rope_scale) + smooth * freq


def llama31_smooth(low_freq_factor=1.0, high_freq_factor=4.0, old_context_len=8192):
    """ref: pos_enc.cuh:976-977."""
    return (old_context_len / (2 * math.pi * high_freq_factor - 2 * math.pi * low_freq_factor),
            -1.0 / (high_freq_factor / low_freq_factor - 1.0))


def rotate(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, interleave: bool) -> torch.Tensor:
    """x [..., rot] with per-element cos/sin [..., rot] (pair partners share values)."""
    rot = x.shape[-1]
    if interleave:
        partner = torch.stack((-x[..., 1::2], x[..., 0::2]), dim=-1).flatten(-2)
    else:
        partner = torch.cat((-x[..., rot // 2:], x[..., : rot // 2]), dim=-1)
    return x * cos + partner * sin


def apply_rope_pos_ids_ref(q, k, pos_ids, rotary_dim=None, interleave=False, rope_scale=1.0, rope_theta=1e4,
                           smooth_a=0.0, smooth_b=0.0):
    """q, k [nnz, H, D] -> rotated copies (f64); elements >= rotary_dim pass through."""
    d = q.shape[-1]
    rot = d if rotary_dim is None else rotary_dim
    freq = rope_freqs(rot, interleave, rope_scale, rope_theta, smooth_a, smooth_b)
    ang = pos_ids.to(torch.float64)[:, None] * freq[None, :]
    cos, sin = torch.cos(ang)[:, None, :], torch.sin(ang)[:, None, :]
    outs = []
    for x in (q, k):
        xd = x.to(torch.float64)
        outs.append(torch.cat((rotate(xd[..., :rot], cos, sin, interleave), xd[..., rot:]), dim=-1))
    return outs[0], outs[1]


def positions_from_indptr(indptr, offsets):
    pos = []
    for b in range(len(indptr) - 1):
        n = int(indptr[b + 1]) - int(indptr[b])
        pos.append(torch.arange(n) + int(offsets[b]))
    return torch.cat(pos) if pos else torch.zeros(0, dtype=torch.int64)


def apply_rope_cos_sin_cache_ref(positions, q, k, cos_sin_cache, is_neox=True):
    """q, k [nnz, H, D]; cache [max_pos, rot] = cos | sin halves.  ref: flashinfer/rope.py:1039-1150."""
    rot = cos_sin_cache.shape[1]
    cs = cos_sin_cache[positions.long()].to(torch.float64)
    cos_h, sin_h = cs[:, : rot // 2], cs[:, rot // 2:]
    if is_neox:
        cos, sin = torch.cat((cos_h, cos_h), -1), torch.cat((sin_h, sin_h), -1)
    else:
        cos, sin = cos_h.repeat_interleave(2, -1), sin_h.repeat_interleave(2, -1)
    outs = []
    for x in (q, k):
        xd = x.to(torch.float64)
        outs.append(torch.cat((rotate(xd[..., :rot], cos[:, None], sin[:, None], not is_neox), xd[..., rot:]), -1))
    return outs[0], outs[1]

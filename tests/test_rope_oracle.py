"""CPU: the RoPE-family oracle against golden vectors produced by the reference's own helpers
(tests/test_helpers/rope_reference.py; generator oracle/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import rope_ref as RR

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "rope_family_golden.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN, allow_pickle=False)


@pytest.mark.parametrize("tag", ["plain_full", "plain_partial", "llama31_full"])
def test_interleaved_rope_matches_reference_helper(golden, tag):
    # golden = rope_reference.apply_rotary_emb (complex multiply == interleaved pairs), optionally with the
    # llama-3.1 apply_scaling frequencies (ref flow: tests/attention/test_rope.py:61-88)
    b, n, offset, rot, theta, llama31 = golden[f"{tag}_meta"]
    b, n, offset, rot = int(b), int(n), int(offset), int(rot)
    q, k = torch.from_numpy(golden[f"{tag}_q"]), torch.from_numpy(golden[f"{tag}_k"])
    pos = RR.positions_from_indptr([i * n for i in range(b + 1)], [offset] * b)
    kw = dict(rope_scale=8.0, rope_theta=theta) if llama31 else dict(rope_scale=1.0, rope_theta=theta)
    if llama31:
        kw["smooth_a"], kw["smooth_b"] = RR.llama31_smooth()
    qo, ko = RR.apply_rope_pos_ids_ref(q, k, pos, rotary_dim=rot, interleave=True, **kw)
    np.testing.assert_allclose(qo.numpy(), golden[f"{tag}_q_out"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(ko.numpy(), golden[f"{tag}_k_out"], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("tag,neox", [("cache_neox", True), ("cache_gptj", False)])
def test_cos_sin_cache_rope_matches_reference_module(golden, tag, neox):
    # golden = rope_reference.RotaryEmbedding.forward_native
    d = 64
    q = torch.from_numpy(golden[f"{tag}_q"]).view(11, -1, d)
    k = torch.from_numpy(golden[f"{tag}_k"]).view(11, -1, d)
    cache = torch.from_numpy(golden[f"{tag}_cache"])
    pos = torch.from_numpy(golden[f"{tag}_pos"])
    qo, ko = RR.apply_rope_cos_sin_cache_ref(pos, q, k, cache, is_neox=neox)
    np.testing.assert_allclose(qo.reshape(11, -1).numpy(), golden[f"{tag}_q_out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ko.reshape(11, -1).numpy(), golden[f"{tag}_k_out"], rtol=1e-5, atol=1e-5)


def test_half_split_rope_agrees_with_attention_oracle():
    # the in-kernel ROPE_LLAMA form (oracle/attention_ref.rope_llama, pinned by rope_alibi_golden.npz) is the
    # non-interleaved member of the same family
    from oracle.attention_ref import rope_llama

    torch.manual_seed(0)
    x = torch.randn(13, 3, 64)
    pos = torch.arange(13) + 7
    a = rope_llama(x, pos, 1.0, 1e4)
    b, _ = RR.apply_rope_pos_ids_ref(x, x, pos, interleave=False)
    torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)

"""CPU: the oracle against the committed golden vectors and the reference's known answers."""
import os

import numpy as np
import pytest
import torch

from oracle import attention_ref as R

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "rope_alibi_golden.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN, allow_pickle=False)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_rope_matches_reference_helper(golden, tag):
    # golden = rope_reference.apply_rotary_pos_emb (non-interleaved), positions 0..n-1
    q = torch.from_numpy(golden[f"rope_{tag}_q"])
    k = torch.from_numpy(golden[f"rope_{tag}_k"])
    theta = float(golden[f"rope_{tag}_theta"])
    pos = torch.arange(q.shape[0])
    q_out = R.rope_llama(q, pos, 1.0, theta)
    k_out = R.rope_llama(k, pos, 1.0, theta)
    np.testing.assert_allclose(q_out.numpy(), golden[f"rope_{tag}_q_out"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(k_out.numpy(), golden[f"rope_{tag}_k_out"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_alibi_attention_matches_reference_helper(golden, tag):
    # golden = alibi_reference.alibi_attention with a causal mask; its bias is slope * kv_idx, i.e. the
    # decode kernels' qo_idx = 0 convention.
    q = torch.from_numpy(golden[f"alibi_{tag}_q"])
    k = torch.from_numpy(golden[f"alibi_{tag}_k"])
    v = torch.from_numpy(golden[f"alibi_{tag}_v"])
    o, _ = R.attention_ref(q, k, v, causal=True, pos_encoding_mode="ALIBI", alibi_qo_idx_zero=True)
    np.testing.assert_allclose(o.numpy(), golden[f"alibi_{tag}_o"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(
        R.get_alibi_slopes(q.shape[1]).numpy(), golden[f"alibi_{tag}_slopes"], rtol=1e-6
    )


def test_alibi_row_offset_is_softmax_invariant():
    torch.manual_seed(1)
    q, k, v = torch.randn(5, 4, 32), torch.randn(11, 4, 32), torch.randn(11, 4, 32)
    o0, _ = R.attention_ref(q, k, v, causal=True, pos_encoding_mode="ALIBI", alibi_qo_idx_zero=True)
    o1, _ = R.attention_ref(q, k, v, causal=True, pos_encoding_mode="ALIBI", alibi_qo_idx_zero=False)
    torch.testing.assert_close(o0, o1, rtol=1e-10, atol=1e-10)


def test_plain_attention_matches_torch_sdpa():
    torch.manual_seed(0)
    qo, kv, hq, hkv, d = 7, 29, 8, 2, 64
    q, k, v = torch.randn(qo, hq, d), torch.randn(kv, hkv, d), torch.randn(kv, hkv, d)
    for causal in (False, True):
        o, lse = R.attention_ref(q, k, v, causal=causal)
        mask = torch.ones(qo, kv, dtype=torch.bool).tril(kv - qo) if causal else None
        ref = torch.nn.functional.scaled_dot_product_attention(
            q.transpose(0, 1)[None].double(),
            k.transpose(0, 1)[None].double().repeat_interleave(hq // hkv, 1),
            v.transpose(0, 1)[None].double().repeat_interleave(hq // hkv, 1),
            attn_mask=mask,
        )[0].transpose(0, 1)
        torch.testing.assert_close(o, ref, rtol=1e-10, atol=1e-10)
        # lse is base 2 (ref: tests/attention/test_blackwell_fmha.py:54 -- lse_ref * log2(e))
        logits = torch.einsum("qhd,khd->hqk", q.double(), k.double().repeat_interleave(4, 1)) / d**0.5
        if causal:
            logits = logits.masked_fill(~mask[None], float("-inf"))
        torch.testing.assert_close(
            lse, (torch.logsumexp(logits, -1) * R.LOG2E).transpose(0, 1), rtol=1e-10, atol=1e-10
        )


def test_empty_request_known_answer():
    # ref: tests/attention/test_decode_prefill_lse.py:24-26 -- kv_indptr=[0,0,9], last_page_len=[0,1]
    torch.manual_seed(0)
    page_size, hkv, hq, d = 16, 2, 4, 64
    cache = torch.randn(9, 2, page_size, hkv, d)
    indptr = torch.tensor([0, 0, 9], dtype=torch.int32)
    indices = torch.arange(9, dtype=torch.int32)
    last = torch.tensor([0, 1], dtype=torch.int32)
    assert R.get_seq_lens(indptr, last, page_size).tolist() == [0, 8 * 16 + 1]
    q = torch.randn(2, hq, d)
    o, lse = R.batch_decode_ref(q, cache, "NHD", indptr, indices, last)
    assert torch.all(o[0] == 0) and torch.all(lse[0] == R.NEG_INF_SENTINEL)
    assert torch.isfinite(o[1]).all() and (lse[1] > -100).all()


def test_doc_example_page_table_lengths():
    # ref: flashinfer/decode.py:602-610
    indptr = torch.tensor([0, 17, 29, 44, 48, 66, 100, 128], dtype=torch.int32)
    last = torch.tensor([1, 7, 14, 4, 3, 1, 16], dtype=torch.int32)
    lens = R.get_seq_lens(indptr, last, 16)
    assert lens.tolist() == [257, 183, 238, 52, 275, 529, 448]


def test_merge_is_split_invariant():
    # recursive attention: attention over [A|B] == merge(attention(A), attention(B))
    # (ref: docs/tutorials/recursive_attention.rst:38-52)
    torch.manual_seed(2)
    q, k, v = torch.randn(3, 4, 32), torch.randn(50, 2, 32), torch.randn(50, 2, 32)
    full, lse_full = R.attention_ref(q, k, v)
    parts = [(0, 13), (13, 14), (14, 50)]
    os_, ls_ = zip(*[R.attention_ref(q, k[a:b], v[a:b]) for a, b in parts])
    vm, sm = R.merge_states_ref(torch.stack(os_, 1), torch.stack(ls_, 1))
    torch.testing.assert_close(vm, full, rtol=1e-10, atol=1e-10)
    torch.testing.assert_close(sm, lse_full, rtol=1e-10, atol=1e-10)
    v2, s2 = R.merge_state_ref(*R.merge_state_ref(os_[0], ls_[0], os_[1], ls_[1]), os_[2], ls_[2])
    torch.testing.assert_close(v2, full, rtol=1e-10, atol=1e-10)
    # commutative
    va, sa = R.merge_state_ref(os_[0], ls_[0], os_[2], ls_[2])
    vb, sb = R.merge_state_ref(os_[2], ls_[2], os_[0], ls_[0])
    torch.testing.assert_close(va, vb)
    # an empty state is the identity
    e_v, e_s = torch.zeros_like(os_[0]), torch.full_like(ls_[0], R.NEG_INF_SENTINEL)
    vi, si = R.merge_state_ref(os_[0], ls_[0], e_v, e_s)
    torch.testing.assert_close(vi, os_[0], rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(si, ls_[0], rtol=1e-12, atol=1e-12)


def test_sliding_window_and_soft_cap_definitions():
    torch.manual_seed(3)
    q, k, v = torch.randn(1, 2, 16), torch.randn(20, 2, 16), torch.randn(20, 2, 16)
    # window_left = w: the decode query (position kv_len-1) sees the last w+1 keys
    o_w, _ = R.attention_ref(q, k, v, window_left=4)
    o_t, _ = R.attention_ref(q, k[-5:], v[-5:])
    torch.testing.assert_close(o_w, o_t, rtol=1e-10, atol=1e-10)
    # soft cap -> logits bounded by the cap
    o_c, lse_c = R.attention_ref(q * 50, k, v, logits_soft_cap=1.0)
    assert (lse_c <= (1.0 + np.log(20)) * R.LOG2E + 1e-9).all()


def test_packbits_oracle_on_reference_docstring_vectors():
    """flashinfer/quantization.py:79-82, 121-127 (the reference's own worked examples)."""
    x = torch.tensor([1, 0, 1, 1, 0, 0, 1, 1], dtype=torch.bool)
    assert R.packbits_ref(x).tolist() == [0b10110011]
    x = torch.tensor([1, 0, 1, 1, 0, 0, 1, 1, 1, 0, 1], dtype=torch.bool)
    y, ind = R.segment_packbits_ref(x, torch.tensor([0, 4, 7, 11]), "big")
    assert y.tolist() == [0b10110000, 0b00100000, 0b11010000] and ind.tolist() == [0, 1, 2, 3]


def test_custom_mask_oracle_equals_causal_oracle():
    torch.manual_seed(0)
    q, k, v = torch.randn(9, 4, 16), torch.randn(20, 2, 16), torch.randn(20, 2, 16)
    m = torch.tril(torch.ones(9, 20, dtype=torch.bool), diagonal=11)
    o1, l1 = R.attention_ref(q, k, v, causal=True)
    o2, l2 = R.attention_ref(q, k, v, custom_mask=m)
    assert torch.equal(o1, o2) and torch.equal(l1, l2)

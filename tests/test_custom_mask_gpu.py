"""GPU parity: packbits / segment_packbits and mask mode CUSTOM of the prefill kernels.
Modelled on the reference's tests/utils/test_quantization.py:33-75 (numpy.packbits is the definition) and
tests/attention/test_batch_prefill_kernels.py:734-839 / test_single_prefill.py (custom mask == causal mask
must reproduce the causal run; random masks against the oracle)."""
import numpy as np
import pytest
import torch

from oracle import attention_ref as R
from test_decode_gpu import make_paged
from test_prefill_gpu import ptol

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n", [1, 7, 8, 9, 63, 64, 65, 1000, 4099, 1 << 20])
@pytest.mark.parametrize("bitorder", ["big", "little"])
def test_packbits_is_numpy_packbits(n, bitorder):
    import flashinfer

    torch.manual_seed(n)
    x = torch.rand(n) < 0.5
    y = flashinfer.packbits(x.to(DEV), bitorder)
    assert y.dtype == torch.uint8 and y.shape == ((n + 7) // 8,)
    assert torch.equal(y.cpu(), R.packbits_ref(x, bitorder))
    # an unaligned view (offset by 3 bytes) goes through the byte-wise path
    if n > 16:
        y2 = flashinfer.packbits(x.to(DEV)[3:], bitorder)
        assert torch.equal(y2.cpu(), R.packbits_ref(x[3:], bitorder))


def test_packbits_reference_docstring_vectors():
    # flashinfer/quantization.py:79-82 and :121-127
    import flashinfer

    x = torch.tensor([1, 0, 1, 1, 0, 0, 1, 1], dtype=torch.bool, device=DEV)
    assert flashinfer.packbits(x).tolist() == [0b10110011]
    x = torch.tensor([1, 0, 1, 1, 0, 0, 1, 1, 1, 0, 1], dtype=torch.bool, device=DEV)
    y, ind = flashinfer.segment_packbits(x, torch.tensor([0, 4, 7, 11], device=DEV), bitorder="big")
    assert y.tolist() == [0b10110000, 0b00100000, 0b11010000] and ind.tolist() == [0, 1, 2, 3]
    with pytest.raises(ValueError):
        flashinfer.packbits(x, "middle")


@pytest.mark.parametrize("bitorder", ["big", "little"])
def test_segment_packbits_matches_oracle(bitorder):
    import flashinfer

    torch.manual_seed(2)
    seglens = [0, 1, 8, 9, 0, 63, 4096, 17, 5]
    indptr = torch.tensor([0] + list(np.cumsum(seglens)), dtype=torch.int32)
    x = torch.rand(int(indptr[-1])) < 0.3
    y, new_indptr = flashinfer.segment_packbits(x.to(DEV), indptr.to(DEV), bitorder)
    y_ref, ind_ref = R.segment_packbits_ref(x, indptr, bitorder)
    assert torch.equal(new_indptr.cpu().to(torch.int32), ind_ref)
    assert torch.equal(y.cpu(), y_ref)


@pytest.mark.parametrize("qo_len,kv_len", [(1, 1), (17, 130), (128, 128), (200, 333)])
@pytest.mark.parametrize("hq,hkv", [(4, 4), (8, 2)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_single_prefill_custom_mask(qo_len, kv_len, hq, hkv, dtype):
    import flashinfer

    d = 128
    torch.manual_seed(qo_len * 1000 + kv_len)
    q = torch.randn(qo_len, hq, d).to(dtype)
    k = torch.randn(kv_len, hkv, d).to(dtype)
    v = torch.randn(kv_len, hkv, d).to(dtype)
    # (a) the causal mask given as a custom mask reproduces the causal kernel (ref test_single_prefill.py)
    causal_mask = torch.tril(torch.ones(qo_len, kv_len, dtype=torch.bool), diagonal=kv_len - qo_len)
    o_c = flashinfer.single_prefill_with_kv_cache(q.to(DEV), k.to(DEV), v.to(DEV), causal=True)
    o_m = flashinfer.single_prefill_with_kv_cache(q.to(DEV), k.to(DEV), v.to(DEV), custom_mask=causal_mask.to(DEV))
    torch.testing.assert_close(o_m.float(), o_c.float(), **ptol(dtype))  # two kernels, two tile histories
    # (b) a random mask (every row keeps at least its last key) against the oracle, bool and packed forms
    mask = torch.rand(qo_len, kv_len) < 0.6
    mask[:, -1] = True
    o, lse = flashinfer.single_prefill_with_kv_cache(q.to(DEV), k.to(DEV), v.to(DEV), custom_mask=mask.to(DEV),
                                                     causal=True, return_lse=True)  # causal ignored
    o_ref, lse_ref = R.attention_ref(q.float(), k.float(), v.float(), custom_mask=mask)
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), **ptol(dtype))
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)
    packed = flashinfer.packbits(mask.to(DEV).view(-1), bitorder="little")
    o_p = flashinfer.single_prefill_with_kv_cache(q.to(DEV), k.to(DEV), v.to(DEV), packed_custom_mask=packed)
    assert torch.equal(o_p, o)


def test_custom_mask_fully_masked_row_and_window():
    import flashinfer

    qo_len, kv_len, hq, hkv, d = 40, 100, 4, 2, 64
    torch.manual_seed(8)
    q, k, v = (torch.randn(qo_len, hq, d).half(), torch.randn(kv_len, hkv, d).half(), torch.randn(kv_len, hkv, d).half())
    mask = torch.rand(qo_len, kv_len) < 0.5
    mask[5] = False  # no visible key: o = 0, lse = -5e4 sentinel (ref variant_helper.cuh:81-84)
    o, lse = flashinfer.single_prefill_with_kv_cache(q.to(DEV), k.to(DEV), v.to(DEV), custom_mask=mask.to(DEV),
                                                     window_left=70, return_lse=True)
    o_ref, lse_ref = R.attention_ref(q.float(), k.float(), v.float(), custom_mask=mask, window_left=70)
    assert torch.all(o[5] == 0)
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("ps,layout", [(16, "NHD"), (1, "HND"), (5, "NHD")])
@pytest.mark.parametrize("logits_soft_cap", [0.0, 30.0])
def test_batch_prefill_paged_custom_mask(ps, layout, logits_soft_cap):
    import flashinfer

    hq, hkv, d = 8, 2, 128
    kv_lens = [54, 300, 1, 129, 17]
    qo_lens = [37, 150, 1, 17, 0]
    torch.manual_seed(12)
    cache, indptr, indices, last = make_paged(len(kv_lens), kv_lens, ps, hkv, d, torch.float16, layout, seed=3)
    q = torch.randn(sum(qo_lens), hq, d).half()
    masks = []
    for ql, kl in zip(qo_lens, kv_lens):
        m = torch.rand(ql, kl) < 0.7
        if ql:
            m[:, 0] = True
        masks.append(m.view(-1))
    mask = torch.cat(masks)
    qo_indptr = torch.tensor([0] + list(np.cumsum(qo_lens)), dtype=torch.int32)
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithPagedKVCacheWrapper(ws, layout)
    w.plan(qo_indptr.to(DEV), indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, ps, custom_mask=mask.to(DEV),
           causal=True, logits_soft_cap=logits_soft_cap, q_data_type=torch.float16)
    o, lse = w.run(q.to(DEV), cache.to(DEV), return_lse=True)
    o_ref, lse_ref = R.batch_prefill_ref(q.float(), qo_indptr, cache.float(), layout, indptr, indices, last,
                                         custom_mask=mask, logits_soft_cap=logits_soft_cap)
    torch.testing.assert_close(o.float().cpu(), o_ref.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(lse.cpu(), lse_ref.float(), rtol=1e-3, atol=1e-3)
    # packed form (segment_packbits layout) gives the same bits
    bit_indptr = torch.tensor([0] + list(np.cumsum([a * b for a, b in zip(qo_lens, kv_lens)])), dtype=torch.int32)
    packed, _ = flashinfer.segment_packbits(mask.to(DEV), bit_indptr.to(DEV), bitorder="little")
    w.plan(qo_indptr.to(DEV), indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, ps,
           packed_custom_mask=packed, logits_soft_cap=logits_soft_cap, q_data_type=torch.float16)
    o2 = w.run(q.to(DEV), cache.to(DEV))
    assert torch.equal(o2, o)
    # and planning again without a mask drops it
    w.plan(qo_indptr.to(DEV), indptr.to(DEV), indices.to(DEV), last.to(DEV), hq, hkv, d, ps, causal=True,
           logits_soft_cap=logits_soft_cap, q_data_type=torch.float16)
    o3 = w.run(q.to(DEV), cache.to(DEV))
    o3_ref, _ = R.batch_prefill_ref(q.float(), qo_indptr, cache.float(), layout, indptr, indices, last, causal=True,
                                    logits_soft_cap=logits_soft_cap)
    torch.testing.assert_close(o3.float().cpu(), o3_ref.float(), rtol=1e-3, atol=1e-3)


def test_batch_prefill_ragged_custom_mask():
    import flashinfer

    hq, hkv, d = 4, 4, 64
    kv_lens = [70, 9, 200]
    qo_lens = [70, 3, 128]
    torch.manual_seed(14)
    q = torch.randn(sum(qo_lens), hq, d).half()
    k = torch.randn(sum(kv_lens), hkv, d).half()
    v = torch.randn(sum(kv_lens), hkv, d).half()
    qo_indptr = torch.tensor([0] + list(np.cumsum(qo_lens)), dtype=torch.int32)
    kv_indptr = torch.tensor([0] + list(np.cumsum(kv_lens)), dtype=torch.int32)
    masks = [torch.rand(a, b) < 0.5 for a, b in zip(qo_lens, kv_lens)]
    for m in masks:
        m[:, -1] = True
    mask = torch.cat([m.view(-1) for m in masks])
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    w = flashinfer.BatchPrefillWithRaggedKVCacheWrapper(ws, "NHD")
    w.plan(qo_indptr.to(DEV), kv_indptr.to(DEV), hq, hkv, d, custom_mask=mask.to(DEV), q_data_type=torch.float16)
    o = w.run(q.to(DEV), k.to(DEV), v.to(DEV))
    outs = []
    for b in range(3):
        ob, _ = R.attention_ref(q[qo_indptr[b]:qo_indptr[b + 1]].float(), k[kv_indptr[b]:kv_indptr[b + 1]].float(),
                                v[kv_indptr[b]:kv_indptr[b + 1]].float(), custom_mask=masks[b])
        outs.append(ob)
    torch.testing.assert_close(o.float().cpu(), torch.cat(outs).float(), rtol=1e-3, atol=1e-3)

"""GPU parity: standalone RoPE kernels against the oracle (ref: tests/attention/test_rope.py:33-360).
fp16 inputs/outputs: rtol = atol = 1e-3 as the reference's own tests use."""
import pytest
import torch

from oracle import rope_ref as RR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("interleave", [False, True])
@pytest.mark.parametrize("head_dim,partial", [(64, 1.0), (128, 0.5), (256, 1.0)])
@pytest.mark.parametrize("llama31", [False, True])
@pytest.mark.parametrize("inplace", [False, True])
def test_apply_rope_indptr_form(dtype, interleave, head_dim, partial, llama31, inplace):
    import flashinfer

    torch.manual_seed(0)
    b, n, hq, hk, offset = 3, 45, 8, 2, 100
    rot = int(head_dim * partial)
    nnz = b * n
    qkv = torch.randn(nnz, (hq + 2 * hk) * head_dim).to(dtype).to(DEV)  # packed, non-contiguous q / k views
    q = qkv[:, : hq * head_dim].view(nnz, hq, head_dim)
    k = qkv[:, hq * head_dim: (hq + hk) * head_dim].view(nnz, hk, head_dim)
    indptr = torch.arange(b + 1, dtype=torch.int32, device=DEV) * n
    offsets = torch.full((b,), offset, dtype=torch.int32, device=DEV)
    pos = RR.positions_from_indptr(indptr.cpu(), offsets.cpu())
    if llama31:
        a, bb = RR.llama31_smooth()
        q_ref, k_ref = RR.apply_rope_pos_ids_ref(q.float().cpu(), k.float().cpu(), pos, rot, interleave, 8.0, 5e5, a, bb)
    else:
        q_ref, k_ref = RR.apply_rope_pos_ids_ref(q.float().cpu(), k.float().cpu(), pos, rot, interleave, 1.0, 1e4)
    fn = {(False, False): flashinfer.apply_rope, (False, True): flashinfer.apply_rope_inplace,
          (True, False): flashinfer.apply_llama31_rope, (True, True): flashinfer.apply_llama31_rope_inplace}[(llama31, inplace)]
    res = fn(q, k, indptr, offsets, rotary_dim=rot, interleave=interleave)
    q_out, k_out = (q, k) if inplace else res
    t = dict(rtol=1e-3, atol=1e-3) if dtype == torch.float16 else dict(rtol=2.0 ** -7, atol=8e-3)  # bf16: 1 ulp
    torch.testing.assert_close(q_out.float().cpu(), q_ref.float(), **t)
    torch.testing.assert_close(k_out.float().cpu(), k_ref.float(), **t)


def test_apply_rope_pos_ids_and_cos_sin_cache():
    import flashinfer

    torch.manual_seed(1)
    nnz, hq, hk, d, rot = 77, 4, 2, 128, 64
    q = torch.randn(nnz, hq, d).half().to(DEV)
    k = torch.randn(nnz, hk, d).half().to(DEV)
    pos = torch.randint(0, 4000, (nnz,), dtype=torch.int32, device=DEV)
    q_ref, k_ref = RR.apply_rope_pos_ids_ref(q.float().cpu(), k.float().cpu(), pos.cpu(), rot, False, 1.0, 1e4)
    q_o, k_o = flashinfer.apply_rope_pos_ids(q, k, pos, rotary_dim=rot)
    torch.testing.assert_close(q_o.float().cpu(), q_ref.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(k_o.float().cpu(), k_ref.float(), rtol=1e-3, atol=1e-3)
    # the same rotation through a cos/sin cache (neox pairing) must agree with the computed-angle form
    inv = 1.0 / (1e4 ** (torch.arange(0, rot, 2).double() / rot))
    ang = torch.arange(4000).double()[:, None] * inv[None]
    cache = torch.cat((ang.cos(), ang.sin()), -1).float().to(DEV)
    q2, k2 = flashinfer.apply_rope_with_cos_sin_cache(pos, q.view(nnz, -1), k.view(nnz, -1), d, cache, is_neox=True)
    torch.testing.assert_close(q2.view(nnz, hq, d).float().cpu(), q_ref.float(), rtol=1e-3, atol=1e-3)
    qr, kr = RR.apply_rope_cos_sin_cache_ref(pos.cpu(), q.float().cpu(), k.float().cpu(), cache.cpu(), is_neox=False)
    q3, k3 = q.clone().view(nnz, -1), k.clone().view(nnz, -1)
    flashinfer.apply_rope_with_cos_sin_cache_inplace(pos, q3, k3, d, cache, is_neox=False)
    torch.testing.assert_close(q3.view(nnz, hq, d).float().cpu(), qr.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(k3.view(nnz, hk, d).float().cpu(), kr.float(), rtol=1e-3, atol=1e-3)


def test_standalone_rope_then_attention_equals_fused_rope():
    """apply_rope + plain decode == decode with pos_encoding_mode='ROPE_LLAMA' (the fused form)."""
    import flashinfer

    torch.manual_seed(2)
    L, hq, hkv, d = 300, 8, 2, 128
    q = torch.randn(hq, d).half().to(DEV)
    k = torch.randn(L, hkv, d).half().to(DEV)
    v = torch.randn(L, hkv, d).half().to(DEV)
    o_fused = flashinfer.single_decode_with_kv_cache(q, k, v, pos_encoding_mode="ROPE_LLAMA")
    kpos = torch.arange(L, dtype=torch.int32, device=DEV)
    _, k_rot = flashinfer.apply_rope_pos_ids(k[:, :1].expand(L, 1, d).contiguous(), k, kpos)
    q_rot, _ = flashinfer.apply_rope_pos_ids(q[None], q[None, :1].contiguous(), torch.tensor([L - 1], dtype=torch.int32, device=DEV))
    o_sep = flashinfer.single_decode_with_kv_cache(q_rot[0], k_rot, v)
    # the separate path rounds the rotated q/k to fp16 once more
    torch.testing.assert_close(o_fused.float(), o_sep.float(), rtol=3e-3, atol=3e-3)

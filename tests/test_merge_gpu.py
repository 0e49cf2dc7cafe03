"""GPU parity: cascade merge operators against the oracle (ref: tests/attention/test_shared_prefix_kernels.py:229-303)."""
import pytest
import torch

from oracle import attention_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("seq_len,heads,d", [(1, 1, 64), (77, 32, 128), (512, 8, 256), (0, 4, 128)])
def test_merge_state(dtype, seq_len, heads, d):
    import flashinfer

    torch.manual_seed(0)
    va, vb = torch.randn(seq_len, heads, d).to(dtype), torch.randn(seq_len, heads, d).to(dtype)
    sa, sb = torch.randn(seq_len, heads) * 5, torch.randn(seq_len, heads) * 5
    v, s = flashinfer.merge_state(va.to(DEV), sa.to(DEV), vb.to(DEV), sb.to(DEV))
    v_ref, s_ref = R.merge_state_ref(va.float(), sa, vb.float(), sb)
    tol = 1e-3 if dtype != torch.bfloat16 else 8e-3
    torch.testing.assert_close(v.float().cpu(), v_ref.float(), rtol=tol, atol=tol)
    torch.testing.assert_close(s.cpu(), s_ref.float(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("use_mask", [False, True])
def test_merge_state_in_place(use_mask):
    import flashinfer

    torch.manual_seed(1)
    n, h, d = 129, 8, 128
    v, vo = torch.randn(n, h, d).half(), torch.randn(n, h, d).half()
    s, so = torch.randn(n, h) * 4, torch.randn(n, h) * 4
    mask = (torch.rand(n) > 0.5) if use_mask else None
    vd, sd = v.to(DEV).clone(), s.to(DEV).clone()
    flashinfer.merge_state_in_place(vd, sd, vo.to(DEV), so.to(DEV), mask=None if mask is None else mask.to(DEV))
    v_ref, s_ref = R.merge_state_ref(v.float(), s, vo.float(), so)
    if mask is not None:
        v_ref = torch.where(mask[:, None, None], v_ref, v.double())
        s_ref = torch.where(mask[:, None], s_ref, s.double())
    torch.testing.assert_close(vd.float().cpu(), v_ref.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(sd.cpu(), s_ref.float(), rtol=1e-4, atol=1e-4)
    if mask is not None:  # untouched rows are bit-identical
        assert torch.equal(vd.cpu()[~mask], v[~mask])


@pytest.mark.parametrize("sets", [1, 2, 8, 33])
def test_merge_states(sets):
    import flashinfer

    torch.manual_seed(2)
    n, h, d = 64, 32, 128
    v = torch.randn(n, sets, h, d).half()
    s = torch.randn(n, sets, h) * 6
    vm, sm = flashinfer.merge_states(v.to(DEV), s.to(DEV))
    v_ref, s_ref = R.merge_states_ref(v.float(), s)
    torch.testing.assert_close(vm.float().cpu(), v_ref.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(sm.cpu(), s_ref.float(), rtol=1e-4, atol=1e-4)


def test_merge_with_empty_state_is_identity():
    import flashinfer

    n, h, d = 5, 4, 128
    v = torch.randn(n, h, d).half().to(DEV)
    s = torch.randn(n, h).to(DEV)
    e_v = torch.zeros_like(v)
    e_s = torch.full_like(s, R.NEG_INF_SENTINEL)
    vm, sm = flashinfer.merge_state(v, s, e_v, e_s)
    torch.testing.assert_close(vm, v, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(sm, s, rtol=1e-5, atol=1e-5)

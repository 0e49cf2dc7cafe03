import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flashinfer-ai_amd"))
import torch
from flashinfer import _lib
from oracle import attention_ref as R
DEV = "cuda:0"
g = torch.Generator().manual_seed(5)
counts = [1, 9, 2, 0, 5, 3, 1, 7]
indptr = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32)
nnz, h, d = int(indptr[-1]), 6, 128
for dt in (torch.float16, torch.bfloat16, torch.float32):
    v = torch.randn(nnz, h, d, generator=g).to(dt)
    s = torch.randn(nnz, h, generator=g) * 3
    vo = torch.full((len(counts), h, d), 7.0, dtype=dt, device=DEV)
    so = torch.full((len(counts), h), 7.0, dtype=torch.float32, device=DEV)
    v_d, s_d, ip_d = v.to(DEV), s.to(DEV), indptr.to(DEV)
    _lib.check(_lib.lib().fi_variable_length_merge_states(v_d.data_ptr(), s_d.data_ptr(), ip_d.data_ptr(), vo.data_ptr(),
               so.data_ptr(), len(counts), h, d, _lib.fi_dtype(dt), _lib.fi_dtype(dt), _lib.current_stream(vo.device)), "x")
    torch.cuda.synchronize()
    for r, c in enumerate(counts):
        lo, hi = int(indptr[r]), int(indptr[r + 1])
        v_ref, s_ref = R.merge_states_ref(v[lo:hi].float()[None], s[lo:hi][None])
        got = vo[r].float().cpu()
        err = (got - v_ref[0].float()).abs()
        print(dt, "row", r, "n", c, "nan", int(torch.isnan(got).sum()), "maxerr", float(err[~torch.isnan(err)].max()) if (~torch.isnan(err)).any() else None,
              "per-head nan", [int(torch.isnan(got[hh]).sum()) for hh in range(h)], "s", so[r].cpu().tolist()[:3], s_ref[0][:3].tolist())

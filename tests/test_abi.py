"""CPU: the C-ABI library loads and exports every symbol include/fi_mi355.h declares; host-side argument
validation reports through fi_last_error().  No kernel is launched here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fi_mi355.h")


def declared_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"FI_API\s+[\w\s\*]+?\b(fi_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "fi_batch_decode_plan" in syms and "fi_batch_decode_run" in syms
    assert "fi_last_error" in syms


def test_library_exports_every_declared_symbol(fi_lib):
    from flashinfer import _lib

    missing = [s for s in declared_symbols() if not hasattr(fi_lib, s)]
    assert not missing, f"library does not export: {missing}"
    # and the Python binding knows every one of them
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()


def test_abi_version_and_cu_count(fi_lib):
    assert fi_lib.fi_abi_version() == 2
    assert fi_lib.fi_num_compute_units() > 0


def test_errors_are_reported_not_thrown(fi_lib):
    from flashinfer import _lib

    info = (C.c_int64 * _lib.FI_DECODE_PLAN_INFO_LEN)()
    # null pinned buffer / indptr
    rc = fi_lib.fi_batch_decode_plan(None, 0, None, None, 0, None, 1, 8, 8, 16, 0, 128, 0, 0, 0, -1, info, None)
    assert rc != 0
    assert b"null" in fi_lib.fi_last_error()
    # num_qo_heads not a multiple of num_kv_heads
    buf = (C.c_char * 4096)()
    indptr = (C.c_int32 * 2)(0, 4)
    rc = fi_lib.fi_batch_decode_plan(None, 0, None, buf, 4096, indptr, 1, 7, 2, 16, 0, 128, 0, 0, 0, -1, info, None)
    assert rc != 0 and b"multiple" in fi_lib.fi_last_error()
    # unsupported head_dim
    rc = fi_lib.fi_batch_decode_plan(None, 0, None, buf, 4096, indptr, 1, 8, 2, 16, 0, 96, 0, 0, 0, -1, info, None)
    assert rc != 0 and b"unsupported" in fi_lib.fi_last_error()
    # run with a plan_info that is not a plan
    with pytest.raises(RuntimeError, match="plan"):
        _lib.check(fi_lib.fi_batch_decode_run(None, 0, None, 0, info, _lib.FI_DECODE_PLAN_INFO_LEN, None, None), "run")


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "flashinfer-ai_amd", "flashinfer")
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            text = open(os.path.join(pkg, name)).read()
            assert "oracle" not in text.replace("oracle/", ""), f"{name} references the oracle"


def test_ops_fail_loudly_on_cpu_tensors():
    import torch

    import flashinfer

    q = torch.zeros(4, 64, dtype=torch.float16)
    k = torch.zeros(8, 4, 64, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="GPU"):
        flashinfer.single_decode_with_kv_cache(q, k, k)
    with pytest.raises(RuntimeError, match="GPU"):
        flashinfer.merge_state(torch.zeros(1, 1, 64), torch.zeros(1, 1), torch.zeros(1, 1, 64), torch.zeros(1, 1))

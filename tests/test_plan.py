"""CPU: the host planner behind the C ABI (host-only mode: int_ws = NULL) against the Python
restatement of the reference scheduler (oracle/plan_ref.py)."""
import ctypes as C
import random

import numpy as np
import pytest

from oracle.plan_ref import decode_plan_ref


def run_plan(fi_lib, indptr, hq, hkv, page_size, max_grid, cuda_graph=False, head_dim=128,
             float_bytes=1 << 30, window_left=-1):
    from flashinfer import _lib

    n = len(indptr) - 1
    pinned = (C.c_char * (8 << 20))()
    arr = (C.c_int32 * len(indptr))(*indptr)
    info = (C.c_int64 * _lib.FI_DECODE_PLAN_INFO_LEN)()
    rc = fi_lib.fi_batch_decode_plan(None, float_bytes, None, pinned, len(pinned), arr, n, hq, hkv,
                                     page_size, int(cuda_graph), head_dim, 1, 1, max_grid, window_left, info, None)
    assert rc == 0, fi_lib.fi_last_error()
    info = list(info)
    raw = np.frombuffer(pinned, dtype=np.uint8)

    def i32(off, count):
        return raw[off: off + 4 * count].view(np.int32).tolist()

    padded, nwork = info[0], info[11]
    out = dict(split_kv=bool(info[9]), kv_chunk_size=info[10], padded_batch_size=padded, num_work=nwork,
               request_indices=i32(info[3], nwork), kv_tile_indices=i32(info[4], nwork),
               o_indptr=i32(info[5], n + 1), chunk_ptr=i32(info[7], 1)[0], info=info)
    if out["split_kv"]:
        out["mask"] = raw[info[6]: info[6] + padded].tolist()
    return out


CASES = [
    # (indptr, hq, hkv, page_size, max_grid)
    ([0, 512], 32, 8, 16, 2048),
    ([i * 512 for i in range(65)], 32, 8, 16, 2048),        # BASELINE config C2
    ([i * 512 for i in range(65)], 32, 8, 16, 256),         # batch*heads >= grid: no split
    ([0, 17, 29, 44, 48, 66, 100, 128], 64, 8, 16, 2048),   # ref doc example decode.py:602-610
    ([0, 0, 9], 4, 4, 16, 64),                              # empty request (test_decode_prefill_lse.py)
    ([0, 3, 3, 1000], 8, 1, 1, 512),
    ([0, 5], 28, 4, 8, 4096),                               # group 7 -> matrix-core kernel, one item per kv head
    ([0, 40, 90], 64, 1, 16, 512),                          # group 64 -> two 32-head column blocks
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("cuda_graph", [False, True])
def test_planner_matches_oracle(fi_lib, case, cuda_graph):
    indptr, hq, hkv, ps, grid = case
    got = run_plan(fi_lib, indptr, hq, hkv, ps, grid, cuda_graph)
    exp = decode_plan_ref(indptr, hq, hkv, ps, grid, cuda_graph)
    for key in ("split_kv", "kv_chunk_size", "padded_batch_size", "num_work", "request_indices",
                "kv_tile_indices", "o_indptr"):
        assert got[key] == exp[key], key
    assert got["chunk_ptr"] == exp["kv_chunk_size"]
    if got["split_kv"]:
        assert got["mask"] == [1] * exp["num_work"] + [0] * (exp["padded_batch_size"] - exp["num_work"])


def test_planner_random_page_tables(fi_lib):
    rng = random.Random(0)
    for _ in range(200):
        b = rng.randint(1, 40)
        ps = rng.choice([1, 4, 8, 16, 32])
        pages = [rng.choice([0, 1, 2, rng.randint(1, 600)]) for _ in range(b)]
        indptr = [0]
        for p in pages:
            indptr.append(indptr[-1] + p)
        hkv = rng.choice([1, 2, 4, 8])
        hq = hkv * rng.choice([1, 2, 3, 4, 5, 8, 16])
        grid = rng.choice([64, 512, 2048, 4096])
        cg = rng.random() < 0.3
        got = run_plan(fi_lib, indptr, hq, hkv, ps, grid, cg)
        exp = decode_plan_ref(indptr, hq, hkv, ps, grid, cg)
        for key in ("split_kv", "kv_chunk_size", "padded_batch_size", "num_work", "request_indices",
                    "kv_tile_indices", "o_indptr"):
            assert got[key] == exp[key], (key, indptr, hq, hkv, ps, grid, cg)
        # structural properties: chunks tile every request exactly
        chunk_pages = got["kv_chunk_size"] // ps
        for r in range(b):
            n = got["o_indptr"][r + 1] - got["o_indptr"][r]
            if got["split_kv"]:
                assert n == max(1, -(-max(pages[r], 1) // chunk_pages))
            else:
                assert n == 1


def test_workspace_too_small_is_an_error(fi_lib):
    from flashinfer import _lib

    indptr = (C.c_int32 * 2)(0, 4096)
    pinned = (C.c_char * 4096)()
    info = (C.c_int64 * _lib.FI_DECODE_PLAN_INFO_LEN)()
    rc = fi_lib.fi_batch_decode_plan(None, 1024, None, pinned, 4096, indptr, 1, 32, 8, 16, 0, 128, 1, 1,
                                     2048, -1, info, None)
    assert rc != 0 and b"workspace too small" in fi_lib.fi_last_error()


# ---- prefill planner (split-KV work list) -------------------------------------------------------------
def run_prefill_plan(fi_lib, qo_indptr, kv_lens, hq, hkv, causal=False, cuda_graph=False, fixed=-1, disable=False,
                     page_size=16, float_bytes=1 << 32, window_left=-1):
    from flashinfer import _lib

    n = len(kv_lens)
    pinned = (C.c_char * (8 << 20))()
    qo = (C.c_int32 * (n + 1))(*qo_indptr)
    kvp = [0]
    for k in kv_lens:
        kvp.append(kvp[-1] + -(-k // page_size))
    kvi = (C.c_int32 * (n + 1))(*kvp)
    kvl = (C.c_int32 * max(n, 1))(*kv_lens)
    info = (C.c_int64 * _lib.FI_PREFILL_PLAN_INFO_LEN)()
    rc = fi_lib.fi_batch_prefill_plan(None, float_bytes, None, pinned, len(pinned), qo, kvi, kvl, qo_indptr[-1], n,
                                      hq, hkv, page_size, int(cuda_graph), 128, 128, int(causal), window_left, fixed,
                                      int(disable), info, None)
    assert rc == 0, fi_lib.fi_last_error()
    info = list(info)
    raw = np.frombuffer(pinned, dtype=np.uint8)

    def i32(off, count):
        return raw[off: off + 4 * count].view(np.int32).tolist()

    nwork = info[12]
    out = dict(split_kv=bool(info[14]), kv_chunk_size=info[9], padded_batch_size=info[0], num_work=nwork,
               request_indices=i32(info[4], nwork), qo_tile_indices=i32(info[5], nwork),
               kv_tile_indices=i32(info[6], nwork), padding=i32(info[4], info[0])[nwork:])
    out["merge_indptr"] = i32(info[7], info[1] + 1) if out["split_kv"] else [0]
    return out


PREFILL_CASES = [
    # (qo_indptr, kv_lens, hq, hkv)
    ([0, 2048 * 1], [8192], 32, 8),                                   # one C3 request: 64 q tiles, no split
    ([0, 128], [32768], 32, 8),                                       # append: 4 q tiles -> kv split
    ([0, 5, 205, 206], [3000, 4097, 130], 8, 2),
    ([0, 16, 32, 48, 64], [16384] * 4, 32, 8),
    ([0, 0, 7], [0, 1], 4, 4),                                        # empty request / single key
    ([0, 300], [300], 28, 4),
    # mixed batches of the reference's bench_batch_attention.py: more q tiles than resident workgroups, so the
    # reference rule never splits -- the build's load-balance rule cuts the few long requests
    (list(range(123)) + [122 + 17 * (i + 1) for i in range(8)], [600] * 122 + [10000] * 8, 28, 4),
    (list(range(129)), [8192] * 128, 28, 4),                          # decode-only through the prefill wrapper
    (list(range(21)) + [20 + 1000], [4096] * 20 + [4096], 28, 4),     # chunked prefill: 20 decode rows + one long prompt
]


@pytest.mark.parametrize("case", PREFILL_CASES)
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("mode", ["auto", "graph", "fixed", "disabled"])
def test_prefill_planner_matches_oracle(fi_lib, monkeypatch, case, causal, mode):
    from oracle.plan_ref import prefill_plan_ref

    monkeypatch.setenv("FI_NUM_CUS", "256")
    qo_indptr, kv_lens, hq, hkv = case
    kw = dict(cuda_graph=mode == "graph", fixed=512 if mode == "fixed" else -1, disable=mode == "disabled")
    got = run_prefill_plan(fi_lib, qo_indptr, kv_lens, hq, hkv, causal=causal, **kw)
    exp = prefill_plan_ref(qo_indptr, kv_lens, hq, hkv, causal=causal, enable_cuda_graph=kw["cuda_graph"],
                           fixed_split_size=kw["fixed"], disable_split_kv=kw["disable"])
    for key in ("split_kv", "kv_chunk_size", "num_work", "request_indices", "qo_tile_indices", "kv_tile_indices",
                "padded_batch_size", "merge_indptr"):
        assert got[key] == exp[key], key
    assert all(r == -1 for r in got["padding"])
    # every (request, q tile, kv chunk) exactly once; chunks cover the kv range
    assert len(set(zip(got["request_indices"], got["qo_tile_indices"], got["kv_tile_indices"]))) == got["num_work"]
    if got["split_kv"]:
        assert got["kv_chunk_size"] % 64 == 0 and got["kv_chunk_size"] >= 128


def test_prefill_planner_grows_chunks_to_fit_the_float_workspace(fi_lib, monkeypatch):
    from oracle.plan_ref import prefill_plan_ref

    monkeypatch.setenv("FI_NUM_CUS", "256")
    qo_indptr, kv_lens, hq, hkv = [0, 64], [65536], 32, 8
    roomy = run_prefill_plan(fi_lib, qo_indptr, kv_lens, hq, hkv)
    tight_bytes = roomy["merge_indptr"][-1] * hq * 129 * 4 * 3 // 4   # three quarters of what the roomy plan's states take
    tight = run_prefill_plan(fi_lib, qo_indptr, kv_lens, hq, hkv, float_bytes=tight_bytes)
    exp = prefill_plan_ref(qo_indptr, kv_lens, hq, hkv, float_ws_bytes=tight_bytes)
    assert roomy["split_kv"] and tight["split_kv"] and tight["kv_chunk_size"] > roomy["kv_chunk_size"]
    assert tight["kv_chunk_size"] == exp["kv_chunk_size"] and tight["kv_tile_indices"] == exp["kv_tile_indices"]
    entries = tight["merge_indptr"][-1]
    assert entries * hq * 129 * 4 <= tight_bytes


@pytest.mark.parametrize("causal", [False, True])
def test_prefill_planner_sliding_window_uses_the_effective_kv_span(fi_lib, monkeypatch, causal):
    from oracle.plan_ref import prefill_plan_ref

    monkeypatch.setenv("FI_NUM_CUS", "256")
    qo_indptr, kv_lens, hq, hkv = [0, 64, 128], [40000, 300], 8, 8
    got = run_prefill_plan(fi_lib, qo_indptr, kv_lens, hq, hkv, causal=causal, window_left=1000)
    exp = prefill_plan_ref(qo_indptr, kv_lens, hq, hkv, causal=causal, window_left=1000)
    full = run_prefill_plan(fi_lib, qo_indptr, kv_lens, hq, hkv, causal=causal)
    for key in ("split_kv", "kv_chunk_size", "request_indices", "qo_tile_indices", "kv_tile_indices", "merge_indptr"):
        assert got[key] == exp[key], key
    # the window shortens the span a q tile walks: smaller chunks than the full-kv plan
    assert got["kv_chunk_size"] < full["kv_chunk_size"]


def test_decode_planner_sliding_window_partitions_only_the_window_pages(fi_lib):
    indptr = [0, 4096, 4100, 4101]  # 65536, 64 and 16 tokens at page 16
    got = run_plan(fi_lib, indptr, 32, 8, 16, 2048, window_left=1000)
    exp = decode_plan_ref(indptr, 32, 8, 16, 2048, window_left=1000)
    full = run_plan(fi_lib, indptr, 32, 8, 16, 2048)
    for key in ("split_kv", "kv_chunk_size", "num_work", "request_indices", "kv_tile_indices", "o_indptr"):
        assert got[key] == exp[key], key
    assert got["info"][14] == 1000 and full["info"][14] == -1
    assert got["kv_chunk_size"] < full["kv_chunk_size"]  # 64 window pages, not 4096, are spread over the grid


def test_prefill_planner_balances_a_mixed_batch(fi_lib, monkeypatch):
    """122 one-row requests of 600 keys + 8 requests of 10 000 keys and 17 rows (bench_batch_attention.py's hybrid):
    130 q tiles > 128 resident items, so the reference's binary search leaves every request whole and the launch ends
    in eight lone workgroups walking 10 000 keys.  The balance rule cuts chunks of about W / (2 x 128) tokens."""
    monkeypatch.setenv("FI_NUM_CUS", "256")
    qo_indptr = list(range(123)) + [122 + 17 * (i + 1) for i in range(8)]
    kv_lens = [600] * 122 + [10000] * 8
    got = run_prefill_plan(fi_lib, qo_indptr, kv_lens, 28, 4, causal=True)
    assert got["split_kv"] and got["kv_chunk_size"] == 640          # (122 x 600 + 8 x 10 000) / 256 = 598 -> 640
    assert got["num_work"] == 122 + 8 * 16
    graph = run_prefill_plan(fi_lib, qo_indptr, kv_lens, 28, 4, causal=True, cuda_graph=True)
    assert graph["kv_chunk_size"] >= 10000 and graph["num_work"] == 130   # graph plans keep the reference rule

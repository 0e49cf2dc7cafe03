#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/r3/pgrid2.py 2>&1 | grep TFLOP
timeout -k 10 600 python tools/bench_ref_grids.py mixed 2>&1 | grep -E " 128 +[0-9.]+ +[0-9.]+ +[0-9.]+$"
timeout -k 10 1000 python -m pytest tests/test_prefill_gpu.py tests/test_fuzz_gpu.py tests/test_page_cascade_gpu.py tests/test_graph_replan_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout -k 10 300 python tools/bench_c3.py 1 quick 2>&1 | grep C3

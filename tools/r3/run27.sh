#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_prefill_gpu.py tests/test_fuzz_gpu.py tests/test_page_cascade_gpu.py tests/test_graph_replan_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 600 python tools/bench_ref_grids.py mixed > gpurun_out/r03_mixed_ref_grid_bal.txt 2>&1
grep -v amdgpu gpurun_out/r03_mixed_ref_grid_bal.txt
timeout -k 10 300 python tools/bench_c3.py 1 quick 2>&1 | grep C3

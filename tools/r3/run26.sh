#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_decode_gpu.py tests/test_plan.py tests/test_cascade_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --no-c5 2>&1 | grep '"metric"' | cut -c1-700
timeout -k 10 900 python tools/bench_ref_grids.py decode > gpurun_out/r03_decode_ref_grid_wpc4.txt 2>&1
tail -3 gpurun_out/r03_decode_ref_grid_wpc4.txt

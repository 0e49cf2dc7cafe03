#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "fp8" 2>&1 | tail -2 || exit 1
timeout -k 10 600 python tools/bench_prefill_dims.py 2>&1 | grep -E "fp8|d64"

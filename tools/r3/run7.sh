#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x -s -k "bf16_prefill or c5_per_gpu or fa3_six or multi_item or f32_rotation or test_batch_prefill_matches_oracle or test_single_prefill_variants or head_dim_256 or ragged or pos_encoding" > gpurun_out/r3_run7_tests.log 2>&1
echo rc=$?
grep -v "^\.*$" gpurun_out/r3_run7_tests.log | tail -40

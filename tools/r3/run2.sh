#!/bin/bash
# r3 GPU call 2: knock-outs of the lagged pipeline vs the second structure (timing only)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r3_run2.log; : > $L
for ko in 0 64 2 1 8 16; do
  for lag in 1 0; do
    if [ $ko = 0 ]; then unset FI_MI355_LIB; else export FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_prefill_fp8_inst_$ko.so; fi
    echo "KO=$ko LAG=$lag" >> $L
    FI_PREFILL_FP8_LAG=$lag timeout -k 10 200 python tools/bench_c3.py 2 quick >> $L 2>&1 || { echo "failed KO=$ko LAG=$lag" >> $L; exit 1; }
  done
done
grep -v amdgpu.ids $L

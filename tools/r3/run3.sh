#!/bin/bash
# r3 GPU call 3: packed f32 ops / vector-register logit scale in the fp8 prefill softmax (second structure)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r3_run4.log; : > $L
for rep in 1 2; do
for pk in 0 1 2 3; do
    if [ $pk = 0 ]; then unset FI_MI355_LIB; else export FI_MI355_LIB=$GRAFT_REPO_ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_prefill_fp8_inst_$pk.so; fi
    echo "PK=$pk" >> $L
    FI_PREFILL_FP8_LAG=0 timeout -k 10 200 python tools/bench_c3.py 2 quick >> $L 2>&1 || { echo "failed PK=$pk" >> $L; exit 1; }
done
done
grep -v amdgpu.ids $L

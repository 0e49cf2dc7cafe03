#!/bin/bash
# rocprofv3 passes for the C3-shape bf16 prefill kernel: kernel trace + three PMC groups (separate runs, no tracing domains
# beside --kernel-trace).  Usage (on the GPU box): bash tools/prof_pf_bf16.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_pfbf16_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/prof_prefill_once.py bf16 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -- python3 $R/tools/prof_prefill_once.py bf16 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc2 -- python3 $R/tools/prof_prefill_once.py bf16 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -- python3 $R/tools/prof_prefill_once.py bf16 > $OUT/pmc3.log 2>&1
python3 $R/tools/pmc_summary.py batch_prefill_kernel $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 > $OUT/pmc_summary.txt 2>&1 || true
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \; || true
cat $OUT/pmc_summary.txt
head -5 $OUT/kernel_stats.csv

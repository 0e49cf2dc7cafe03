#!/bin/bash
# r3 rocprofv3 passes (on the GPU box: bash tools/prof_r03.sh): C2 through bench.py itself, C3 and C4 through the
# harness that repeats bench.py's secondary_workloads() launch for launch (same inputs, same L2 flush).  Kernel trace
# and every PMC group in separate runs (no tracing domain beside --kernel-trace).  Output: gpurun_out/prof_r03/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r03
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_trace -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-c5 > $OUT/c2_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c2_fetch -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-c5 --steps 30 --warmup 10 > $OUT/c2_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c2_write -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-c5 --steps 30 --warmup 10 > $OUT/c2_write.log 2>&1
echo c2 done
for W in c3 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_trace -- python3 $R/tools/prof_secondary_once.py $W > $OUT/${W}_trace.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/${W}_pmc1 -- python3 $R/tools/prof_secondary_once.py $W > $OUT/${W}_pmc1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/${W}_pmc2 -- python3 $R/tools/prof_secondary_once.py $W > $OUT/${W}_pmc2.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/${W}_pmc3 -- python3 $R/tools/prof_secondary_once.py $W > $OUT/${W}_pmc3.log 2>&1
  rocprofv3 --pmc TCC_EA0_WRREQ_sum --output-format csv -d $OUT/${W}_pmc4 -- python3 $R/tools/prof_secondary_once.py $W > $OUT/${W}_pmc4.log 2>&1
  echo $W done
done
python3 $R/tools/summarize_r03.py $OUT $R/gpurun_out/prof_r03_summary
ls $R/gpurun_out/prof_r03_summary

#!/bin/bash
# rocprofv3 passes for the C2 headline (bench.py, decode only): kernel trace + FETCH_SIZE and WRITE_SIZE in separate
# --pmc runs.  Usage (on the GPU box): bash tools/prof_c2.sh <tag>; summaries land in gpurun_out/prof_c2_<tag>/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_c2_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-secondary --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --no-secondary --no-cpu-baseline --steps 30 --warmup 10 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --no-secondary --no-cpu-baseline --steps 30 --warmup 10 > $OUT/write.log 2>&1
python3 $R/tools/summarize_prof.py $OUT/trace $OUT/fetch $OUT/write $OUT/r02_c2_decode decode_mfma16_kernel

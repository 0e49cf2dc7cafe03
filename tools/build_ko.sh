#!/bin/bash
# Knock-out builds of one translation unit (timing experiments; results of these libraries are wrong).
#   tools/build_ko.sh <unit> <macro> <value> [<value> ...]     e.g.  tools/build_ko.sh prefill_fp8_inst FI_PF8_KO 1 2 4
# Each value gives flashinfer-ai_amd/flashinfer/ko/libfi_<unit>_<value>.so (git-ignored, travels with gpurun);
# select it with FI_MI355_LIB=<path>.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
UNIT=$1; MACRO=$2; shift 2
make -s -j8 -C $ROOT/flashinfer-ai_amd/csrc >/dev/null
mkdir -p $ROOT/flashinfer-ai_amd/flashinfer/ko $ROOT/build/ko
OBJS=$(ls $ROOT/build/csrc/*.o | grep -v "/$UNIT.o")
EXTRA=""
[ "$UNIT" = "prefill_fp8_inst" ] && EXTRA="-fno-slp-vectorize"
for V in "$@"; do
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$ROOT/flashinfer-ai_amd/csrc -Wno-unused-function \
      -fno-gpu-rdc -fvisibility=hidden -DFI_BUILDING_LIB $EXTRA -D$MACRO=$V -c $ROOT/flashinfer-ai_amd/csrc/$UNIT.hip -o $ROOT/build/ko/${UNIT}_$V.o && \
    hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/flashinfer-ai_amd/flashinfer/ko/libfi_${UNIT}_$V.so $OBJS $ROOT/build/ko/${UNIT}_$V.o ) &
done
wait
ls -la $ROOT/flashinfer-ai_amd/flashinfer/ko/

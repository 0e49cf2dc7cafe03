// One translation unit per (KV dtype, head_dim): compiled several times by the Makefile with
// -DFI_INST_KV_DT=<fi_dtype> -DFI_INST_HEAD_DIM=<n>.  Exposes one launcher that selects the
// (q-head tile, rope) instantiation at run time.
#include "decode_kernel.h"

#ifndef FI_INST_KV_DT
#error "FI_INST_KV_DT / FI_INST_HEAD_DIM must be defined"
#endif

#define FI_CAT_(a, b, c) a##b##_##c
#define FI_CAT(a, b, c) FI_CAT_(a, b, c)
#define FI_LAUNCHER FI_CAT(decode_launch_, FI_INST_KV_DT, FI_INST_HEAD_DIM)

namespace fi {

// loads per register tile: 4 x 16 B of 16-bit KV (16 tokens at head_dim 128); fp8 rows carry twice the
// elements per load (q/o registers double), so 2 loads per tile keep the kernel below 256 VGPRs unspilled
constexpr int kNLoad = (FI_INST_KV_DT == FI_DTYPE_FP8_E4M3 || FI_INST_KV_DT == FI_DTYPE_FP8_E5M2) ? 2 : 4;

// Groups wider than the head tile run as several tiles on neighbouring waves that stream the same K/V
// rows: those launches use temporal loads (the partner hits in L2).  Only the widest tile can be "multi".
constexpr int kMaxTile = 4;

template <int GT, bool ROPE>
static hipError_t launch(const DecodeKernelParams& p, int grid, hipStream_t stream) {
  if constexpr (GT == kMaxTile) {
    if (p.head_tiles > 1) {
      if (p.fast_path)
        batch_decode_kernel<FI_INST_KV_DT, FI_INST_HEAD_DIM, GT, ROPE, true, kNLoad, false>
            <<<dim3(grid), dim3(kDecodeThreads), 0, stream>>>(p);
      else
        batch_decode_kernel<FI_INST_KV_DT, FI_INST_HEAD_DIM, GT, ROPE, false, kNLoad, false>
            <<<dim3(grid), dim3(kDecodeThreads), 0, stream>>>(p);
      return hipGetLastError();
    }
  }
  if (p.fast_path)
    batch_decode_kernel<FI_INST_KV_DT, FI_INST_HEAD_DIM, GT, ROPE, true, kNLoad>
        <<<dim3(grid), dim3(kDecodeThreads), 0, stream>>>(p);
  else
    batch_decode_kernel<FI_INST_KV_DT, FI_INST_HEAD_DIM, GT, ROPE, false, kNLoad>
        <<<dim3(grid), dim3(kDecodeThreads), 0, stream>>>(p);
  return hipGetLastError();
}

hipError_t FI_LAUNCHER(const DecodeKernelParams& p, int gt, int rope, int grid,
                       hipStream_t stream) {
#define FI_CASE(G)                                         \
  case G:                                                  \
    return rope ? launch<G, true>(p, grid, stream) : launch<G, false>(p, grid, stream);
  switch (gt) {
    FI_CASE(1)
    FI_CASE(2)
    FI_CASE(4)
    default:
      return hipErrorInvalidValue;
  }
#undef FI_CASE
}

}  // namespace fi

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

constexpr int kNLoad = 4;

template <int GT, bool ROPE>
static hipError_t launch(const DecodeKernelParams& p, int grid, hipStream_t stream) {
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

// Matrix-core decode instantiations: q dtype (f16 | bf16) x cache dtype (same | e4m3 | e5m2) x head_dim
// (64 | 128) x (paged | identity pages).
#include "decode_mfma_kernel.h"

namespace fi {

template <int T16, int KVS, int D>
static hipError_t launch(const DecodeKernelParams& p, int rope, int grid, hipStream_t stream) {
  if (rope) {
    if (p.indices)
      decode_mfma_kernel<T16, KVS, D, true, true><<<dim3(grid), dim3(kDecodeThreads), 0, stream>>>(p);
    else
      decode_mfma_kernel<T16, KVS, D, false, true><<<dim3(grid), dim3(kDecodeThreads), 0, stream>>>(p);
    return hipGetLastError();
  }
  if (p.indices)
    decode_mfma_kernel<T16, KVS, D, true><<<dim3(grid), dim3(kDecodeThreads), 0, stream>>>(p);
  else
    decode_mfma_kernel<T16, KVS, D, false><<<dim3(grid), dim3(kDecodeThreads), 0, stream>>>(p);
  return hipGetLastError();
}

hipError_t decode_mfma_launch(const DecodeKernelParams& p, int q_dtype, int kv_dtype, int head_dim, int rope, int grid,
                              hipStream_t stream) {
#define FI_CASE(T, K, D) \
  if (q_dtype == T && kv_dtype == K && head_dim == D) return launch<T, K, D>(p, rope, grid, stream);
#define FI_ROW(T, K) FI_CASE(T, K, 64) FI_CASE(T, K, 128)
  FI_ROW(FI_DTYPE_F16, FI_DTYPE_F16)
  FI_ROW(FI_DTYPE_F16, FI_DTYPE_FP8_E4M3)
  FI_ROW(FI_DTYPE_F16, FI_DTYPE_FP8_E5M2)
  FI_ROW(FI_DTYPE_BF16, FI_DTYPE_BF16)
  FI_ROW(FI_DTYPE_BF16, FI_DTYPE_FP8_E4M3)
  FI_ROW(FI_DTYPE_BF16, FI_DTYPE_FP8_E5M2)
#undef FI_ROW
#undef FI_CASE
  return hipErrorInvalidValue;
}

}  // namespace fi

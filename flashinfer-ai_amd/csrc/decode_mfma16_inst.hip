// 16x16x32 matrix-core decode instantiations (groups of <= 16 query heads): q dtype (f16 | bf16) x cache dtype
// (same | e4m3 | e5m2) x head_dim (64 | 128) x (paged | identity pages) x (plain | fused RoPE).
#include "decode_mfma16_kernel.h"

namespace fi {

template <int T16, int KVS, int D>
static hipError_t launch16(const DecodeKernelParams& p, int rope, int grid, hipStream_t stream) {
  const dim3 g(grid), b(kDecodeThreads);
  if (rope) {
    if (p.indices) decode_mfma16_kernel<T16, KVS, D, true, true><<<g, b, 0, stream>>>(p);
    else decode_mfma16_kernel<T16, KVS, D, false, true><<<g, b, 0, stream>>>(p);
  } else {
    if (p.indices) decode_mfma16_kernel<T16, KVS, D, true, false><<<g, b, 0, stream>>>(p);
    else decode_mfma16_kernel<T16, KVS, D, false, false><<<g, b, 0, stream>>>(p);
  }
  return hipGetLastError();
}

hipError_t decode_mfma16_launch(const DecodeKernelParams& p, int q_dtype, int kv_dtype, int head_dim, int rope,
                                int grid, hipStream_t stream) {
#define FI_CASE(T, K, D) \
  if (q_dtype == T && kv_dtype == K && head_dim == D) return launch16<T, K, D>(p, rope, grid, stream);
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

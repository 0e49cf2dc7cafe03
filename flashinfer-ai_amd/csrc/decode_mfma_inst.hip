// Matrix-core decode instantiations: (f16 | bf16) x head_dim (64 | 128); K/V stored in the q dtype.
#include "decode_mfma_kernel.h"

namespace fi {

hipError_t decode_mfma_launch(const DecodeKernelParams& p, int dtype, int head_dim, int grid,
                              hipStream_t stream) {
#define FI_CASE(T, D)                                                                          \
  if (dtype == T && head_dim == D) {                                                           \
    decode_mfma_kernel<T, D><<<dim3(grid), dim3(kDecodeThreads), 0, stream>>>(p);              \
    return hipGetLastError();                                                                  \
  }
  FI_CASE(FI_DTYPE_F16, 64)
  FI_CASE(FI_DTYPE_F16, 128)
  FI_CASE(FI_DTYPE_BF16, 64)
  FI_CASE(FI_DTYPE_BF16, 128)
#undef FI_CASE
  return hipErrorInvalidValue;
}

}  // namespace fi

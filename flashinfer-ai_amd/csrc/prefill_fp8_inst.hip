// fp8-native prefill (prefill_fp8_kernel.h): launchers for the two output types.
#include "prefill_fp8_kernel.h"

namespace fi {

hipError_t prefill_fp8_launch(const PrefillKernelParams& p, int out_dtype, hipStream_t stream) {
  const int grid = p.num_work * p.num_kv_heads;
  if (grid == 0) return hipSuccess;
  if (out_dtype == FI_DTYPE_BF16)
    batch_prefill_fp8_kernel<FI_DTYPE_BF16><<<dim3(grid), dim3(kPrefillThreads), 0, stream>>>(p);
  else
    batch_prefill_fp8_kernel<FI_DTYPE_F16><<<dim3(grid), dim3(kPrefillThreads), 0, stream>>>(p);
  return hipGetLastError();
}

}  // namespace fi

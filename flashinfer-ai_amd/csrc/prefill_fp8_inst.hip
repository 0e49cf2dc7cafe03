// fp8-native prefill (prefill_fp8_kernel.h): launchers for the two output types.
#include <stdlib.h>

#include "prefill_fp8_kernel.h"

namespace fi {

template <int OUT16, bool BF8>
static hipError_t launch_v2(const PrefillKernelParams& p, int head_dim, int grid, hipStream_t stream) {
  // group sizes 1 / 2 / 4: one query head per wave, logit scale in a scalar register (see the kernel)
  const bool uni = p.group_size == 1 || p.group_size == 2 || p.group_size == 4;
  if (head_dim == 256) {  // one workgroup per CU, plain step (prefill_fp8_kernel.h)
    if (uni) batch_prefill_fp8_kernel<OUT16, true, 4, BF8, 256><<<dim3(grid), dim3(kPrefillThreads), 0, stream>>>(p);
    else batch_prefill_fp8_kernel<OUT16, false, 4, BF8, 256><<<dim3(grid), dim3(kPrefillThreads), 0, stream>>>(p);
  } else if (head_dim == 64) {
    if (uni) batch_prefill_fp8_kernel<OUT16, true, 4, BF8, 64><<<dim3(grid), dim3(kPrefillThreads), 0, stream>>>(p);
    else batch_prefill_fp8_kernel<OUT16, false, 4, BF8, 64><<<dim3(grid), dim3(kPrefillThreads), 0, stream>>>(p);
  } else if (p.tile_q == 2 * kTileQ) {  // plan cut for 256-row q tiles: the 8-wave form
    if (uni) batch_prefill_fp8_kernel<OUT16, true, 8, BF8><<<dim3(grid), dim3(512), 0, stream>>>(p);
    else batch_prefill_fp8_kernel<OUT16, false, 8, BF8><<<dim3(grid), dim3(512), 0, stream>>>(p);
  } else {
    if (uni) batch_prefill_fp8_kernel<OUT16, true, 4, BF8><<<dim3(grid), dim3(kPrefillThreads), 0, stream>>>(p);
    else batch_prefill_fp8_kernel<OUT16, false, 4, BF8><<<dim3(grid), dim3(kPrefillThreads), 0, stream>>>(p);
  }
  return hipGetLastError();
}

hipError_t prefill_fp8_launch(const PrefillKernelParams& p, int out_dtype, int e5m2, int head_dim, hipStream_t stream) {
  const int grid = p.num_work * p.num_kv_heads;
  if (grid == 0) return hipSuccess;
  if (e5m2)
    return out_dtype == FI_DTYPE_BF16 ? launch_v2<FI_DTYPE_BF16, true>(p, head_dim, grid, stream)
                                      : launch_v2<FI_DTYPE_F16, true>(p, head_dim, grid, stream);
  return out_dtype == FI_DTYPE_BF16 ? launch_v2<FI_DTYPE_BF16, false>(p, head_dim, grid, stream)
                                    : launch_v2<FI_DTYPE_F16, false>(p, head_dim, grid, stream);
}

}  // namespace fi

// One translation unit per (compute type, kv storage, q storage, head_dim); compiled by the Makefile with
// -DFI_PF_T16=.. -DFI_PF_KVS=.. -DFI_PF_QS=.. -DFI_PF_D=..
#include <cstdlib>

#include "prefill_kernel.h"

#define FI_CAT5_(a, b, c, d, e) a##b##_##c##_##d##_##e
#define FI_CAT5(a, b, c, d, e) FI_CAT5_(a, b, c, d, e)
#define FI_LAUNCHER FI_CAT5(prefill_launch_, FI_PF_T16, FI_PF_KVS, FI_PF_QS, FI_PF_D)

namespace fi {

template <bool ROPE, bool GENERAL, bool SPLIT_P = false>
static hipError_t launch(const PrefillKernelParams& p, hipStream_t stream) {
  auto kern = batch_prefill_kernel<FI_PF_T16, FI_PF_KVS, FI_PF_QS, FI_PF_D, ROPE, GENERAL, SPLIT_P>;
  constexpr int smem = 2 * 2 * kTileKV * FI_PF_D * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int grid = p.num_work * p.num_kv_heads;
  if (grid == 0) return hipSuccess;
  kern<<<dim3(grid), dim3(kPrefillThreads), smem, stream>>>(p);
  return hipGetLastError();
}

hipError_t FI_LAUNCHER(const PrefillKernelParams& p, int rope, hipStream_t stream) {
  const bool general = p.use_alibi || p.logits_soft_cap > 0.f || p.custom_mask != nullptr || p.prefix_len_ptr != nullptr;
#if FI_PF_T16 == 1 && FI_PF_QS == 1  // FI_DTYPE_BF16 (an enumerator: not visible to the preprocessor)
  static_assert(FI_DTYPE_BF16 == 1, "bf16 tag");
  // bf16: P as hi + lo halves unless FI_PREFILL_BF16_SINGLE_P=1 asks for the reference's single rounding
  // (prefill.cuh:1263-1275 rounds P once; ~8 % faster, absolute error up to ~4e-3 on unit-variance V)
  static const bool single = [] {
    const char* e = getenv("FI_PREFILL_BF16_SINGLE_P");
    return e && atoi(e) != 0;
  }();
  if (!single) {
    if (rope) return general ? launch<true, true, true>(p, stream) : launch<true, false, true>(p, stream);
    return general ? launch<false, true, true>(p, stream) : launch<false, false, true>(p, stream);
  }
#endif
  if (rope) return general ? launch<true, true>(p, stream) : launch<true, false>(p, stream);
  return general ? launch<false, true>(p, stream) : launch<false, false>(p, stream);
}

}  // namespace fi

// One translation unit per (compute type, kv storage, q storage, head_dim); compiled by the Makefile with
// -DFI_PF_T16=.. -DFI_PF_KVS=.. -DFI_PF_QS=.. -DFI_PF_D=..
#include <cstdlib>

#include "prefill_kernel.h"

#define FI_CAT5_(a, b, c, d, e) a##b##_##c##_##d##_##e
#define FI_CAT5(a, b, c, d, e) FI_CAT5_(a, b, c, d, e)
#define FI_LAUNCHER FI_CAT5(prefill_launch_, FI_PF_T16, FI_PF_KVS, FI_PF_QS, FI_PF_D)

namespace fi {

template <bool ROPE, int GEN, int PMODE = 0>
static hipError_t launch(const PrefillKernelParams& p, hipStream_t stream) {
  auto kern = batch_prefill_kernel<FI_PF_T16, FI_PF_KVS, FI_PF_QS, FI_PF_D, ROPE, GEN, PMODE>;
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

// feature mask of a run (prefill_kernel.h, GEN): a single feature without fused RoPE gets its own instantiation,
// combinations (and every fused-RoPE run with a feature) the all-features one
template <int PMODE>
static hipError_t launch_features(const PrefillKernelParams& p, int rope, hipStream_t stream) {
  const int gen = (p.use_alibi ? 1 : 0) | (p.logits_soft_cap > 0.f ? 2 : 0) | (p.custom_mask != nullptr ? 4 : 0) |
                  (p.prefix_len_ptr != nullptr ? 8 : 0);
  if (rope) return gen ? launch<true, 15, PMODE>(p, stream) : launch<true, 0, PMODE>(p, stream);
  switch (gen) {
    case 0: return launch<false, 0, PMODE>(p, stream);
    case 1: return launch<false, 1, PMODE>(p, stream);
    case 2: return launch<false, 2, PMODE>(p, stream);
    case 4: return launch<false, 4, PMODE>(p, stream);
    case 8: return launch<false, 8, PMODE>(p, stream);
    default: return launch<false, 15, PMODE>(p, stream);
  }
}

hipError_t FI_LAUNCHER(const PrefillKernelParams& p, int rope, hipStream_t stream) {
#if FI_PF_T16 == 1 && FI_PF_QS == 1  // FI_DTYPE_BF16 (an enumerator: not visible to the preprocessor)
  static_assert(FI_DTYPE_BF16 == 1, "bf16 tag");
  // bf16: P.V on the f16 MFMA (prefill_kernel.h, PMODE 2) unless FI_PREFILL_BF16_P selects 0 = the reference's single
  // bf16 rounding of P (prefill.cuh:962-985; absolute error up to ~4e-3 on unit-variance V) or 1 = hi + lo bf16 halves
  // (no f16 range limit on V; 25 % slower).  FI_PREFILL_BF16_SINGLE_P=1 is the older spelling of 0.
  static const int pmode = [] {
    if (const char* e = getenv("FI_PREFILL_BF16_P")) return atoi(e);
    const char* s1 = getenv("FI_PREFILL_BF16_SINGLE_P");
    return (s1 && atoi(s1) != 0) ? 0 : 2;
  }();
  // the caller's choice (fi_batch_prefill_params_t.bf16_pv_mode: 1 hi + lo, 2 f16 P.V, 3 single rounding) wins over
  // the process-wide default
  const int mode = p.bf16_pv_mode == 1 ? 1 : p.bf16_pv_mode == 2 ? 2 : p.bf16_pv_mode == 3 ? 0 : pmode;
  if (mode == 2) return launch_features<2>(p, rope, stream);
  if (mode == 1) return launch_features<1>(p, rope, stream);
#endif
  return launch_features<0>(p, rope, stream);
}

}  // namespace fi

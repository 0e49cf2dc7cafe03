// Shared by the fp8 groupwise GEMM translation units (gemm.hip, gemm_big.hip): launch parameters and the
// device-side (group, m tile) search.
#pragma once
#include "common.h"

namespace fi {

constexpr int kGemmThreads = 256;
constexpr int kBM = 128, kBN = 128, kBK = 128;

struct GemmParams {
  const uint8_t* a;
  const uint8_t* b;
  const float* a_scale;
  const float* b_scale;
  void* d;
  const int32_t* m_indptr;  // [G+1] device; NULL: one group of m_total rows
  int32_t num_groups, m_total, n, k;
  int32_t a_gran_m;         // 1 or 128
  int32_t scale_k_major;    // 0: "MN" major, 1: "K" major
  int32_t out_dtype;
  int32_t num_m_tiles_bound;  // grid bound on (group, m tile) pairs
  int32_t num_m_tiles_bound_ws;  // the same for the 256-row tiles of the producer/consumer kernel
  int32_t n_tiles;
  int32_t a_is_e5m2, b_is_e5m2;
  // 256 x 256 kernel only: device word written by scales_pow2_check_kernel before the launch -- 0: every a / b scale
  // is an exact power of two (what the reference's quantiser produces, flashinfer/testing/utils.py:96-98), so the
  // scales can ride the MFMA's hardware block scales; non-zero (or a null pointer): the general fold path
  const uint32_t* pow2_flag;
};

using f32x16g = __attribute__((ext_vector_type(16))) float;

// (group, m tile) of the mt_global-th m tile from the running count of tiles per group, on the device (the
// reference's arg-prep kernel group_gemm_fp8_groupwise_sm100.cuh:35-72 also sizes the groups on the device, no
// host sync).  Wave-parallel: 64 groups per pass -- lane i loads m_indptr[i], [i + 1], an inclusive scan over
// the lanes gives each group's first tile -- so that 256 experts cost 4 passes, not 256 dependent loads.
// Every lane of every wave runs it with the same arguments and gets the same (uniform) answer.
template <int TILE_M>
__device__ __forceinline__ bool find_group_tile(const int32_t* m_indptr, int num_groups, int mt_global, int lane,
                                                int& g, int& m_begin, int& m_end, int& mt) {
  int first = 0;  // tiles in the groups of earlier passes
  for (int base = 0; base < num_groups; base += 64) {
    const int gi = base + lane;
    const bool in = gi < num_groups;
    const int lo = in ? m_indptr[gi] : 0, hi = in ? m_indptr[gi + 1] : 0;
    const int tiles = (hi - lo + TILE_M - 1) / TILE_M;
    int incl = tiles;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    const int start = first + incl - tiles;
    const uint64_t hit = __ballot(in && mt_global >= start && mt_global < start + tiles);
    if (hit) {
      const int src = __builtin_ctzll(hit);
      g = base + src;
      m_begin = __builtin_amdgcn_readlane(lo, src);
      m_end = __builtin_amdgcn_readlane(hi, src);
      mt = mt_global - __builtin_amdgcn_readlane(start, src);
      return true;
    }
    first += __builtin_amdgcn_readlane(incl, 63);
  }
  return false;
}

using i32x8g = __attribute__((ext_vector_type(8))) int;

// 256 x 256 tile persistent kernel (gemm_big.hip); p.num_m_tiles_bound counts 256-row tiles
// The power-of-two verdict of a call: kPow2Words words, one per workgroup of the check kernel (non-zero = that
// workgroup saw a scale that is not a positive normal power of two).  No word is ever reset: every call's check
// kernel rewrites all of its slot's words, so there is no memset node in front of it.
constexpr int kPow2Words = 64;
__device__ __forceinline__ bool fi_scales_are_pow2(const uint32_t* flag) {
  return flag != nullptr && !__any(flag[threadIdx.x & (kPow2Words - 1)] != 0);
}

// hws_only_flag == nullptr: both variants (power-of-two scales on the hardware path, anything else folded).
// Otherwise only the hardware-scale variant, and *hws_only_flag receives the call's device flag words (fi_scales_are_pow2 after the check
// kernel = that variant does the call; the caller's own kernel must return at once then), or nullptr when nothing
// was launched.
hipError_t launch_gemm_big(const GemmParams& p, int grid, hipStream_t stream, uint32_t** hws_only_flag);

}  // namespace fi

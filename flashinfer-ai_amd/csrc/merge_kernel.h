// Attention-state merge kernels (cascade).  HBM-bound: every state is read once, one wave per
// (row, head); lanes stride over head_dim so each load instruction is a contiguous 128/256-byte row
// segment.  Operator (ref: include/flashinfer/attention/cascade.cuh:44-71, state.cuh:52-63):
//   (v, s) (+) (v', s'):  m = max(s, s');  w = 2^(s-m), w' = 2^(s'-m)
//   v <- (w v + w' v') / (w + w'),  s <- m + log2(w + w')
#pragma once
#include "common.h"

namespace fi {

constexpr int kMergeThreads = 256;
constexpr int kMergeWaves = kMergeThreads / 64;
constexpr int kMergeMaxPerLane = 8;  // head_dim <= 512

struct MergeNParams {
  const void* v;         // states
  const float* s;
  const int32_t* indptr;  // ragged: row r owns entries indptr[r]..indptr[r+1]; NULL: fixed n
  void* v_out;
  float* s_out;
  int32_t n_fixed;  // entries per row when indptr == NULL; layout [row, n, H, D]
  int32_t seq_len, num_heads, head_dim;
  int32_t in_dtype, out_dtype;
  int32_t skip_empty;  // rows without entries are left untouched (padding rows of a fixed-shape launch)
};

struct Merge2Params {
  const void* v_a;
  const float* s_a;
  const void* v_b;
  const float* s_b;
  void* v_out;
  float* s_out;
  const uint8_t* mask;  // optional per-row; 0 = keep (v_a, s_a)
  int32_t seq_len, num_heads, head_dim, dtype;
};

hipError_t launch_merge_n(const MergeNParams& p, hipStream_t stream);
hipError_t launch_merge_2(const Merge2Params& p, hipStream_t stream);

}  // namespace fi

// Cascade merge kernels + their C-ABI entry points.
#include "merge_kernel.h"

namespace fi {

// Merge n states per (row, head).  Ragged layout [nnz, H, D] (ref: VariableLengthMergeStates,
// cascade.cuh:366-467) or dense [row, n, H, D] (ref: MergeStates, cascade.cuh:213-256).
// One WORKGROUP (4 waves) per (row, head).  Two passes so that no load depends on a previous one:
// (1) every wave reads the n log-sum-exp values 64 at a time, one per lane, and reduces them to the row
// maximum; (2) the value rows are split over the four waves (entry j -> wave j % 4), streamed with the
// weight of entry j broadcast from its lane, and the four partial sums are combined through LDS.  Long
// context / small batch decode merges hundreds of split-KV partials per head, which a single wave would
// read as one latency-bound chain.
__global__ void __launch_bounds__(kMergeThreads) merge_n_kernel(const MergeNParams p) {
  __shared__ float red[kMergeWaves][kMergeMaxPerLane + 1][64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int64_t item = blockIdx.x;
  const int row = (int)(item / p.num_heads);
  const int head = (int)(item % p.num_heads);
  int64_t first;
  int n;
  if (p.indptr) {
    first = p.indptr[row];
    n = p.indptr[row + 1] - (int)first;
  } else {
    first = (int64_t)row * p.n_fixed;
    n = p.n_fixed;
  }
  const int D = p.head_dim;
  // pass 1: maximum (every wave computes it: n floats, L2-resident)
  float mx = -1.0e30f;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    if (j < n) mx = fmaxf(mx, p.s[(first + j) * p.num_heads + head]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float acc[kMergeMaxPerLane];
#pragma unroll
  for (int k = 0; k < kMergeMaxPerLane; ++k) acc[k] = 0.f;
  float dsum = 0.f;
  // pass 2: this wave owns entries j with j % 4 == wave inside each group of 64
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    const float w_lane = j < n ? fast_exp2(p.s[(first + j) * p.num_heads + head] - mx) : 0.f;
    if (wave == 0) dsum += w_lane;
    const int cnt = min(64, n - j0);
#pragma unroll 4
    for (int jj = wave; jj < cnt; jj += kMergeWaves) {
      const float w = __shfl(w_lane, jj, 64);
      const int64_t e = (first + j0 + jj) * p.num_heads + head;
#pragma unroll
      for (int k = 0; k < kMergeMaxPerLane; ++k) {
        const int i = lane + 64 * k;
        if (i < D) acc[k] += w * load_any_float(p.v, e * D + i, p.in_dtype);
      }
    }
  }
  if (n > 1) {
#pragma unroll
    for (int k = 0; k < kMergeMaxPerLane; ++k) red[wave][k][lane] = acc[k];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int k = 0; k < kMergeMaxPerLane; ++k)
      acc[k] = red[0][k][lane] + red[1][k][lane] + red[2][k][lane] + red[3][k][lane];
  } else if (wave != 0) {
    return;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off, 64);
  const int64_t ob = ((int64_t)row * p.num_heads + head) * D;
  // n == 0, or nothing but empty partial states -> zeros / -inf sentinel (ref: cascade.cuh:397-405)
  if (n == 0 && p.skip_empty) return;
  const bool empty = !(dsum > 0.f) || mx <= FI_NEG_INF;
  const float inv = empty ? 0.f : 1.0f / dsum;
#pragma unroll
  for (int k = 0; k < kMergeMaxPerLane; ++k) {
    const int i = lane + 64 * k;
    if (i < D) store_any_float(p.v_out, ob + i, acc[k] * inv, p.out_dtype);
  }
  if (lane == 0 && p.s_out)
    p.s_out[(int64_t)row * p.num_heads + head] = empty ? FI_NEG_INF : mx + fast_log2(dsum);
}

// The same merge for f32 partial states with head_dim = 64 * VEC (the split-KV partials of decode and
// prefill): every lane owns VEC CONTIGUOUS elements, so an entry is one 4*VEC-byte load per lane, and each
// wave keeps kMergeInFlight entries in flight.  Long-context / small-batch decode merges hundreds of
// partials per (row, head) on few workgroups: with one dependent 4-byte load pair per entry that merge
// took longer than the attention kernel itself (bs 8 x 32k tokens, 8 q heads / 1 kv head: 51 us vs 28 us).
constexpr int kMergeInFlight = 8;
template <int VEC>
__global__ void __launch_bounds__(kMergeThreads) merge_n_f32_kernel(const MergeNParams p) {
  __shared__ float red[kMergeWaves][VEC][64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int64_t item = blockIdx.x;
  const int row = (int)(item / p.num_heads);
  const int head = (int)(item % p.num_heads);
  int64_t first;
  int n;
  if (p.indptr) {
    first = p.indptr[row];
    n = p.indptr[row + 1] - (int)first;
  } else {
    first = (int64_t)row * p.n_fixed;
    n = p.n_fixed;
  }
  if (n == 0 && p.skip_empty) return;
  constexpr int D = 64 * VEC;
  using vec_t = __attribute__((ext_vector_type(VEC))) float;
  const float* const v = (const float*)p.v;
  float mx = -1.0e30f;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    if (j < n) mx = fmaxf(mx, p.s[(first + j) * p.num_heads + head]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  float dsum = 0.f;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    const float w_lane = j < n ? fast_exp2(p.s[(first + j) * p.num_heads + head] - mx) : 0.f;
    if (wave == 0) dsum += w_lane;
    const int cnt = min(64, n - j0);
    // this wave owns entries jj = wave, wave + 4, ... of the group; kMergeInFlight of them per round
    for (int jj0 = wave; jj0 < cnt; jj0 += kMergeWaves * kMergeInFlight) {
      float vals[kMergeInFlight][VEC], ws[kMergeInFlight];
#pragma unroll
      for (int u = 0; u < kMergeInFlight; ++u) {
        const int jj = jj0 + kMergeWaves * u;
        const int jc = min(jj, cnt - 1);
        const int64_t e = (first + j0 + jc) * p.num_heads + head;
        if constexpr (VEC == 1) {
          vals[u][0] = v[e * D + lane];
        } else {
          const vec_t t = *(const vec_t*)(v + e * D + lane * VEC);
#pragma unroll
          for (int i = 0; i < VEC; ++i) vals[u][i] = t[i];
        }
        const float w = __shfl(w_lane, jc, 64);
        ws[u] = jj < cnt ? w : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kMergeInFlight; ++u)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = __builtin_fmaf(ws[u], vals[u][i], acc[i]);
    }
  }
  if (n > 1) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = red[0][i][lane] + red[1][i][lane] + red[2][i][lane] + red[3][i][lane];
  } else if (wave != 0) {
    return;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off, 64);
  const int64_t ob = ((int64_t)row * p.num_heads + head) * D + lane * VEC;
  const bool empty = !(dsum > 0.f) || mx <= FI_NEG_INF;
  const float inv = empty ? 0.f : 1.0f / dsum;
#pragma unroll
  for (int i = 0; i < VEC; ++i) store_any_float(p.v_out, ob + i, acc[i] * inv, p.out_dtype);
  if (lane == 0 && p.s_out)
    p.s_out[(int64_t)row * p.num_heads + head] = empty ? FI_NEG_INF : mx + fast_log2(dsum);
}

// ref: MergeStateKernel cascade.cuh:44-71 and MergeStateInPlaceKernel cascade.cuh:86-116
// (in place: v_out == v_a, s_out == s_a).
__global__ void __launch_bounds__(kMergeThreads) merge_2_kernel(const Merge2Params p) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * kMergeWaves + wave;
  if (item >= (int64_t)p.seq_len * p.num_heads) return;
  const int row = (int)(item / p.num_heads);
  if (p.mask && !p.mask[row]) return;
  const float sa = p.s_a[item], sb = p.s_b[item];
  const float mx = fmaxf(sa, sb);
  const float wa = fast_exp2(sa - mx), wb = fast_exp2(sb - mx);
  const float inv = 1.0f / (wa + wb);
  const float a_scale = wa * inv, b_scale = wb * inv;
  const int D = p.head_dim;
  const int64_t base = item * D;
  for (int i = lane; i < D; i += 64) {
    float va = load_any_float(p.v_a, base + i, p.dtype);
    float vb = load_any_float(p.v_b, base + i, p.dtype);
    store_any_float(p.v_out, base + i, a_scale * va + b_scale * vb, p.dtype);
  }
  if (lane == 0 && p.s_out) p.s_out[item] = fast_log2(wa + wb) + mx;
}


hipError_t launch_merge_n(const MergeNParams& p, hipStream_t stream) {
  const int64_t items = (int64_t)p.seq_len * p.num_heads;
  if (items == 0) return hipSuccess;
  const int grid = (int)items;  // one workgroup per (row, head)
  if (p.in_dtype == FI_DTYPE_F32 && ((uintptr_t)p.v % 16) == 0) {
    switch (p.head_dim) {
      case 64: merge_n_f32_kernel<1><<<dim3(grid), dim3(kMergeThreads), 0, stream>>>(p); return hipGetLastError();
      case 128: merge_n_f32_kernel<2><<<dim3(grid), dim3(kMergeThreads), 0, stream>>>(p); return hipGetLastError();
      case 256: merge_n_f32_kernel<4><<<dim3(grid), dim3(kMergeThreads), 0, stream>>>(p); return hipGetLastError();
      default: break;
    }
  }
  merge_n_kernel<<<dim3(grid), dim3(kMergeThreads), 0, stream>>>(p);
  return hipGetLastError();
}

hipError_t launch_merge_2(const Merge2Params& p, hipStream_t stream) {
  const int64_t items = (int64_t)p.seq_len * p.num_heads;
  if (items == 0) return hipSuccess;
  const int grid = (int)((items + kMergeWaves - 1) / kMergeWaves);
  merge_2_kernel<<<dim3(grid), dim3(kMergeThreads), 0, stream>>>(p);
  return hipGetLastError();
}

static bool merge_dtype_ok(int dt) {
  return dt == FI_DTYPE_F16 || dt == FI_DTYPE_BF16 || dt == FI_DTYPE_F32;
}

}  // namespace fi

using namespace fi;

extern "C" FI_API int fi_merge_state(const void* v_a, const float* s_a, const void* v_b, const float* s_b,
                              void* v_merged, float* s_merged, int32_t seq_len, int32_t num_heads,
                              int32_t head_dim, int32_t dtype, fi_stream_t stream) {
  if (seq_len == 0 || num_heads == 0) return 0;
  FI_REQUIRE(v_a && s_a && v_b && s_b && v_merged, "merge_state: null tensor");
  FI_REQUIRE(merge_dtype_ok(dtype), "merge_state: unsupported dtype %d", dtype);
  FI_REQUIRE(head_dim > 0 && head_dim <= 64 * kMergeMaxPerLane, "merge_state: head_dim %d unsupported", head_dim);
  Merge2Params p{v_a, s_a, v_b, s_b, v_merged, s_merged, nullptr, seq_len, num_heads, head_dim, dtype};
  FI_HIP_CALL(launch_merge_2(p, (hipStream_t)stream));
  return 0;
}

extern "C" FI_API int fi_merge_state_in_place(void* v, float* s, const void* v_other, const float* s_other,
                                       const uint8_t* mask, int32_t seq_len, int32_t num_heads,
                                       int32_t head_dim, int32_t dtype, fi_stream_t stream) {
  if (seq_len == 0 || num_heads == 0) return 0;
  FI_REQUIRE(v && s && v_other && s_other, "merge_state_in_place: null tensor");
  FI_REQUIRE(merge_dtype_ok(dtype), "merge_state_in_place: unsupported dtype %d", dtype);
  FI_REQUIRE(head_dim > 0 && head_dim <= 64 * kMergeMaxPerLane, "merge_state_in_place: head_dim %d unsupported", head_dim);
  Merge2Params p{v, s, v_other, s_other, v, s, mask, seq_len, num_heads, head_dim, dtype};
  FI_HIP_CALL(launch_merge_2(p, (hipStream_t)stream));
  return 0;
}

extern "C" FI_API int fi_merge_states(const void* v, const float* s, void* v_merged, float* s_merged,
                               int32_t num_index_sets, int32_t seq_len, int32_t num_heads,
                               int32_t head_dim, int32_t dtype, fi_stream_t stream) {
  if (seq_len == 0 || num_heads == 0) return 0;
  FI_REQUIRE((v && s || num_index_sets == 0) && v_merged, "merge_states: null tensor");
  FI_REQUIRE(merge_dtype_ok(dtype), "merge_states: unsupported dtype %d", dtype);
  FI_REQUIRE(head_dim > 0 && head_dim <= 64 * kMergeMaxPerLane, "merge_states: head_dim %d unsupported", head_dim);
  FI_REQUIRE(num_index_sets >= 0, "merge_states: negative num_index_sets");
  MergeNParams p{v, s, nullptr, v_merged, s_merged, num_index_sets, seq_len, num_heads, head_dim, dtype, dtype};
  FI_HIP_CALL(launch_merge_n(p, (hipStream_t)stream));
  return 0;
}

extern "C" FI_API int fi_variable_length_merge_states(const void* v, const float* s, const int32_t* indptr,
                                               void* v_merged, float* s_merged, int32_t seq_len,
                                               int32_t num_heads, int32_t head_dim, int32_t in_dtype,
                                               int32_t out_dtype, fi_stream_t stream) {
  if (seq_len == 0 || num_heads == 0) return 0;
  FI_REQUIRE(indptr && v_merged, "variable_length_merge_states: null tensor");
  FI_REQUIRE(merge_dtype_ok(in_dtype) && merge_dtype_ok(out_dtype), "variable_length_merge_states: unsupported dtype");
  FI_REQUIRE(head_dim > 0 && head_dim <= 64 * kMergeMaxPerLane, "variable_length_merge_states: head_dim %d unsupported", head_dim);
  MergeNParams p{v, s, indptr, v_merged, s_merged, 0, seq_len, num_heads, head_dim, in_dtype, out_dtype};
  FI_HIP_CALL(launch_merge_n(p, (hipStream_t)stream));
  return 0;
}

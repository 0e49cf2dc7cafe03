// Standalone rotary-embedding kernels (the step callers run right before the attention path).
// ref: include/flashinfer/pos_enc.cuh:123-209 (rotation forms), 465-640 (kernels), 820-1070 (launchers);
//      Python flashinfer/rope.py:321-1150.
// HBM-bound elementwise op: one thread rotates 16 elements of one (token, head) -- two 16-byte chunks that
// contain each other's rotation partners (adjacent chunks when interleaved, chunks rotary_dim/2 apart
// otherwise) -- so every load and store is a full 16-byte vector and no cross-lane traffic is needed.
// Angles: theta_i = pos * freq_i in f32 (as the reference), freq_i = rope_rcp_theta^(2 j / rotary_dim)
// blended by the llama-3.1 smooth factor (pos_enc.cuh:491-493); sin/cos by the hardware v_sin/v_cos after
// a two-term Cody-Waite reduction, or read from a caller-supplied cos/sin table.
#include <string.h>

#include <algorithm>

#include "common.h"

namespace fi {

constexpr int kRopeThreads = 256;

struct RopeParams {
  const void* q;
  const void* k;
  void* q_out;
  void* k_out;
  const int32_t* pos_ids;        // [nnz]
  const float* cos_sin_cache;    // optional [max_pos, rotary_dim]: cos | sin halves
  int64_t q_stride_n, q_stride_h, k_stride_n, k_stride_h;
  int64_t qo_stride_n, qo_stride_h, ko_stride_n, ko_stride_h;
  int32_t nnz, num_q_heads, num_k_heads, head_dim, rotary_dim;
  int32_t interleave, dtype;
  int32_t heads_per_thread;  // heads one thread walks (all of them when there are enough tokens to fill the chip)
  float rope_rcp_scale, rope_rcp_theta, smooth_a, smooth_b;
  // fused append (fi_apply_rope_append_paged_kv_cache): the rotated k rows go straight into the paged cache at
  // (batch_indices[i], positions[i]) and the v rows are copied beside them; k_cache == nullptr: plain RoPE
  void* k_cache;
  void* v_cache;
  const void* v;
  const int32_t* batch_indices;
  const int32_t* positions;
  const int32_t* kv_indptr;
  const int32_t* kv_indices;
  int64_t v_stride_n, v_stride_h;
  int64_t c_stride_page, c_stride_n, c_stride_h;
  FastDiv page_div;
  int32_t page_size;
};

// One thread owns one (token, chunk pair) and walks ALL q and k heads of that token: the angles depend on
// (position, pair index) only, so the 8 sin/cos values (a powf, a range reduction and two transcendental
// instructions each) are computed once and reused for every head; per head only the loads, the rotation
// FMAs and the stores remain (measured at 32k tokens, 32 + 8 heads: 1.4 -> 4.4 TB/s of read + write traffic).
// Small batches split the heads over several threads (heads_per_thread) to keep the chip busy.
__global__ void __launch_bounds__(kRopeThreads) rope_kernel(const RopeParams p) {
  const int cph = p.head_dim / 8;             // 16-byte chunks per head row
  const int rot_chunks = p.rotary_dim / 8;    // chunks inside the rotary part
  const int pairs = rot_chunks / 2;           // threads that rotate (2 chunks each)
  const int pass = cph - rot_chunks;          // pass-through chunks (only copied when out of place)
  const int tph = pairs + (pass + 1) / 2;     // threads per token
  const int heads = p.num_q_heads + p.num_k_heads;
  const int hgroups = (heads + p.heads_per_thread - 1) / p.heads_per_thread;
  const int64_t total = (int64_t)p.nnz * hgroups * tph;
  for (int64_t it = (int64_t)blockIdx.x * kRopeThreads + threadIdx.x; it < total;
       it += (int64_t)gridDim.x * kRopeThreads) {
    const int t = (int)(it % tph);
    const int64_t r_ = it / tph;
    const int hg = (int)(r_ % hgroups);
    const int tok = (int)(r_ / hgroups);
    const int h_begin = hg * p.heads_per_thread, h_end = min(heads, h_begin + p.heads_per_thread);
    const uint16_t* const q_src = (const uint16_t*)p.q + (int64_t)tok * p.q_stride_n;
    const uint16_t* const k_src = (const uint16_t*)p.k + (int64_t)tok * p.k_stride_n;
    uint16_t* const q_dst = (uint16_t*)p.q_out + (int64_t)tok * p.qo_stride_n;
    uint16_t* k_dst = (uint16_t*)p.k_out + (int64_t)tok * p.ko_stride_n;
    int64_t ko_stride_h = p.ko_stride_h;
    const uint16_t* v_src = nullptr;
    uint16_t* v_dst = nullptr;
    if (p.k_cache) {
      // ref: page.cuh:272-275 -- page_iter = indptr[b] + pos / page_size, entry = pos % page_size
      const int b = p.batch_indices[tok];
      const int apos = p.positions[tok];
      const int pi = (int)fast_div((uint32_t)apos, p.page_div);
      const int entry = apos - pi * p.page_size;
      const int64_t base = (int64_t)p.kv_indices[p.kv_indptr[b] + pi] * p.c_stride_page + (int64_t)entry * p.c_stride_n;
      k_dst = (uint16_t*)p.k_cache + base;
      v_dst = (uint16_t*)p.v_cache + base;
      v_src = (const uint16_t*)p.v + (int64_t)tok * p.v_stride_n;
      ko_stride_h = p.c_stride_h;
    }
    if (t >= pairs) {  // pass-through part
      const int c0 = rot_chunks + 2 * (t - pairs);
      for (int h = h_begin; h < h_end; ++h) {
        const bool is_q = h < p.num_q_heads;
        const int hh = is_q ? h : h - p.num_q_heads;
        const uint16_t* src = (is_q ? q_src : k_src) + (int64_t)hh * (is_q ? p.q_stride_h : p.k_stride_h);
        uint16_t* dst = (is_q ? q_dst : k_dst) + (int64_t)hh * (is_q ? p.qo_stride_h : ko_stride_h);
        if (v_dst && !is_q) {
          const uint16_t* vs = v_src + (int64_t)hh * p.v_stride_h;
          uint16_t* vd = v_dst + (int64_t)hh * p.c_stride_h;
          *(u32x4*)(vd + 8 * c0) = *(const u32x4*)(vs + 8 * c0);
          if (c0 + 1 < cph) *(u32x4*)(vd + 8 * (c0 + 1)) = *(const u32x4*)(vs + 8 * (c0 + 1));
        }
        if (src == dst) continue;
        *(u32x4*)(dst + 8 * c0) = *(const u32x4*)(src + 8 * c0);
        if (c0 + 1 < cph) *(u32x4*)(dst + 8 * (c0 + 1)) = *(const u32x4*)(src + 8 * (c0 + 1));
      }
      continue;
    }
    // chunks a, b; pair index of element j: interleaved -> elements (2m, 2m+1) of the 16-element span are
    // pair 8t + m; otherwise element j of chunk a pairs with element j of chunk b: pair 8t + j
    const int ca = p.interleave ? 2 * t : t;
    const int cb = p.interleave ? 2 * t + 1 : t + pairs;
    const int pos = p.pos_ids[tok];
    float sn[8], cs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int m = 8 * t + j;
      if (p.cos_sin_cache) {
        const float* row = p.cos_sin_cache + (int64_t)pos * p.rotary_dim;
        cs[j] = row[m];
        sn[j] = row[p.rotary_dim / 2 + m];
      } else {
        float freq = __powf(p.rope_rcp_theta, (float)(2 * m) / (float)p.rotary_dim);
        float smooth = fminf(fmaxf(freq * p.smooth_a + p.smooth_b, 0.f), 1.f);
        freq = (1.f - smooth) * (freq * p.rope_rcp_scale) + smooth * freq;
        fast_sincos((float)pos * freq, &sn[j], &cs[j]);
      }
    }
#pragma unroll 4
    for (int h = h_begin; h < h_end; ++h) {
      const bool is_q = h < p.num_q_heads;
      const int hh = is_q ? h : h - p.num_q_heads;
      const uint16_t* src = (is_q ? q_src : k_src) + (int64_t)hh * (is_q ? p.q_stride_h : p.k_stride_h);
      uint16_t* dst = (is_q ? q_dst : k_dst) + (int64_t)hh * (is_q ? p.qo_stride_h : ko_stride_h);
      const u32x4 ra = *(const u32x4*)(src + 8 * ca);
      const u32x4 rb = *(const u32x4*)(src + 8 * cb);
      if (v_dst && !is_q) {  // the v row's two chunks ride along
        const uint16_t* vs = v_src + (int64_t)hh * p.v_stride_h;
        uint16_t* vd = v_dst + (int64_t)hh * p.c_stride_h;
        *(u32x4*)(vd + 8 * ca) = *(const u32x4*)(vs + 8 * ca);
        *(u32x4*)(vd + 8 * cb) = *(const u32x4*)(vs + 8 * cb);
      }
      float xa[8], xb[8];
      if (p.dtype == FI_DTYPE_BF16) {
        KVTraits<FI_DTYPE_BF16>::unpack(ra, xa);
        KVTraits<FI_DTYPE_BF16>::unpack(rb, xb);
      } else {
        KVTraits<FI_DTYPE_F16>::unpack(ra, xa);
        KVTraits<FI_DTYPE_F16>::unpack(rb, xb);
      }
      float ya[8], yb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x0, x1;
        if (p.interleave) {
          // pair m lives at span elements 2j, 2j+1 (span = chunk a then chunk b)
          x0 = (j < 4) ? xa[2 * j] : xb[2 * j - 8];
          x1 = (j < 4) ? xa[2 * j + 1] : xb[2 * j - 7];
        } else {
          x0 = xa[j];
          x1 = xb[j];
        }
        const float y0 = x0 * cs[j] - x1 * sn[j];  // ref: pos_enc.cuh:93-96, 143-145
        const float y1 = x1 * cs[j] + x0 * sn[j];
        if (p.interleave) {
          if (j < 4) { ya[2 * j] = y0; ya[2 * j + 1] = y1; }
          else { yb[2 * j - 8] = y0; yb[2 * j - 7] = y1; }
        } else {
          ya[j] = y0;
          yb[j] = y1;
        }
      }
      u32x4 wa, wb;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wa[j] = (uint32_t)f32_to_16bit(ya[2 * j], p.dtype) | ((uint32_t)f32_to_16bit(ya[2 * j + 1], p.dtype) << 16);
        wb[j] = (uint32_t)f32_to_16bit(yb[2 * j], p.dtype) | ((uint32_t)f32_to_16bit(yb[2 * j + 1], p.dtype) << 16);
      }
      *(u32x4*)(dst + 8 * ca) = wa;
      *(u32x4*)(dst + 8 * cb) = wb;
    }
  }
}

__global__ void rope_positions_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ offsets,
                                      int32_t* pos, int nnz) {
  const int b = blockIdx.x;
  const int lo = indptr[b], hi = indptr[b + 1];
  const int off = offsets[b];
  for (int i = lo + threadIdx.x; i < hi && i < nnz; i += blockDim.x) pos[i] = off + i - lo;
}

}  // namespace fi

using namespace fi;

extern "C" FI_API int fi_rope_positions_from_indptr(const int32_t* indptr, const int32_t* offsets,
                                                    int32_t batch_size, int32_t nnz, int32_t* pos_ids,
                                                    fi_stream_t stream) {
  if (batch_size == 0 || nnz == 0) return 0;
  FI_REQUIRE(indptr && offsets && pos_ids, "rope_positions_from_indptr: null tensor");
  rope_positions_kernel<<<dim3(batch_size), dim3(256), 0, (hipStream_t)stream>>>(indptr, offsets, pos_ids, nnz);
  FI_HIP_CALL(hipGetLastError());
  return 0;
}

static int rope_fill_and_launch(const char* who, const fi_rope_params_t* a, RopeParams& p, fi_stream_t stream) {
  FI_REQUIRE(a->q && a->k && a->q_out && a->pos_ids, "%s: null tensor", who);
  FI_REQUIRE(a->dtype == FI_DTYPE_F16 || a->dtype == FI_DTYPE_BF16, "%s: dtype must be f16/bf16", who);
  FI_REQUIRE(a->head_dim % 8 == 0 && a->rotary_dim % 16 == 0 && a->rotary_dim > 0 && a->rotary_dim <= a->head_dim,
             "%s: head_dim must be a multiple of 8 and rotary_dim of 16 (got %d / %d)", who, a->head_dim,
             a->rotary_dim);
  const int64_t strides[6] = {a->q_stride_n, a->q_stride_h, a->k_stride_n, a->k_stride_h, a->qo_stride_n, a->qo_stride_h};
  for (int64_t s : strides) FI_REQUIRE(s % 8 == 0, "%s: rows must be 16-byte aligned", who);
  FI_REQUIRE(((uintptr_t)a->q % 16) == 0 && ((uintptr_t)a->k % 16) == 0 && ((uintptr_t)a->q_out % 16) == 0,
             "%s: tensors must be 16-byte aligned", who);
  p.q = a->q; p.k = a->k; p.q_out = a->q_out; p.k_out = a->k_out;
  p.pos_ids = a->pos_ids; p.cos_sin_cache = a->cos_sin_cache;
  p.q_stride_n = a->q_stride_n; p.q_stride_h = a->q_stride_h; p.k_stride_n = a->k_stride_n; p.k_stride_h = a->k_stride_h;
  p.qo_stride_n = a->qo_stride_n; p.qo_stride_h = a->qo_stride_h; p.ko_stride_n = a->ko_stride_n; p.ko_stride_h = a->ko_stride_h;
  p.nnz = a->nnz; p.num_q_heads = a->num_q_heads; p.num_k_heads = a->num_k_heads;
  p.head_dim = a->head_dim; p.rotary_dim = a->rotary_dim; p.interleave = a->interleave; p.dtype = a->dtype;
  p.rope_rcp_scale = a->rope_rcp_scale; p.rope_rcp_theta = a->rope_rcp_theta;
  p.smooth_a = a->smooth_a; p.smooth_b = a->smooth_b;
  const int cph = a->head_dim / 8, rot_chunks = a->rotary_dim / 8;
  const int tph = rot_chunks / 2 + (cph - rot_chunks + 1) / 2;
  // one thread per (token, head group, chunk pair): all heads per thread once the tokens alone give
  // >= 64K threads, fewer heads per thread (more threads) for small batches
  const int heads = a->num_q_heads + a->num_k_heads;
  const int64_t base_threads = (int64_t)a->nnz * tph;
  p.heads_per_thread = (int)std::min<int64_t>(heads, std::max<int64_t>(1, heads * base_threads / 65536));
  const int64_t total = base_threads * ((heads + p.heads_per_thread - 1) / p.heads_per_thread);
  const int grid = (int)std::min<int64_t>((total + kRopeThreads - 1) / kRopeThreads, 256 * 16);
  rope_kernel<<<dim3(grid), dim3(kRopeThreads), 0, (hipStream_t)stream>>>(p);
  FI_HIP_CALL(hipGetLastError());
  return 0;
}

extern "C" FI_API int fi_apply_rope_pos_ids(const fi_rope_params_t* a, fi_stream_t stream) {
  FI_REQUIRE(a, "apply_rope_pos_ids: null params");
  if (a->nnz == 0) return 0;
  FI_REQUIRE(a->k_out && ((uintptr_t)a->k_out % 16) == 0 && a->ko_stride_n % 8 == 0 && a->ko_stride_h % 8 == 0,
             "apply_rope_pos_ids: k_out must be a 16-byte aligned tensor");
  RopeParams p;
  memset(&p, 0, sizeof(p));
  return rope_fill_and_launch("apply_rope_pos_ids", a, p, stream);
}

extern "C" FI_API int fi_apply_rope_append_paged_kv_cache(const fi_rope_params_t* a, const void* append_value,
                                                          int64_t v_stride_n, int64_t v_stride_h,
                                                          const int32_t* batch_indices, const int32_t* positions,
                                                          const fi_paged_kv_t* kv, fi_stream_t stream) {
  const char* who = "apply_rope_append_paged_kv_cache";
  FI_REQUIRE(a && kv, "%s: null params", who);
  if (a->nnz == 0) return 0;
  FI_REQUIRE(append_value && batch_indices && positions, "%s: null tensor", who);
  FI_REQUIRE(kv->k_data && kv->v_data && kv->indptr && kv->indices, "%s: null cache", who);
  FI_REQUIRE(kv->dtype == a->dtype, "%s: the cache must have the dtype of k (rotated rows are stored as they are)", who);
  FI_REQUIRE(kv->head_dim == a->head_dim && kv->num_kv_heads == a->num_k_heads,
             "%s: cache is %d heads x %d, k is %d x %d", who, kv->num_kv_heads, kv->head_dim, a->num_k_heads,
             a->head_dim);
  FI_REQUIRE(kv->stride_n % 8 == 0 && kv->stride_h % 8 == 0 && kv->stride_page % 8 == 0 && v_stride_n % 8 == 0 &&
                 v_stride_h % 8 == 0 && ((uintptr_t)append_value % 16) == 0 && ((uintptr_t)kv->k_data % 16) == 0 &&
                 ((uintptr_t)kv->v_data % 16) == 0,
             "%s: rows must be 16-byte aligned", who);
  RopeParams p;
  memset(&p, 0, sizeof(p));
  p.k_cache = (void*)kv->k_data;
  p.v_cache = (void*)kv->v_data;
  p.v = append_value;
  p.batch_indices = batch_indices;
  p.positions = positions;
  p.kv_indptr = kv->indptr;
  p.kv_indices = kv->indices;
  p.v_stride_n = v_stride_n;
  p.v_stride_h = v_stride_h;
  p.c_stride_page = kv->stride_page;
  p.c_stride_n = kv->stride_n;
  p.c_stride_h = kv->stride_h;
  p.page_div = FastDiv((uint32_t)kv->page_size);
  p.page_size = kv->page_size;
  return rope_fill_and_launch(who, a, p, stream);
}

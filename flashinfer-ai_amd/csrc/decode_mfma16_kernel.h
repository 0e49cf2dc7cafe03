// Batch paged-KV decode on the matrix cores for GQA groups of up to 16 query heads: the 16x16x32 form of
// decode_mfma_kernel.h (same work list, same wave-private K | V LDS tiles of 32 tokens, no workgroup barrier).
//
//   S^T[16 kv][16 heads] = K Q^T  (two 16-row halves of the tile),   O^T[16 d][16 heads] += V^T P^T (k = 32 kv)
//
// Against the 32x32x16 form a wave keeps 32 instead of 64 accumulator registers, 16 instead of 32 for Q and 8
// instead of 16 for the logits, which is what lets the fused-RoPE variant (ROPE: cos / sin state + rotation
// temporaries) stay under 256 registers, i.e. two workgroups per CU -- the 32x32 form needs ~370 for it and the
// VALU kernel is vector-ALU bound (4.8 TB/s) with the rotation.
//
// Layouts (g = lane >> 4, c = lane & 15):
//   QK^T  A = K: lane holds K[16 t + c][32 ks + 8 g .. + 8]  (ds_read_b128, chunk-swizzled rows)
//         B = Q: lane holds Q[head c][32 ks + 8 g .. + 8]     (registers for the whole kernel)
//         C    : s[t][j] = S[kv 16 t + 4 g + j][head c]
//   P.V   B = P: element 4 t + j of the lane = p[t][j]        (k index 8 g + 4 t + j <-> kv row 16 t + 4 g + j)
//         A = V^T: lane holds V[those 8 rows][16 db + c]      (two ds_read_b64_tr_b16: rows 4 g.., 16 + 4 g..)
//         C    : o[db][j] = O[head c][16 db + 4 g + j]
// ROPE (pos_encoding_mode = ROPE_LLAMA; ref: decode.cuh:445-466, and the reference's tensor-core decode = its
// prefill kernel, prefill.cuh:465-612): q rotated once, every K row while it is staged into LDS; both rounded to
// T16 for the MFMA as that path rounds them in shared memory.
#pragma once
#include "decode_kernel.h"
#include "prefill_kernel.h"

namespace fi {

template <int T16>
struct Mfma16;
template <>
struct Mfma16<FI_DTYPE_F16> {
  static __device__ __forceinline__ f32x4 mfma(MfmaType<FI_DTYPE_F16>::frag a, MfmaType<FI_DTYPE_F16>::frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Mfma16<FI_DTYPE_BF16> {
  static __device__ __forceinline__ f32x4 mfma(MfmaType<FI_DTYPE_BF16>::frag a, MfmaType<FI_DTYPE_BF16>::frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

// max / sum over the four 16-lane rows of a wave (v_permlane16_swap + v_permlane32_swap; VALU, no LDS).
// The results are copied to scalars before the bit_cast: __builtin_bit_cast of a vector-element lvalue reads
// element 0 with this compiler (hipcc, ROCm 7.2).
template <bool MAX>
__device__ __forceinline__ float reduce_rows(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // {rows 0 0 2 2}, {rows 1 1 3 3}
  const uint32_t r0 = r[0], r1 = r[1];
  const float a = __builtin_bit_cast(float, r0), b = __builtin_bit_cast(float, r1);
  const float y = MAX ? fmaxf(a, b) : a + b;
  const uint32_t v = __builtin_bit_cast(uint32_t, y);
  const auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // {lo lo}, {hi hi}
  const uint32_t q0 = q[0], q1 = q[1];
  const float c = __builtin_bit_cast(float, q0), d = __builtin_bit_cast(float, q1);
  return MAX ? fmaxf(c, d) : c + d;
}

// (an fp8 cache with ROPE at head_dim 128 stages 16 dims per lane -- twice the rotation state -- and does not fit
// 256 registers: that instantiation runs one workgroup per CU)
template <int T16, int KVS, int D, bool PAGED, bool ROPE>
__global__ void __launch_bounds__(kDecodeThreads, (ROPE && D == 128 && KVS != T16) ? 1 : 2)
    decode_mfma16_kernel(const DecodeKernelParams p) {
  using M = MfmaType<T16>;
  using frag_t = typename M::frag;
  constexpr int kTile = 32;
  constexpr bool KV_FP8 = (KVS == FI_DTYPE_FP8_E4M3 || KVS == FI_DTYPE_FP8_E5M2);
  constexpr int KV_BYTES = KV_FP8 ? 1 : 2;
  constexpr int ROWB = D * 2;             // bytes per row of the (16-bit) LDS images
  constexpr int CPR = D / 8;              // 16-byte chunks per LDS row
  constexpr int GCH = D * KV_BYTES / 16;  // 16-byte chunks per row in the cache
  constexpr int RPP = 64 / GCH;           // rows staged per pass by one wave
  constexpr int NPASS = kTile / RPP;
  constexpr int KSTEPS = D / 32;
  constexpr int DBLK = D / 16;
  constexpr int TILE_BYTES = kTile * ROWB;
  constexpr int NE = KV_FP8 ? 16 : 8;     // dims one lane stages per pass
  static_assert(D == 64 || D == 128, "head_dim 64 / 128");

  __shared__ __attribute__((aligned(16))) char smem[kDecodeWaves][2 * TILE_BYTES];  // per wave: K | V
  // fused RoPE: cos / signed sin of (rows per pass) x freq for the NE dims of every staging chunk
  __shared__ __attribute__((aligned(16))) float rope_step[ROPE ? GCH * NE * 2 : 4];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int lb;  // XCD-contiguous logical block id (see decode_kernel.h)
  {
    const int b = blockIdx.x, total = gridDim.x;
    const int xcd = b & 7, slot = b >> 3;
    const int qn = total >> 3, rn = total & 7;
    lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
  }
  const int item = lb * kDecodeWaves + wave;
  if (item >= p.num_items) return;
  char* const kb = smem[wave];
  char* const vb = kb + TILE_BYTES;
  const int lc = lane & 15, lg = lane >> 4;

  // ---- item -> (work, kv head); one head tile: the whole group (<= 16 heads) is this wave's 16 columns ----
  const int kv_head = item % p.num_kv_heads;
  const int work = item / p.num_kv_heads;
  int req = 0, kv_tile = work;
  if (p.request_indices) {
    if (p.block_valid_mask && !p.block_valid_mask[work]) return;
    req = p.request_indices[work];
    kv_tile = p.kv_tile_indices[work];
  }
  int page_begin = 0, kv_len;
  if (p.indptr) {
    page_begin = p.indptr[req];
    const int np = p.indptr[req + 1] - page_begin;
    kv_len = np > 0 ? (np - 1) * p.page_size + p.last_page_len[req] : 0;
  } else {
    kv_len = p.single_kv_len;
  }
  // sliding window / planned window / chunk bounds: as in decode_mfma_kernel.h
  int chunk_base = 0;
  if (p.plan_window_left >= 0 && p.indptr) {
    const int np = p.indptr[req + 1] - page_begin;
    chunk_base = np > 0 ? max((np - 1) * p.page_size - p.plan_window_left, 0) / p.page_size * p.page_size : 0;
  }
  const int win_start = p.window_left >= 0 ? max(0, kv_len - 1 - p.window_left) : 0;
  const int kv_chunk_size = p.kv_chunk_size_ptr ? *p.kv_chunk_size_ptr : p.kv_chunk_size;
  int chunk_start = chunk_base + (p.split_kv ? kv_tile * kv_chunk_size : 0);
  const int chunk_end = p.split_kv ? min(chunk_start + kv_chunk_size, kv_len) : kv_len;
  if (win_start > chunk_start) chunk_start += (win_start - chunk_start) / kTile * kTile;
  const int G = p.group_size;  // <= 16 (checked on the host)
  const int head0 = kv_head * G;
  const int head = head0 + min(lc, G - 1);

  // ---- Q fragments ----
  frag_t qf[KSTEPS];
  {
    const uint16_t* qrow = (const uint16_t*)p.q + (int64_t)req * p.q_stride_n + (int64_t)head * p.q_stride_h;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
      qf[ks] = __builtin_bit_cast(frag_t, *(const u32x4*)(qrow + 32 * ks + 8 * lg));
    if constexpr (ROPE) {
      // dims i and i + D/2 pair up: k-steps ks and ks + KSTEPS/2 of the same lane; q sits at kv_len - 1 unless
      // the caller passed its position
      const int q_pos = p.q_rope_offset ? p.q_rope_offset[req] : (kv_len - 1);
#pragma unroll
      for (int ks = 0; ks < KSTEPS / 2; ++ks) {
        const u32x4 lo4 = __builtin_bit_cast(u32x4, qf[ks]), hi4 = __builtin_bit_cast(u32x4, qf[ks + KSTEPS / 2]);
        u32x4 out_lo, out_hi;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint32_t lo_w = lo4[w], hi_w = hi4[w];
          float ra[2], rb[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int i = 32 * ks + 8 * lg + 2 * w + e;  // < D/2
            const float freq = p.rope_rcp_scale * __powf(p.rope_rcp_theta, (float)(2 * i) / (float)D);
            float sn, cs;
            sincos_ool((float)q_pos * freq, &sn, &cs);
            const float a = M::to_f32((uint16_t)(lo_w >> (16 * e)));
            const float b = M::to_f32((uint16_t)(hi_w >> (16 * e)));
            ra[e] = a * cs - b * sn;
            rb[e] = b * cs + a * sn;
          }
          out_lo[w] = pack2<T16>(ra[0], ra[1]);
          out_hi[w] = pack2<T16>(rb[0], rb[1]);
        }
        qf[ks] = __builtin_bit_cast(frag_t, out_lo);
        qf[ks + KSTEPS / 2] = __builtin_bit_cast(frag_t, out_hi);
      }
    }
  }
  // logits: plain -> exp2 argument = s * sm_scale * log2(e) (folded into the FMA below).  ALiBi / soft cap (ref:
  // variants.cuh:67-76 -- bias with qo_idx = 0 as the decode kernels pass it, decode.cuh:93-94, then the cap) are
  // applied to the 8 logits a lane holds per tile and leave them in base-2 units (c = 1).
  const bool general = p.use_alibi || p.logits_soft_cap > 0.f;
  const float c_log2 = general ? 1.0f : p.sm_scale * kLog2e;
  const float slope = p.use_alibi ? p.alibi_slopes[head] : 0.f;
  const float inv_cap = p.logits_soft_cap > 0.f ? 1.0f / p.logits_soft_cap : 0.f;

  // ---- staging: pass ps covers rows ps * RPP + lane / GCH, cache chunk lane % GCH ----
  const int st_row = lane / GCH, st_ch = lane % GCH;
  const int64_t thread_off = (int64_t)kv_head * p.kv_stride_h + st_ch * (16 / KV_BYTES);
  const uint32_t stride_page32 = (uint32_t)p.kv_stride_page, stride_n32 = (uint32_t)p.kv_stride_n;
  auto fetch_pages = [&](int tok0, int (&pg)[NPASS], int (&en)[NPASS]) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int kvi = max(min(tok0 + ps * RPP + st_row, chunk_end - 1), 0);
      const int pi = (int)fast_div((uint32_t)kvi, p.page_div);
      en[ps] = kvi - pi * p.page_size;
      pg[ps] = PAGED ? p.indices[page_begin + pi] : pi;
    }
  };
  struct Stage {
    u32x4 k[NPASS], v[NPASS];
  };
  auto issue_loads = [&](const int (&pg)[NPASS], const int (&en)[NPASS], Stage& st) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int64_t off = (int64_t)((uint64_t)(uint32_t)pg[ps] * stride_page32 +
                                    (uint64_t)(uint32_t)en[ps] * stride_n32) + thread_off;
      st.k[ps] = __builtin_nontemporal_load((const u32x4*)((const char*)p.k + off * KV_BYTES));
      st.v[ps] = __builtin_nontemporal_load((const u32x4*)((const char*)p.v + off * KV_BYTES));
    }
  };
  // K image: 16-byte chunks XOR-swizzled so that the 16 rows one ds_read_b128 lane group touches spread over
  // the banks.  V image: 32-byte slots (= one 16-column d block) swizzled per row for ds_read_b64_tr_b16, whose
  // 32-lane halves read 8 rows x 32 bytes.
  auto k_lds_off = [&](int row, int ch) -> int {
    const int sw = (CPR >= 16) ? (row & 15) : ((row >> 1) & 7);
    return row * ROWB + ((ch ^ sw) << 4);
  };
  auto v_slot_key = [&](int row) -> int { return (ROWB >= 256) ? (row & 7) : ((row >> 1) & 3); };
  auto v_lds_off = [&](int row, int ch) -> int {  // ch: 16-byte chunk of the row
    return row * ROWB + (((ch >> 1) ^ v_slot_key(row)) << 5) + ((ch & 1) << 4);
  };

  // ---- fused RoPE on K (see decode_mfma_kernel.h): per-lane cos / sin recurrence over the passes, re-seeded
  // every 4 tiles; the per-pass rotation lives in LDS, the sign of the sine term is folded into the state ----
  [[maybe_unused]] float rc[ROPE ? NE : 1], rs[ROPE ? NE : 1];
  [[maybe_unused]] const int rope_pos0 = ROPE && p.kv_rope_pos_offset ? p.kv_rope_pos_offset[req] : 0;
  [[maybe_unused]] auto rope_freq = [&](int e) {
    const int i = (st_ch * NE + e) % (D / 2);
    return p.rope_rcp_scale * __powf(p.rope_rcp_theta, (float)(2 * i) / (float)D);
  };
  [[maybe_unused]] const float rope_sgn = (st_ch < GCH / 2) ? -1.f : 1.f;
  if constexpr (ROPE) {
    // every wave writes the same values: no workgroup barrier (waves that returned above never reach one)
    if (st_row == 0) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        float sn, cs;
        fast_sincos((float)RPP * rope_freq(e), &sn, &cs);
        rope_step[st_ch * NE * 2 + e] = cs;
        rope_step[st_ch * NE * 2 + NE + e] = rope_sgn * sn;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  [[maybe_unused]] auto rope_seed = [&](int pos) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      float sn, cs;
      fast_sincos((float)pos * rope_freq(e), &sn, &cs);
      rc[e] = cs;
      rs[e] = rope_sgn * sn;
    }
  };
  [[maybe_unused]] auto rope_chunk = [&](u32x4 kw, int e0) {  // rotate 8 dims (state slots e0 .. e0 + 7)
    u32x4 pw, outw;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t mine = kw[w];
      pw[w] = __builtin_bit_cast(uint32_t, lane_xor<GCH / 2>(__builtin_bit_cast(float, mine)));
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float y[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float x = M::to_f32((uint16_t)(kw[w] >> (16 * e)));
        const float partner = M::to_f32((uint16_t)(pw[w] >> (16 * e)));
        y[e] = __builtin_fmaf(partner, rs[e0 + 2 * w + e], x * rc[e0 + 2 * w + e]);
      }
      outw[w] = pack2<T16>(y[0], y[1]);
    }
    return outw;
  };
  [[maybe_unused]] uint32_t step_off = (uint32_t)(st_ch * NE * 2) * 4u;
  [[maybe_unused]] auto rope_advance = [&]() {
    asm volatile("" : "+v"(step_off));  // keeps the table reads inside the loop (2 NE registers otherwise)
    const float* const tab = (const float*)((const char*)rope_step + step_off);
#pragma unroll
    for (int e4 = 0; e4 < NE; e4 += 4) {
      const f32x4 dc4 = *(const f32x4*)(tab + e4), ds4 = *(const f32x4*)(tab + NE + e4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float cn = rc[e4 + e] * dc4[e] - rs[e4 + e] * ds4[e];
        const float sn = rs[e4 + e] * dc4[e] + rc[e4 + e] * ds4[e];
        rc[e4 + e] = cn;
        rs[e4 + e] = sn;
      }
    }
  };
  auto write_stage = [&](const Stage& st) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int row = ps * RPP + st_row;
      if constexpr (KV_FP8) {  // 16 fp8 -> two 16-byte chunks of T16
        u32x4 k0 = fp8x8_to_16<T16, KVS>(u32x2{st.k[ps][0], st.k[ps][1]});
        u32x4 k1 = fp8x8_to_16<T16, KVS>(u32x2{st.k[ps][2], st.k[ps][3]});
        if constexpr (ROPE) {
          k0 = rope_chunk(k0, 0);
          k1 = rope_chunk(k1, 8);
          rope_advance();
        }
        *(u32x4*)(kb + k_lds_off(row, 2 * st_ch)) = k0;
        *(u32x4*)(kb + k_lds_off(row, 2 * st_ch + 1)) = k1;
        *(u32x4*)(vb + v_lds_off(row, 2 * st_ch)) = fp8x8_to_16<T16, KVS>(u32x2{st.v[ps][0], st.v[ps][1]});
        *(u32x4*)(vb + v_lds_off(row, 2 * st_ch + 1)) = fp8x8_to_16<T16, KVS>(u32x2{st.v[ps][2], st.v[ps][3]});
      } else {
        u32x4 k0 = st.k[ps];
        if constexpr (ROPE) {
          k0 = rope_chunk(k0, 0);
          rope_advance();
        }
        *(u32x4*)(kb + k_lds_off(row, st_ch)) = k0;
        *(u32x4*)(vb + v_lds_off(row, st_ch)) = st.v[ps];
      }
      if constexpr (ROPE) __builtin_amdgcn_sched_barrier(0);  // one pass at a time (register pressure)
    }
  };
  // fragment read addresses
  int k_rd[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) k_rd[ks] = k_lds_off(lc, 4 * ks + lg);
  // V^T: the 16 lanes of row group g present rows 4 g + q4, 8-byte pieces p4 of a 32-byte d block
  const int q4 = lc >> 2, p4 = lc & 3;
  const int v_row = 4 * lg + q4;  // second read: + 16 rows (same swizzle key)
  int v_rd[DBLK];
#pragma unroll
  for (int db = 0; db < DBLK; ++db) v_rd[db] = v_row * ROWB + ((db ^ v_slot_key(v_row)) << 5) + p4 * 8;

  f32x4 o_acc[DBLK];
#pragma unroll
  for (int db = 0; db < DBLK; ++db) o_acc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1.0e30f, l_run = 0.f;

  const int n_tok = chunk_end - chunk_start;
  if (n_tok > 0) {
    const int ntiles = (n_tok + kTile - 1) / kTile;
    int pg[NPASS], en[NPASS];
    Stage st;
    fetch_pages(chunk_start, pg, en);
    issue_loads(pg, en, st);
    fetch_pages(chunk_start + kTile, pg, en);
    for (int t = 0; t < ntiles; ++t) {
      const int tile0 = chunk_start + t * kTile;
      if constexpr (ROPE)
        if ((t & 3) == 0) rope_seed(rope_pos0 + tile0 + st_row);
      write_stage(st);  // waits for the tile's loads
      __builtin_amdgcn_wave_barrier();
      if (t + 1 < ntiles) {
        issue_loads(pg, en, st);  // tile t + 1 streams in while tile t is consumed
        fetch_pages(tile0 + 2 * kTile, pg, en);
      }
      // ---- S^T = K Q^T ----
      f32x4 s_acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32x4 a = *(const u32x4*)(kb + k_rd[ks] + h * 16 * ROWB);
          s_acc[h] = Mfma16<T16>::mfma(__builtin_bit_cast(frag_t, a), qf[ks], s_acc[h]);
        }
      }
      if (general) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float t = s_acc[h][j] * p.sm_scale;
            if (p.use_alibi) t += slope * (float)(tile0 + 16 * h + 4 * lg + j);
            if (p.logits_soft_cap > 0.f) t = p.logits_soft_cap * fast_tanh(t * inv_cap);
            s_acc[h][j] = t * kLog2e;
          }
      }
      if (tile0 + kTile > chunk_end || tile0 < win_start) {
        // register j of half h is kv row 16 h + 4 g + j; visible rows are [win_start, chunk_end)
        const int lo = win_start - tile0 - 4 * lg, hi = chunk_end - tile0 - 4 * lg;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = 16 * h + j;
            s_acc[h][j] = (row >= lo && row < hi) ? s_acc[h][j] : -INFINITY;
          }
      }
      // ---- online softmax (base 2) ----
      float mx = fmaxf(fmaxf(fmaxf(s_acc[0][0], s_acc[0][1]), fmaxf(s_acc[0][2], s_acc[0][3])),
                       fmaxf(fmaxf(s_acc[1][0], s_acc[1][1]), fmaxf(s_acc[1][2], s_acc[1][3])));
      mx = reduce_rows<true>(mx);
      const float m_new = fmaxf(m_run, mx * c_log2);
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      float psum = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s_acc[h][j] = fast_exp2(__builtin_fmaf(s_acc[h][j], c_log2, -m_new));
          psum += s_acc[h][j];
        }
      l_run = l_run * alpha + psum;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int db = 0; db < DBLK; ++db) o_acc[db] *= alpha;
      }
      // ---- O^T += V^T P^T; with bf16 the probabilities go through as hi + lo (see decode_mfma_kernel.h) ----
      u32x4 w, wl;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
          const float a = s_acc[h][2 * j2], b = s_acc[h][2 * j2 + 1];
          const uint32_t pk = pack2<T16>(a, b);
          w[2 * h + j2] = pk;
          if constexpr (T16 == FI_DTYPE_BF16) {
            const float ha = __builtin_bit_cast(float, pk << 16);
            const float hb = __builtin_bit_cast(float, pk & 0xffff0000u);
            wl[2 * h + j2] = pack2<T16>(a - ha, b - hb);
          }
        }
      const frag_t pfrag = __builtin_bit_cast(frag_t, w);
#pragma unroll
      for (int db = 0; db < DBLK; ++db) {
        const char* base = vb + v_rd[db];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(base + 16 * ROWB));
        using s16x8 = __attribute__((ext_vector_type(8))) short;
        const s16x8 a8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o_acc[db] = Mfma16<T16>::mfma(__builtin_bit_cast(frag_t, a8), pfrag, o_acc[db]);
        if constexpr (T16 == FI_DTYPE_BF16)
          o_acc[db] = Mfma16<T16>::mfma(__builtin_bit_cast(frag_t, a8), __builtin_bit_cast(frag_t, wl), o_acc[db]);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }

  // ---- finalize and write (partial state or final output) ----
  l_run = reduce_rows<false>(l_run);
  const bool empty = !(l_run > 0.f);
  const float inv = empty ? 0.f : 1.0f / l_run;
  const float lse_v = empty ? FI_NEG_INF : m_run + fast_log2(l_run);
  if (lc < G) {
    const int qo_head = head0 + lc;
    const int64_t out_row = p.split_kv ? (int64_t)(p.o_indptr ? p.o_indptr[req] : 0) + kv_tile : req;
    const int64_t ob = (out_row * p.num_qo_heads + qo_head) * D;
#pragma unroll
    for (int db = 0; db < DBLK; ++db) {
      const int d0 = 16 * db + 4 * lg;
      if (p.split_kv) {
        *(f32x4*)(p.tmp_o + ob + d0) = o_acc[db] * inv;
      } else {
        const uint32_t w0 = pack2<T16>(o_acc[db][0] * inv, o_acc[db][1] * inv);
        const uint32_t w1 = pack2<T16>(o_acc[db][2] * inv, o_acc[db][3] * inv);
        *(u32x2*)((uint16_t*)p.o + ob + d0) = u32x2{w0, w1};
      }
    }
    if (lg == 0) {
      if (p.split_kv) p.tmp_lse[out_row * p.num_qo_heads + qo_head] = lse_v;
      else if (p.lse) p.lse[out_row * p.num_qo_heads + qo_head] = lse_v;
    }
  }
}

}  // namespace fi

// Batch paged-KV decode on the matrix cores, for wide GQA groups (the reference's "tensor core" decode,
// flashinfer/decode.py:1025-1063, which runs the prefill kernel with qo_len = 1).
//
// The VALU decode kernel (decode_kernel.h) spends ~2 G flops of vector ALU per KV byte; above G = 4 it is
// VALU-bound (measured 4.4 TB/s at G = 8).  Here one WAVE still owns one (request kv-chunk, kv head) item of
// the same work list, but the G query heads become the columns of a 32x32x16 MFMA:
//   S^T[32 kv][G..32] = K Q^T,   O^T[D][G..32] += V^T P^T
// K/V tiles of 32 tokens are gathered with coalesced 16-byte loads into registers one tile ahead, written
// to a wave-private LDS region (no workgroup barrier anywhere: LDS operations of one wave execute in
// order), and read back as MFMA fragments (K: swizzled ds_read_b128, V^T: ds_read_b64_tr_b16) exactly as in
// prefill_kernel.h.  Vector ALU work drops to the softmax of a 32 x 32 tile per 16 KB of KV, so the kernel
// is HBM-bound for any group size up to 32.
#pragma once
#include "decode_kernel.h"
#include "prefill_kernel.h"

namespace fi {

constexpr int kDmTileKV = 32;

// T16: q/o (and MFMA) dtype; KVS: storage dtype of the cache (T16, or fp8 upcast to T16 while staging, as
// the reference's decode does with an fp8 cache); PAGED: page table present (false: identity pages).
template <int T16, int KVS, int D, bool PAGED>
__global__ void __launch_bounds__(kDecodeThreads, 2) decode_mfma_kernel(const DecodeKernelParams p) {
  using M = MfmaType<T16>;
  using frag_t = typename M::frag;
  constexpr bool KV_FP8 = (KVS == FI_DTYPE_FP8_E4M3 || KVS == FI_DTYPE_FP8_E5M2);
  constexpr int KV_BYTES = KV_FP8 ? 1 : 2;
  constexpr int ROWB = D * 2;           // bytes per row of the (16-bit) LDS images
  constexpr int CPR = D / 8;            // 16-byte chunks per LDS row
  constexpr int GCH = D * KV_BYTES / 16;  // 16-byte chunks per row in the cache
  constexpr int RPP = 64 / GCH;         // rows staged per pass by one wave
  constexpr int NPASS = kDmTileKV / RPP;
  constexpr int KSTEPS = D / 16;
  constexpr int DBLK = D / 32;
  constexpr int TILE_BYTES = kDmTileKV * ROWB;

  __shared__ __attribute__((aligned(16))) char smem[kDecodeWaves][2 * TILE_BYTES];  // per wave: K | V

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
  const int lq = lane & 31, lh = lane >> 5;

  // ---- item -> (work, kv head, 32-head column block); the work list is the VALU kernel's, with one
  // head tile per 32 query heads of the group (one tile for every group up to 32) ----
  const int col_blk = item % p.head_tiles;
  const int kv_head = (item / p.head_tiles) % p.num_kv_heads;
  const int work = item / (p.head_tiles * p.num_kv_heads);
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
  // sliding window (ref: variants.cuh:78-91 with qo_len = 1): visible iff kv_idx >= kv_len - 1 - window_left.
  // With a planned window the chunks start at the first page that can intersect it (decode.hip); inside its
  // chunk the wave starts at the 32-token tile that holds the window start.
  int chunk_base = 0;
  if (p.plan_window_left >= 0 && p.indptr) {
    const int np = p.indptr[req + 1] - page_begin;
    chunk_base = np > 0 ? max((np - 1) * p.page_size - p.plan_window_left, 0) / p.page_size * p.page_size : 0;
  }
  const int win_start = p.window_left >= 0 ? max(0, kv_len - 1 - p.window_left) : 0;
  const int kv_chunk_size = p.kv_chunk_size_ptr ? *p.kv_chunk_size_ptr : p.kv_chunk_size;
  int chunk_start = chunk_base + (p.split_kv ? kv_tile * kv_chunk_size : 0);
  const int chunk_end = p.split_kv ? min(chunk_start + kv_chunk_size, kv_len) : kv_len;
  if (win_start > chunk_start) chunk_start += (win_start - chunk_start) / kDmTileKV * kDmTileKV;
  const int G = min(p.group_size - 32 * col_blk, 32);  // heads of this column block
  const int head0 = kv_head * p.group_size + 32 * col_blk;
  const int head = head0 + min(lq, G - 1);

  // ---- Q fragments: lane (q = head column, h) holds Q[head][16 ks + 8 h + 0..7] ----
  frag_t qf[KSTEPS];
  {
    const uint16_t* qrow = (const uint16_t*)p.q + (int64_t)req * p.q_stride_n + (int64_t)head * p.q_stride_h;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
      qf[ks] = __builtin_bit_cast(frag_t, *(const u32x4*)(qrow + 16 * ks + 8 * lh));
  }
  const float c_log2 = p.sm_scale * kLog2e;

  // ---- staging: pass ps covers rows ps*RPP + lane/CPR, chunk lane%CPR ----
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
  auto k_lds_off = [&](int row, int ch) -> int {
    const int sw = (CPR >= 16) ? (row & 15) : ((row >> 1) & 7);
    return row * ROWB + ((ch ^ sw) << 4);
  };
  auto v_lds_off = [&](int row, int ch) -> int {
    const int f = (ROWB >= 256) ? (row & 3) : ((row >> 1) & 1);
    return row * ROWB + (((ch >> 2) ^ f) << 6) + ((ch & 3) << 4);
  };
  auto write_stage = [&](const Stage& st) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int row = ps * RPP + st_row;
      if constexpr (KV_FP8) {  // 16 fp8 -> two 16-byte chunks of T16
        *(u32x4*)(kb + k_lds_off(row, 2 * st_ch)) = fp8x8_to_16<T16, KVS>(u32x2{st.k[ps][0], st.k[ps][1]});
        *(u32x4*)(kb + k_lds_off(row, 2 * st_ch + 1)) = fp8x8_to_16<T16, KVS>(u32x2{st.k[ps][2], st.k[ps][3]});
        *(u32x4*)(vb + v_lds_off(row, 2 * st_ch)) = fp8x8_to_16<T16, KVS>(u32x2{st.v[ps][0], st.v[ps][1]});
        *(u32x4*)(vb + v_lds_off(row, 2 * st_ch + 1)) = fp8x8_to_16<T16, KVS>(u32x2{st.v[ps][2], st.v[ps][3]});
      } else {
        *(u32x4*)(kb + k_lds_off(row, st_ch)) = st.k[ps];
        *(u32x4*)(vb + v_lds_off(row, st_ch)) = st.v[ps];
      }
    }
  };
  int k_rd[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) k_rd[ks] = k_lds_off(lq, 2 * ks + lh);
  const int q4 = (lane & 15) >> 2, p4 = lane & 3, gpar = (lane >> 4) & 1;
  int v_rd[DBLK];
#pragma unroll
  for (int db = 0; db < DBLK; ++db) {
    const int row = 4 * lh + q4;
    const int col_byte = (32 * db + 16 * gpar + 4 * p4) * 2;
    const int f = (ROWB >= 256) ? (row & 3) : ((row >> 1) & 1);
    v_rd[db] = row * ROWB + (((col_byte >> 6) ^ f) << 6) + (col_byte & 63);
  }

  f32x16 o_acc[DBLK];
#pragma unroll
  for (int db = 0; db < DBLK; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[db][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  const int n_tok = chunk_end - chunk_start;
  if (n_tok > 0) {
    const int ntiles = (n_tok + kDmTileKV - 1) / kDmTileKV;
    int pg[NPASS], en[NPASS];
    Stage st;
    fetch_pages(chunk_start, pg, en);
    issue_loads(pg, en, st);
    fetch_pages(chunk_start + kDmTileKV, pg, en);
    for (int t = 0; t < ntiles; ++t) {
      const int tile0 = chunk_start + t * kDmTileKV;
      write_stage(st);  // waits for the tile's loads
      __builtin_amdgcn_wave_barrier();
      if (t + 1 < ntiles) {
        issue_loads(pg, en, st);  // tile t+1 streams in while tile t is consumed
        fetch_pages(tile0 + 2 * kDmTileKV, pg, en);
      }
      // ---- S^T = K Q^T ----
      f32x16 s_acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) s_acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const u32x4 a = *(const u32x4*)(kb + k_rd[ks]);
        s_acc = M::mfma(__builtin_bit_cast(frag_t, a), qf[ks], s_acc);
      }
      if (tile0 + kDmTileKV > chunk_end || tile0 < win_start) {
        // tail tile / window-start tile: accumulator register r of lane (q, lh) is kv row
        // 8 (r >> 2) + 4 lh + (r & 3); visible rows are [win_start, chunk_end)
        const int lo = win_start - tile0 - 4 * lh, hi = chunk_end - tile0 - 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2);
          s_acc[r] = (row >= lo && row < hi) ? s_acc[r] : -INFINITY;
        }
      }
      // ---- online softmax (base 2) ----
      float mx = s_acc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s_acc[r]);
      mx = fmaxf(mx, swap_halves(mx));
      const float m_new = fmaxf(m_run, mx * c_log2);
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s_acc[r] = fast_exp2(__builtin_fmaf(s_acc[r], c_log2, -m_new));
        psum += s_acc[r];
      }
      l_run = l_run * alpha + psum;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int db = 0; db < DBLK; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o_acc[db][r] *= alpha;
      }
      // ---- O^T += V^T P^T (accumulator registers 8s..8s+7 are the B operand of k-step s) ----
      // The VALU decode kernel (and the reference's CUDA-core decode, decode.cuh:131-144) accumulates p * v
      // with p in f32.  A bf16 P would lose 8 of its bits, so with bf16 the probabilities go through the
      // matrix pipe as hi + lo = bf16(p) + bf16(p - bf16(p)) (two MFMAs; the pipe is idle most of the time
      // in this HBM-bound kernel) and the result does not depend on which decode kernel was selected.
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u32x4 w, wl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = s_acc[8 * s2 + 2 * j], b = s_acc[8 * s2 + 2 * j + 1];
          w[j] = pack2<T16>(a, b);
          if constexpr (T16 == FI_DTYPE_BF16) {
            const float ha = __builtin_bit_cast(float, w[j] << 16);
            const float hb = __builtin_bit_cast(float, w[j] & 0xffff0000u);
            wl[j] = pack2<T16>(a - ha, b - hb);
          }
        }
        const frag_t pfrag = __builtin_bit_cast(frag_t, w);
#pragma unroll
        for (int db = 0; db < DBLK; ++db) {
          const char* base = vb + (16 * s2) * ROWB + v_rd[db];
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(base + 8 * ROWB));
          using s16x8 = __attribute__((ext_vector_type(8))) short;
          const s16x8 a8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o_acc[db] = M::mfma(__builtin_bit_cast(frag_t, a8), pfrag, o_acc[db]);
          if constexpr (T16 == FI_DTYPE_BF16)
            o_acc[db] = M::mfma(__builtin_bit_cast(frag_t, a8), __builtin_bit_cast(frag_t, wl), o_acc[db]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }

  // ---- finalize and write (partial state or final output) ----
  l_run += swap_halves(l_run);
  const bool empty = !(l_run > 0.f);
  const float inv = empty ? 0.f : 1.0f / l_run;
  const float lse_v = empty ? FI_NEG_INF : m_run + fast_log2(l_run);
  if (lq < G) {
    const int qo_head = head0 + lq;
    const int64_t out_row = p.split_kv ? (int64_t)(p.o_indptr ? p.o_indptr[req] : 0) + kv_tile : req;
    const int64_t ob = (out_row * p.num_qo_heads + qo_head) * D;
#pragma unroll
    for (int db = 0; db < DBLK; ++db) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d0 = 32 * db + 8 * r4 + 4 * lh;
        if (p.split_kv) {
          const f32x4 w = {o_acc[db][4 * r4] * inv, o_acc[db][4 * r4 + 1] * inv, o_acc[db][4 * r4 + 2] * inv,
                           o_acc[db][4 * r4 + 3] * inv};
          *(f32x4*)(p.tmp_o + ob + d0) = w;
        } else {
          const uint32_t w0 = pack2<T16>(o_acc[db][4 * r4] * inv, o_acc[db][4 * r4 + 1] * inv);
          const uint32_t w1 = pack2<T16>(o_acc[db][4 * r4 + 2] * inv, o_acc[db][4 * r4 + 3] * inv);
          *(u32x2*)((uint16_t*)p.o + ob + d0) = u32x2{w0, w1};
        }
      }
    }
    if (lh == 0) {
      if (p.split_kv) p.tmp_lse[out_row * p.num_qo_heads + qo_head] = lse_v;
      else if (p.lse) p.lse[out_row * p.num_qo_heads + qo_head] = lse_v;
    }
  }
}

}  // namespace fi

// fp8-native batch paged prefill for gfx950 (BASELINE config C3): Q, K, V are e4m3 and BOTH contractions
// run on the block-scaled MFMA v_mfma_scale_f32_32x32x64_f8f6f4 with unit (E8M0 = 127) block scales,
// i.e. the plain fp8 product at twice the rate of the 16-bit / non-scaled fp8 MFMA.
//
// Same decomposition as prefill_kernel.h (workgroup = 4 waves = 128 GQA-packed query rows x one kv head,
// 64-row kv tiles, S^T = K Q^T with the query row on the lane, O^T += V^T P^T with the S^T accumulator
// registers as the B operand).  Differences:
//   * K=64 per MFMA: S^T needs 2 MFMAs per 32-row kv block (head_dim 128), O^T needs ONE MFMA per 32-row
//     block of head_dim per 64-row kv tile: 8 MFMAs (512 pipe cycles) per tile and wave instead of 32 (1024).
//   * P is quantised to e4m3 (x448) in registers exactly as the reference does
//     (hopper/variants.cuh:72, 84-90) and used directly as the B operand: lane (q, h) holds the 32
//     probabilities kv = 32 kb + 8 g + 4 h + e  (kb<2, g<4, e<4) in accumulator order.
//   * V has to be the A operand with that same k order along each head_dim row.  The V tile stays
//     ROW-MAJOR in LDS ([64 kv][128 B], staged exactly like K with 16-byte loads and ds_write_b128) and is
//     transposed on the way out by ds_read_b64_tr_b8: in a 16-lane group, lanes 2b and 2b+1 address the
//     16 bytes of "row b" (any row), and lane j receives byte j of rows 0..7 -- so each lane gathers, per
//     read, the 8 kv rows of its k order for its own head_dim column.
//   * both images are XOR-swizzled on 16-byte chunks so that every LDS access is conflict-free.
//   * page gather: one wave per tile resolves (page id, entry) of the 64 kv rows into a byte-offset table
//     in LDS, two tiles ahead; K and V share it.
// Softmax arithmetic (ref hopper/attention_updater.cuh:167-256): row sum from the UNROUNDED probabilities,
// O *= scale_v / 448 / rowsum at the end, lse = m + log2(sum) in base 2.
#pragma once
#include <type_traits>

#include "prefill_kernel.h"

#ifndef FI_PF8_KO
#define FI_PF8_KO 0  // experiments only, bit mask: 1 no K/V loads and LDS stores in the tile loop, 2 no exp2 in the softmax
#endif
namespace fi {

using i32x8 = __attribute__((ext_vector_type(8))) int;
using i32x2 = __attribute__((ext_vector_type(2))) int;

__device__ __forceinline__ f32x16 mfma_fp8_k64(i32x8 a, i32x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, /*A fmt e4m3*/ 0, /*B fmt e4m3*/ 0, 0,
                                                         0x7F7F7F7F, 0, 0x7F7F7F7F);
}

// OUT16: output dtype (FI_DTYPE_F16 / FI_DTYPE_BF16); head_dim 128; page_size % 4 == 0
#ifndef FI_FP8_V1_WAVES_PER_SIMD
#define FI_FP8_V1_WAVES_PER_SIMD 2
#endif
template <int OUT16>
__global__ void __launch_bounds__(kPrefillThreads, FI_FP8_V1_WAVES_PER_SIMD)
    batch_prefill_fp8_v1_kernel(const PrefillKernelParams p) {
  constexpr int D = 128;
  constexpr int K_ROWB = 128;               // bytes per row of the K image (one kv row)
  constexpr int V_ROWB = 128;               // bytes per row of the V image (one kv row)
  constexpr int K_TILE = kTileKV * K_ROWB;  // 8 KB
  constexpr int V_TILE = kTileKV * V_ROWB;  // 8 KB
  constexpr int STAGE = K_TILE + V_TILE;
  constexpr int DBLK = D / 32;

  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  __shared__ uint64_t row_off_tab[4][kTileKV];  // byte offset of every kv row of a tile; slot = tile % 4

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 31;
  const int lh = lane >> 5;

  // ---- (request, q tile, kv head): same mapping as prefill_kernel.h ----
  const int total = p.num_work * p.num_kv_heads;
  int logical;
  {
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int qn = total >> 3, rn = total & 7;
    logical = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
  }
  const int kv_head = logical / p.num_work;
  const int work = logical - kv_head * p.num_work;
  int req = 0, q_tile = work, kv_chunk = 0;
  const bool split = p.kv_tile_indices != nullptr || p.num_kv_chunks > 1;  // see prefill_kernel.h
  if (p.request_indices) {
    req = p.request_indices[work];
    q_tile = p.qo_tile_indices[work];
    if (req < 0) return;
    if (p.kv_tile_indices) kv_chunk = p.kv_tile_indices[work];
  } else if (p.num_kv_chunks > 1) {
    q_tile = work / p.num_kv_chunks;
    kv_chunk = work - q_tile * p.num_kv_chunks;
  }
  int qo_start = 0, qo_len, kv_len, page_begin = 0;
  if (p.qo_indptr) {
    qo_start = p.qo_indptr[req];
    qo_len = p.qo_indptr[req + 1] - qo_start;
  } else {
    qo_len = p.single_qo_len;
  }
  if (p.kv_indptr) {
    page_begin = p.kv_indptr[req];
    const int np = p.kv_indptr[req + 1] - page_begin;
    // ragged KV (no last_page_len): every page is full (ref ragged wrapper: prefill.py:2255-3007)
    kv_len = p.kv_last_page_len ? (np > 0 ? (np - 1) * p.page_size + p.kv_last_page_len[req] : 0)
                                : np * p.page_size;
  } else {
    kv_len = p.single_kv_len;
  }
  const int G = p.group_size;
  const int packed_len = qo_len * G;
  const int row0 = q_tile * kTileQ + wave * 32;
  const int pr = row0 + lq;
  const bool row_valid = pr < packed_len;
  const int prc = row_valid ? pr : (packed_len > 0 ? packed_len - 1 : 0);
  const int qo_idx = (int)fast_div((uint32_t)prc, p.group_div);
  const int hg = prc - qo_idx * G;
  const int qo_head = kv_head * G + hg;
  const int q_pos = kv_len - qo_len + qo_idx;

  // ---- Q fragments: lane (q, h) holds bytes [64 kk + 32 h, +32) of its row ----
  i32x8 qf[2];
  {
    const uint8_t* qrow = (const uint8_t*)p.q + (int64_t)(qo_start + qo_idx) * p.q_stride_n +
                          (int64_t)qo_head * p.q_stride_h;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const u32x4 lo = *(const u32x4*)(qrow + 64 * kk + 32 * lh);
      const u32x4 hi = *(const u32x4*)(qrow + 64 * kk + 32 * lh + 16);
      qf[kk] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    }
  }

  float qk_scale = p.sm_scale;
  if (p.scale_q) qk_scale *= p.scale_q[qo_head];
  if (p.scale_k) qk_scale *= p.scale_k[kv_head];
  const float c_log2 = qk_scale * kLog2e;

  int kv_end = kv_len;
  if (p.causal) {
    const int last_pr = min(q_tile * kTileQ + kTileQ, packed_len) - 1;
    const int last_qo = last_pr >= 0 ? (int)fast_div((uint32_t)last_pr, p.group_div) : 0;
    kv_end = min(kv_len, max(0, kv_len - qo_len + last_qo + 1));
  }
  int kv_begin = 0;
  if (split) {  // split-KV work item (see prefill_kernel.h)
    const int kv_chunk_size = p.kv_chunk_size_ptr ? *p.kv_chunk_size_ptr : p.kv_chunk_size;
    kv_begin = kv_chunk * kv_chunk_size;
    kv_end = min(kv_end, kv_begin + kv_chunk_size);
  }
  const int tile_base = kv_begin / kTileKV;
  const int num_tiles = kv_end > kv_begin ? (kv_end - kv_begin + kTileKV - 1) / kTileKV : 0;
  // a row that sees no key (causal with qo_len > kv_len) gets a range no index can fall into
  const int vis_hi_raw = p.causal ? min(kv_len - 1, q_pos) : kv_len - 1;
  const int vis_lo = vis_hi_raw < 0 ? 0x40000000 : 0;
  const int vis_hi = vis_hi_raw < 0 ? 0x40000000 : vis_hi_raw;
  const int first_qo_wave = (int)fast_div((uint32_t)min(row0, max(packed_len - 1, 0)), p.group_div);
  const int min_qpos_wave = kv_len - qo_len + first_qo_wave;

  // ---- staging geometry: K and V alike, thread -> (row = tid/8 + 32 pass, 16-byte chunk tid%8) ----
  const int k_row = tid >> 3, k_ch = tid & 7;
  const int64_t head_off = (int64_t)kv_head * p.kv_stride_h;
  const char* const k_thr = (const char*)p.k + head_off + k_ch * 16;
  const char* const v_thr = (const char*)p.v + head_off + k_ch * 16;
  const uint32_t stride_page32 = (uint32_t)p.kv_stride_page, stride_n32 = (uint32_t)p.kv_stride_n;
  auto tab_lookup = [&](int tile, int& pg, int& en) {
    const int kvi = max(min(tile * kTileKV + lane, kv_len - 1), 0);
    const int pi = (int)fast_div((uint32_t)kvi, p.page_div);
    en = kvi - pi * p.page_size;
    pg = p.kv_indices ? p.kv_indices[page_begin + pi] : page_begin + pi;
  };
  auto tab_store = [&](int slot, int pg, int en) {
    row_off_tab[slot][lane] = (uint64_t)(uint32_t)pg * stride_page32 + (uint64_t)(uint32_t)en * stride_n32;
  };
  struct Stage {
    u32x4 k[2], v[2];
  };
  auto issue_loads = [&](int slot, Stage& st) {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const uint64_t off = row_off_tab[slot][k_row + 32 * ps];
      st.k[ps] = *(const u32x4*)(k_thr + off);
      st.v[ps] = *(const u32x4*)(v_thr + off);
    }
  };
  auto k_lds_off = [](int row, int ch) { return row * K_ROWB + ((ch ^ ((row >> 1) & 7)) << 4); };
  // V chunk swizzle: the 8 rows of one transposed read are r0 + {0..3, 8..11}; bits 1 and 3 of the row
  // tell the four rows of either parity apart
  auto v_swz = [](int row) { return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2); };
  auto write_stage = [&](int buf, const Stage& st) {
    char* kb = smem + buf * STAGE;
    char* vb = kb + K_TILE;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int row = k_row + 32 * ps;
      *(u32x4*)(kb + k_lds_off(row, k_ch)) = st.k[ps];
      *(u32x4*)(vb + row * V_ROWB + ((k_ch ^ v_swz(row)) << 4)) = st.v[ps];
    }
  };

  // ---- per-lane LDS read offsets ----
  // K fragment: row lq (+32 kb), 16-byte chunks 4 kk + 2 lh + e.  The three chunk bits are disjoint
  // (bit 0: e, bit 1: lh, bit 2: kk), so every address is ONE lane-constant base XOR a literal -- one
  // register instead of four (the kernel sits at the 256-register line; see the spill note below)
  int k_rd_base = k_lds_off(lq, 2 * lh);
  auto k_rd = [&](int kk, int e) { return k_rd_base ^ ((4 * kk + e) << 4); };
  // V^T fragment (A operand): lane (d = 32 db + 16 g + j, lh) needs, as byte p = 8 r + b of its 32 bytes,
  // V[kv(p)][d] with kv(p) = 32 (p >> 4) + 8 ((p & 15) >> 2) + 4 lh + (p & 3) -- the order in which the
  // S^T accumulator registers hold P.  Transposed read r covers p = 8 r .. 8 r + 7: rows
  // 32 (r >> 1) + 16 (r & 1) + 4 lh + {0..3, 8..11}; lane i of the 16-lane group addresses row b = i >> 1,
  // bytes 8 (i & 1) .. +8 of chunk 2 db + g.
  int v_rd_base;  // chunk 2 db + g: db occupies chunk bits 1-2, so v_rd(db) = base ^ (db << 5)
  {
    const int i16 = lane & 15, g = (lane >> 4) & 1, b = i16 >> 1;
    const int row = 4 * lh + (b & 3) + 8 * (b >> 2);
    v_rd_base = row * V_ROWB + ((g ^ v_swz(row)) << 4) + 8 * (i16 & 1);
  }
  auto v_rd = [&](int db) { return v_rd_base ^ (db << 5); };

  f32x16 o_acc[DBLK];
#pragma unroll
  for (int db = 0; db < DBLK; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[db][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  if (num_tiles > 0) {
    Stage st;
    if (wave < 2) {
      int pg0, en0;
      tab_lookup(tile_base + min(wave, num_tiles - 1), pg0, en0);
      tab_store(wave, pg0, en0);
    }
    __syncthreads();
    issue_loads(0, st);
    write_stage(0, st);
    __syncthreads();
    // The next tile is always staged (past the end: the last tile again, into the idle buffer), so there
    // is no branch around a load.  The row-offset table of tile t+2 is produced during tile t by wave t % 4.
    auto tile_body = [&](auto buf_c, const int t) {
      constexpr int buf = decltype(buf_c)::value;
      // keep the derived LDS addresses out of the loop-invariant set: hoisted, they cost four registers
      // each, get spilled, and every reload (a VMEM op) drags a vmcnt(0) wait -- on the K/V loads just
      // issued -- into the MFMA section
      asm volatile("" : "+v"(k_rd_base), "+v"(v_rd_base));
      if (!(FI_PF8_KO & 1)) issue_loads((t + 1) & 3, st);
      const bool tab_wave = wave == (t & 3);
      int tab_pg = 0, tab_en = 0;
      if (tab_wave) tab_lookup(tile_base + min(t + 2, num_tiles - 1), tab_pg, tab_en);
      const char* kb = smem + buf * STAGE;
      const char* vb = kb + K_TILE;
      const int tile0 = (tile_base + t) * kTileKV;

      // ---- S^T = K Q^T: 2 kv blocks x 2 k-steps of 64 ----
      f32x16 s_acc[2];
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[kbk][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const u32x4 lo = *(const u32x4*)(kb + kbk * 32 * K_ROWB + k_rd(kk, 0));
          const u32x4 hi = *(const u32x4*)(kb + kbk * 32 * K_ROWB + k_rd(kk, 1));
          const i32x8 a = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
          s_acc[kbk] = mfma_fp8_k64(a, qf[kk], s_acc[kbk]);
        }
      }

      const bool need_mask = (tile0 + kTileKV > kv_len) || (p.causal && tile0 + kTileKV - 1 > min_qpos_wave);
      if (need_mask) {
        const unsigned span = (unsigned)(vis_hi - vis_lo);
        const int base_idx = tile0 + 4 * lh - vis_lo;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned rel = (unsigned)(base_idx + 32 * kbk + (r & 3) + 8 * (r >> 2));
            s_acc[kbk][r] = rel <= span ? s_acc[kbk][r] : -INFINITY;
          }
      }

      // ---- online softmax (base 2) ----
      float mx = s_acc[0][0];
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[kbk][r]);
      mx = fmaxf(mx, swap_halves(mx));
      const float m_new = fmaxf(m_run, mx * c_log2);
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      // p * 448 = 2^(s c - m + log2 448): the e4m3 scale is folded into the exponent; the row sum is taken
      // from these unrounded values and divided by 448 once at the end
      const float m_adj = m_new - 8.807354922057604f;  // log2(448)
      float psum = 0.f;
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s_acc[kbk][r] = (FI_PF8_KO & 2) ? __builtin_fmaf(s_acc[kbk][r], c_log2, -m_adj)
                                          : fast_exp2(__builtin_fmaf(s_acc[kbk][r], c_log2, -m_adj));
          psum += s_acc[kbk][r];
        }
      l_run = l_run * alpha + psum;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int db = 0; db < DBLK; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o_acc[db][r] *= alpha;
      }

      // ---- P -> e4m3, the B operand (32 bytes per lane, accumulator order) ----
      i32x8 p8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kbk = j >> 2, r = 4 * (j & 3);
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(s_acc[kbk][r], s_acc[kbk][r + 1], 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(s_acc[kbk][r + 2], s_acc[kbk][r + 3], w, true);
        p8[j] = w;
      }

      // ---- O^T += V^T P^T: one K=64 MFMA per 32 rows of head_dim, A gathered by 4 transposed reads ----
#pragma unroll
      for (int db = 0; db < DBLK; ++db) {
        i32x8 a;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const i32x2 w = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
              (__attribute__((address_space(3))) i32x2*)(vb + (32 * (r >> 1) + 16 * (r & 1)) * V_ROWB + v_rd(db)));
          a[2 * r] = w[0];
          a[2 * r + 1] = w[1];
        }
        o_acc[db] = mfma_fp8_k64(a, p8, o_acc[db]);
      }

      if (!(FI_PF8_KO & 1)) write_stage(buf ^ 1, st);
      if (tab_wave) tab_store((t + 2) & 3, tab_pg, tab_en);
      __syncthreads();
    };
    int t = 0;
    for (; t + 1 < num_tiles; t += 2) {
      tile_body(std::integral_constant<int, 0>{}, t);
      tile_body(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < num_tiles) tile_body(std::integral_constant<int, 0>{}, t);
  }

  // ---- finalize: l_run carries the x448 of P, so O / l_run is already free of it ----
  l_run += swap_halves(l_run);
  const bool empty = !(l_run > 0.f);
  float inv = empty ? 0.f : 1.0f / l_run;
  if (p.scale_v) inv *= p.scale_v[kv_head];
  if (row_valid && split) {
    const int64_t entry = p.merge_indptr ? (int64_t)p.merge_indptr[qo_start + qo_idx] + kv_chunk
                                         : (int64_t)(qo_start + qo_idx) * p.num_kv_chunks + kv_chunk;
    const int64_t ob = (entry * p.num_qo_heads + qo_head) * D;
#pragma unroll
    for (int db = 0; db < DBLK; ++db) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d0 = 32 * db + 8 * r4 + 4 * lh;
        *(f32x4*)(p.tmp_o + ob + d0) = f32x4{o_acc[db][4 * r4 + 0] * inv, o_acc[db][4 * r4 + 1] * inv,
                                             o_acc[db][4 * r4 + 2] * inv, o_acc[db][4 * r4 + 3] * inv};
      }
    }
    if (lh == 0)
      p.tmp_lse[entry * p.num_qo_heads + qo_head] =
          empty ? FI_NEG_INF : m_run + fast_log2(l_run) - 8.807354922057604f;
  } else if (row_valid) {
    const int64_t ob = ((int64_t)(qo_start + qo_idx) * p.num_qo_heads + qo_head) * D;
#pragma unroll
    for (int db = 0; db < DBLK; ++db) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d0 = 32 * db + 8 * r4 + 4 * lh;
        const uint32_t w0 = pack2<OUT16>(o_acc[db][4 * r4 + 0] * inv, o_acc[db][4 * r4 + 1] * inv);
        const uint32_t w1 = pack2<OUT16>(o_acc[db][4 * r4 + 2] * inv, o_acc[db][4 * r4 + 3] * inv);
        *(u32x2*)((uint16_t*)p.o + ob + d0) = u32x2{w0, w1};
      }
    }
    if (p.lse && lh == 0)
      p.lse[(int64_t)(qo_start + qo_idx) * p.num_qo_heads + qo_head] =
          empty ? FI_NEG_INF : m_run + fast_log2(l_run) - 8.807354922057604f;
  }
}

}  // namespace fi

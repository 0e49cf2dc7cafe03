// Batch paged-KV decode attention for gfx950 (wave64), HBM-bound flash-decoding.
//
// Work decomposition (what replaces the reference's grid (padded_batch, num_kv_heads) x block
// (HEAD_DIM/vec, GROUP, bdz), ref: include/flashinfer/attention/decode.cuh:739-825):
//   one WAVE owns one (request kv-chunk, kv head, q-head tile) item and streams that chunk's K and V
//   rows straight from HBM into VGPRs with 16-byte loads -- no LDS, no barriers.  A 64-lane load
//   instruction covers TPL = 64/LPT tokens, LPT = HEAD_DIM/VEC lanes per token (VEC = 16 B of KV),
//   so every quad of lanes reads 64 contiguous bytes and every token row is read as whole lines.
//   The GT (<= 8) query heads of the GQA group share each K/V register tile ("head-query fusion",
//   ref: decode.cuh:757, 539-545).  Each token-row of lanes keeps its own online-softmax state
//   (m, d, o[GT][VEC]) in base 2 (ref: attention/state.cuh:29-78); rows are merged with shuffles at
//   the end of the chunk.  Page indirection: one page id per load (scalar when page_size is a power of
//   two >= TPL), fetched a tile ahead so HBM loads issue back-to-back.
//
// Math follows ref decode.cuh:62-116 (compute_qk), 131-144 (update_local_state), variants.cuh:31-92.
#pragma once
#include "common.h"

namespace fi {

#ifndef FI_DECODE_XCD_REMAP
#define FI_DECODE_XCD_REMAP 1
#endif
constexpr int kDecodeThreads = 256;  // 4 waves; waves are independent
constexpr int kDecodeWaves = kDecodeThreads / 64;
constexpr float kMInit = -1.0e30f;  // finite "minus infinity" for the running max

struct DecodeKernelParams {
  const void* q;
  void* o;
  float* lse;
  float* tmp_o;  // split-KV partial outputs (f32, normalised) [num_partials, Hq, D]
  float* tmp_lse;
  const void* k;
  const void* v;
  const int32_t* indptr;
  const int32_t* indices;
  const int32_t* last_page_len;
  const int32_t* request_indices;
  const int32_t* kv_tile_indices;
  const int32_t* o_indptr;
  const uint8_t* block_valid_mask;
  const int32_t* q_rope_offset;
  const int32_t* kv_rope_pos_offset;
  const float* alibi_slopes;
  int64_t q_stride_n, q_stride_h;
  int64_t kv_stride_page, kv_stride_n, kv_stride_h;
  int32_t num_items;  // num_work * num_kv_heads * head_tiles
  int32_t num_qo_heads, num_kv_heads, group_size, head_tiles;
  int32_t page_size, log2_page_size;
  int32_t uniform_page;  // page_size is a power of two >= tokens per load: one page id per load
  int32_t fast_path;     // launch the FAST instantiation
  FastDiv page_div;
  int32_t kv_chunk_size;  // tokens; only read when split_kv and kv_chunk_size_ptr is null
  // device copy of the chunk size in the int workspace: plan() rewrites it, so a captured run() replayed
  // after a new plan() sees the new value (ref: *kv_chunk_size_ptr, decode.cuh:424, 926)
  const int32_t* kv_chunk_size_ptr;
  int32_t split_kv;
  int32_t single_kv_len;  // used when indptr == nullptr
  int32_t window_left;    // < 0: off
  int32_t plan_window_left;  // window the planner cut the chunks for (< 0: chunks start at token 0)
  int32_t q_dtype;
  int32_t use_alibi;
  float logits_soft_cap;  // 0: off
  float sm_scale;
  float rope_rcp_scale, rope_rcp_theta;
};

// K/V rows are read once per wave: non-temporal.  When a GQA group is processed as several head tiles the
// neighbouring waves stream the SAME rows, so those loads stay temporal and the partner hits in L2.
template <bool NT>
__device__ __forceinline__ u32x4 load16(const void* base, int64_t byte_off) {
  if constexpr (NT) return __builtin_nontemporal_load((const u32x4*)((const char*)base + byte_off));
  else return *(const u32x4*)((const char*)base + byte_off);
}
// uniform base + 32-bit per-lane byte offset (global_load ... saddr form)
template <bool NT>
__device__ __forceinline__ u32x4 load16(const char* ubase, uint32_t lane_byte_off) {
  if constexpr (NT) return __builtin_nontemporal_load((const u32x4*)(ubase + lane_byte_off));
  else return *(const u32x4*)(ubase + lane_byte_off);
}

// NT: non-temporal K/V loads (false when several head tiles stream the same rows)
template <int KV_DT, int HEAD_DIM, int GT, bool ROPE, bool FAST, int NLOAD, bool NT = true>
struct DecodeWave {
  using T = KVTraits<KV_DT>;
  static constexpr int VEC = T::VEC;
  static constexpr int LPT = HEAD_DIM / VEC;  // lanes per token
  static constexpr int TPL = 64 / LPT;        // tokens per load instruction
  static constexpr int TILE = NLOAD * TPL;    // tokens per tile
  static_assert(LPT >= 1 && LPT <= 64 && (64 % LPT) == 0, "bad head_dim");

  const DecodeKernelParams& p;
  int lane, c, r;
  int kv_head, page_begin, kv_len, chunk_start, chunk_end, win_start;
  int64_t head_off;      // kv_head*stride_h
  uint32_t lane_voff;    // (r*stride_n + c*VEC) * BYTES: per-lane byte offset inside a load (FAST)
  int rope_pos0;         // position offset of kv token 0
  float q[GT][VEC];
  float o[GT][VEC];
  float m[GT], d[GT];
  float slope_l2[GT];
  float s_scale;  // multiplies the (pre-scaled) dot product; != 1 only with soft cap
  // rope state
  float freq[ROPE ? VEC : 1];
  float rc[ROPE ? VEC : 1], rs[ROPE ? VEC : 1];    // cos/sin at the next load's position
  float dc[ROPE ? VEC : 1], ds[ROPE ? VEC : 1];    // cos/sin of TPL*freq (advance per load)

  __device__ __forceinline__ DecodeWave(const DecodeKernelParams& params) : p(params) {}

  struct Buf {
    u32x4 k[NLOAD];
    u32x4 v[NLOAD];
  };

  // ---- page ids for one tile -------------------------------------------------------------
  // FAST (page_size a power of two >= TPL, real page table, full tiles): one SCALAR page id per load,
  // so the load address is {scalar page base} + {per-lane constant offset} and costs no vector ALU.
  // Generic: one page id per lane per load, token index clamped to the chunk end.
  template <bool SCALAR>
  __device__ __forceinline__ void fetch_pages(int tile_tok0, int (&pg)[NLOAD]) const {
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
      const int tok0 = tile_tok0 + j * TPL;
      if constexpr (SCALAR) {
        const int pi = min(tok0, chunk_end - 1) >> p.log2_page_size;
        pg[j] = p.indices[page_begin + pi];
      } else {
        const int tok = min(tok0 + r, chunk_end - 1);
        const int pi = (int)fast_div((uint32_t)tok, p.page_div);
        pg[j] = p.indices ? p.indices[page_begin + pi] : pi;
      }
    }
  }

  template <bool SCALAR>
  __device__ __forceinline__ void issue_loads(int tile_tok0, const int (&pg)[NLOAD], Buf& b) const {
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
      const int tok0 = tile_tok0 + j * TPL;
      if constexpr (SCALAR) {
        const int page = __builtin_amdgcn_readfirstlane(pg[j]);
        const int entry0 = tok0 & (p.page_size - 1);
        const int64_t sbase =
            ((int64_t)page * p.kv_stride_page + head_off + (int64_t)entry0 * p.kv_stride_n) * T::BYTES;
        b.k[j] = load16<NT>((const char*)p.k + sbase, lane_voff);
        b.v[j] = load16<NT>((const char*)p.v + sbase, lane_voff);
      } else {
        const int tok = min(tok0 + r, chunk_end - 1);
        const int pi = (int)fast_div((uint32_t)tok, p.page_div);
        const int entry = tok - pi * p.page_size;
        const int64_t off = (int64_t)pg[j] * p.kv_stride_page + head_off +
                            (int64_t)entry * p.kv_stride_n + (int64_t)(c * VEC);
        b.k[j] = load16<NT>(p.k, off * T::BYTES);
        b.v[j] = load16<NT>(p.v, off * T::BYTES);
      }
    }
  }

  // 16-bit query words -> f32 (select instead of branch so that loads stay in flight)
  static __device__ __forceinline__ void unpack_q(const u32x4 (&raw)[VEC / 8], bool is_bf16,
                                                  float (&out)[VEC]) {
#pragma unroll
    for (int v8 = 0; v8 < VEC / 8; ++v8) {
      float a[8], b[8];
      KVTraits<FI_DTYPE_BF16>::unpack(raw[v8], a);
      KVTraits<FI_DTYPE_F16>::unpack(raw[v8], b);
#pragma unroll
      for (int i = 0; i < 8; ++i) out[8 * v8 + i] = is_bf16 ? a[i] : b[i];
    }
  }

  // ---- rope helpers ------------------------------------------------------------------------
  __device__ __forceinline__ void rope_seed(int pos) {
    if constexpr (ROPE) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float e = (float)pos * freq[i];
        float sn, cs;
        sincos_ool(e, &sn, &cs);
        rc[i] = cs;
        rs[i] = sn;
      }
    }
  }
  __device__ __forceinline__ void rope_advance() {
    if constexpr (ROPE) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float cn = rc[i] * dc[i] - rs[i] * ds[i];
        float sn = rs[i] * dc[i] + rc[i] * ds[i];
        rc[i] = cn;
        rs[i] = sn;
      }
    }
  }

  // ---- one tile of online softmax ----------------------------------------------------------
  template <bool MASKED>
  __device__ __forceinline__ void compute(int tile_tok0, const Buf& b) {
    float s[NLOAD][GT];
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
      float kf[VEC];
      T::unpack(b.k[j], kf);
      if constexpr (ROPE) {
        // non-interleaved rotation: dims i and i+D/2 pair up (ref: pos_enc.cuh:78-101); the partner
        // sits LPT/2 lanes away in the same token row.
        float kr[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          float partner = lane_xor<LPT / 2>(kf[i]);
          kr[i] = kf[i] * rc[i] + ((c < LPT / 2) ? -partner : partner) * rs[i];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) kf[i] = kr[i];
        rope_advance();
      }
      const int tok = tile_tok0 + j * TPL + r;
#pragma unroll
      for (int g = 0; g < GT; ++g) {
        f32x2 acc2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < VEC; i += 2) {
          f32x2 qq = {q[g][i], q[g][i + 1]};
          f32x2 kk = {kf[i], kf[i + 1]};
          acc2 = __builtin_elementwise_fma(qq, kk, acc2);
        }
        float acc = group_sum<LPT>(acc2[0] + acc2[1]);
        if constexpr (!FAST) {
          if (p.use_alibi) acc += slope_l2[g] * (float)tok;  // ref: variants.cuh:67-70 (qo_idx = 0)
          if (p.logits_soft_cap > 0.f) acc = fast_tanh(acc) * s_scale;  // ref: variants.cuh:71-73
        }
        if constexpr (MASKED) {
          bool valid = (tok < chunk_end) && (tok >= win_start);
          acc = valid ? acc : -INFINITY;
        }
        s[j][g] = acc;
      }
    }
    // running max / rescale (ref: decode.cuh:103-115)
    float alpha[GT];
    bool grow = false;
#pragma unroll
    for (int g = 0; g < GT; ++g) {
      float mx = s[0][g];
#pragma unroll
      for (int j = 1; j < NLOAD; ++j) mx = fmaxf(mx, s[j][g]);
      float m_new = fmaxf(m[g], mx);
      grow |= (m_new > m[g]);
      alpha[g] = fast_exp2(m[g] - m_new);
      m[g] = m_new;
      float dsum = 0.f;
#pragma unroll
      for (int j = 0; j < NLOAD; ++j) {
        s[j][g] = fast_exp2(s[j][g] - m_new);
        dsum += s[j][g];
      }
      d[g] = d[g] * alpha[g] + dsum;
    }
    if (__any(grow)) {
#pragma unroll
      for (int g = 0; g < GT; ++g)
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[g][i] *= alpha[g];
    }
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
      float vf[VEC];
      T::unpack(b.v[j], vf);
#pragma unroll
      for (int g = 0; g < GT; ++g) {
#pragma unroll
        for (int i = 0; i < VEC; i += 2) {
          f32x2 vv = {vf[i], vf[i + 1]};
          f32x2 oo = {o[g][i], o[g][i + 1]};
          f32x2 pp = {s[j][g], s[j][g]};
          oo = __builtin_elementwise_fma(pp, vv, oo);
          o[g][i] = oo[0];
          o[g][i + 1] = oo[1];
        }
      }
    }
  }

  __device__ __forceinline__ void run(int item) {
    lane = threadIdx.x & 63;
    c = lane % LPT;
    r = lane / LPT;
    // item -> (work, kv_head, head tile); kv_head fastest so that the waves of a workgroup read
    // neighbouring head rows of the same tokens.
    const int ht = item % p.head_tiles;
    const int rem = item / p.head_tiles;
    kv_head = rem % p.num_kv_heads;
    const int work = rem / p.num_kv_heads;
    int req = 0, kv_tile = work;
    if (p.request_indices) {
      if (p.block_valid_mask && !p.block_valid_mask[work]) return;
      req = p.request_indices[work];
      kv_tile = p.kv_tile_indices[work];
    }
    if (p.indptr) {
      page_begin = p.indptr[req];
      const int np = p.indptr[req + 1] - page_begin;
      kv_len = np > 0 ? (np - 1) * p.page_size + p.last_page_len[req] : 0;  // ref: page.cuh:147-152
    } else {
      page_begin = 0;
      kv_len = p.single_kv_len;
    }
    // with a planned sliding window the chunks start at the first page that can intersect it
    int chunk_base = 0;
    if (p.plan_window_left >= 0 && p.indptr) {
      const int np = p.indptr[req + 1] - page_begin;
      chunk_base = np > 0 ? max((np - 1) * p.page_size - p.plan_window_left, 0) / p.page_size * p.page_size : 0;
    }
    const int kv_chunk_size = p.kv_chunk_size_ptr ? *p.kv_chunk_size_ptr : p.kv_chunk_size;
    chunk_start = chunk_base + (p.split_kv ? kv_tile * kv_chunk_size : 0);
    chunk_end = p.split_kv ? min(chunk_start + kv_chunk_size, kv_len) : kv_len;
    // sliding window (ref: variants.cuh:78-91 with qo_len = 1, qo_idx = 0):
    //   visible iff kv_idx + 1 + window_left >= kv_len
    win_start = p.window_left >= 0 ? max(0, kv_len - 1 - p.window_left) : 0;
    head_off = (int64_t)kv_head * p.kv_stride_h;
    lane_voff = (uint32_t)(((int64_t)r * p.kv_stride_n + c * VEC) * T::BYTES);
    rope_pos0 = p.kv_rope_pos_offset ? p.kv_rope_pos_offset[req] : 0;

    // ---- q: load, (rope), pre-scale ----
    const bool soft_cap = p.logits_soft_cap > 0.f;
    // without soft cap: s = (q.k) * sm_scale*log2e; alibi: s = (q.k*sm_scale + slope*pos)*log2e
    // with soft cap:    s = tanh(q.k * sm_scale/cap) * cap*log2e      (ref: variants.cuh:47-53)
    const float q_scale = soft_cap ? p.sm_scale / p.logits_soft_cap : p.sm_scale * kLog2e;
    s_scale = soft_cap ? p.logits_soft_cap * kLog2e : 1.0f;
    if constexpr (ROPE) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        // ref: decode.cuh:455-459
        freq[i] = p.rope_rcp_scale *
                  __powf(p.rope_rcp_theta,
                         (float)(2 * ((c * VEC + i) % (HEAD_DIM / 2))) / (float)HEAD_DIM);
        float sn, cs;
        sincos_ool((float)TPL * freq[i], &sn, &cs);
        dc[i] = cs;
        ds[i] = sn;
      }
    }
    // all q loads are issued before the first one is consumed
    u32x4 qraw[GT][VEC / 8], qraw_p[ROPE ? GT : 1][VEC / 8];
    const int cp = (c + LPT / 2) % LPT;
#pragma unroll
    for (int g = 0; g < GT; ++g) {
      const int head = kv_head * p.group_size + min(ht * GT + g, p.group_size - 1);
      const int64_t qb = (int64_t)req * p.q_stride_n + (int64_t)head * p.q_stride_h;
#pragma unroll
      for (int v8 = 0; v8 < VEC / 8; ++v8) {
        qraw[g][v8] = *(const u32x4*)((const uint16_t*)p.q + qb + c * VEC + 8 * v8);
        if constexpr (ROPE)
          qraw_p[g][v8] = *(const u32x4*)((const uint16_t*)p.q + qb + cp * VEC + 8 * v8);
      }
    }
    const bool q_is_bf16 = p.q_dtype == FI_DTYPE_BF16;
#pragma unroll
    for (int g = 0; g < GT; ++g) {
      const int head = kv_head * p.group_size + min(ht * GT + g, p.group_size - 1);
      float qv[VEC];
      unpack_q(qraw[g], q_is_bf16, qv);
      if constexpr (ROPE) {
        const int q_pos = p.q_rope_offset ? p.q_rope_offset[req] : (kv_len - 1);
        float partner[VEC];
        unpack_q(qraw_p[g], q_is_bf16, partner);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          float e = (float)q_pos * freq[i];
          float sn, cs;
          sincos_ool(e, &sn, &cs);
          qv[i] = qv[i] * cs + ((c < LPT / 2) ? -partner[i] : partner[i]) * sn;
        }
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        q[g][i] = qv[i] * q_scale;
        o[g][i] = 0.f;
      }
      m[g] = kMInit;
      d[g] = 0.f;
      slope_l2[g] = p.use_alibi ? p.alibi_slopes[head] * kLog2e : 0.f;
    }

    // ---- stream the chunk ----
    // sliding window: start at the tile that holds the window start; only that tile needs the window
    // mask (everything after it is visible), so it is the one masked tile ahead of the pipelined loop
    int first = chunk_start;
    bool lead_masked = false;
    if (p.window_left >= 0 && win_start > chunk_start) {
      first = chunk_start + (win_start - chunk_start) / TILE * TILE;
      lead_masked = win_start > first;
    }
    int pgA[NLOAD], pgB[NLOAD];
    Buf A, B;
    if (lead_masked && first < chunk_end) {
      fetch_pages<false>(first, pgA);
      issue_loads<false>(first, pgA, A);
      if constexpr (ROPE) rope_seed(rope_pos0 + first + r);
      compute<true>(first, A);
      first += TILE;
    }
    const int n_tok = chunk_end - first;
    if (n_tok > 0) {
      const int ntot = (n_tok + TILE - 1) / TILE;
      const int nfull = n_tok / TILE;
      if constexpr (ROPE) rope_seed(rope_pos0 + first + r);
      int t = 0;
      if (nfull > 0) {
        // Software pipeline over the full tiles: while tile t is consumed from one register buffer,
        // tile t+1 streams into the other and the page ids of tile t+2 are fetched.  The steady-state
        // loop has NO branch around a load, so the compiler's vmcnt bookkeeping stays exact (a
        // conditional prefetch makes it wait for the newest loads: measured r1).
        fetch_pages<FAST>(first, pgA);
        fetch_pages<FAST>(first + TILE, pgB);
        issue_loads<FAST>(first, pgA, A);
        while (t + 2 < nfull) {
          issue_loads<FAST>(first + (t + 1) * TILE, pgB, B);
          fetch_pages<FAST>(first + (t + 2) * TILE, pgA);
          if constexpr (ROPE)
            if ((t & 15) == 0 && t) rope_seed(rope_pos0 + first + t * TILE + r);
          compute<false>(first + t * TILE, A);
          issue_loads<FAST>(first + (t + 2) * TILE, pgA, A);
          fetch_pages<FAST>(first + (t + 3) * TILE, pgB);
          compute<false>(first + (t + 1) * TILE, B);
          t += 2;
        }
        if (nfull - t == 2) {
          issue_loads<FAST>(first + (t + 1) * TILE, pgB, B);
          compute<false>(first + t * TILE, A);
          compute<false>(first + (t + 1) * TILE, B);
          t += 2;
        } else {
          compute<false>(first + t * TILE, A);
          t += 1;
        }
      }
      // the partial last tile
      for (; t < ntot; ++t) {
        fetch_pages<false>(first + t * TILE, pgA);
        issue_loads<false>(first + t * TILE, pgA, A);
        if constexpr (ROPE) rope_seed(rope_pos0 + first + t * TILE + r);
        compute<true>(first + t * TILE, A);
      }
    }

    // ---- merge the TPL token rows of the wave (ref: sync_state, decode.cuh:155-189) ----
#pragma unroll
    for (int off = LPT; off < 64; off <<= 1) {
#pragma unroll
      for (int g = 0; g < GT; ++g) {
        float m_o = __shfl_xor(m[g], off, 64);
        float d_o = __shfl_xor(d[g], off, 64);
        float mm = fmaxf(m[g], m_o);
        float a = fast_exp2(m[g] - mm), bsc = fast_exp2(m_o - mm);
        d[g] = d[g] * a + d_o * bsc;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          float o_o = __shfl_xor(o[g][i], off, 64);
          o[g][i] = o[g][i] * a + o_o * bsc;
        }
        m[g] = mm;
      }
    }

    // ---- normalise and write (ref: variant_helper.cuh:81-84, state.cuh:45) ----
    if (r == 0) {
      int64_t out_row;
      if (p.split_kv)
        out_row = (int64_t)(p.o_indptr ? p.o_indptr[req] : 0) + kv_tile;
      else
        out_row = req;
#pragma unroll
      for (int g = 0; g < GT; ++g) {
        const int hg = ht * GT + g;
        if (hg >= p.group_size) continue;
        const int head = kv_head * p.group_size + hg;
        const bool empty = !(d[g] > 0.f);
        const float inv = empty ? 0.f : 1.0f / d[g];
        const float lse_v = empty ? FI_NEG_INF : m[g] + fast_log2(d[g]);
        const int64_t ob = (out_row * p.num_qo_heads + head) * HEAD_DIM + c * VEC;
        if (p.split_kv) {
#pragma unroll
          for (int i = 0; i < VEC; i += 4) {
            f32x4 w = {o[g][i] * inv, o[g][i + 1] * inv, o[g][i + 2] * inv, o[g][i + 3] * inv};
            *(f32x4*)(p.tmp_o + ob + i) = w;
          }
          if (c == 0) p.tmp_lse[out_row * p.num_qo_heads + head] = lse_v;
        } else {
#pragma unroll
          for (int i = 0; i < VEC; i += 8) {
            u32x4 w;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
              uint32_t lo = f32_to_16bit(o[g][i + 2 * k2] * inv, p.q_dtype);
              uint32_t hi = f32_to_16bit(o[g][i + 2 * k2 + 1] * inv, p.q_dtype);
              w[k2] = lo | (hi << 16);
            }
            *(u32x4*)((uint16_t*)p.o + ob + i) = w;
          }
          if (c == 0 && p.lse) p.lse[out_row * p.num_qo_heads + head] = lse_v;
        }
      }
    }
  }
};

template <int KV_DT, int HEAD_DIM, int GT, bool ROPE, bool FAST, int NLOAD, bool NT = true>
__global__ void __launch_bounds__(kDecodeThreads, 2)
    batch_decode_kernel(const DecodeKernelParams p) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // XCD-contiguous logical block id (blocks b and b+8 share an XCD): the workgroups that read the other kv
  // heads of the same pages run on the same XCD at about the same time
  int lb;
  {
    const int b = blockIdx.x, total = gridDim.x;
    const int xcd = b & 7, slot = b >> 3;
    const int qn = total >> 3, rn = total & 7;
    lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
  }
  if (FI_DECODE_XCD_REMAP == 0) lb = blockIdx.x;
  const int item = lb * kDecodeWaves + wave;
  if (item >= p.num_items) return;
  DecodeWave<KV_DT, HEAD_DIM, GT, ROPE, FAST, NLOAD, NT> w(p);
  w.run(item);
}

}  // namespace fi

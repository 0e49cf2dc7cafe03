// Batch paged-KV prefill (flash) attention for gfx950: MFMA 32x32x16, wave64, LDS-staged K/V tiles.
//
// What it replaces: the reference's FA2 `BatchPrefillWithPagedKVCacheDevice`
// (include/flashinfer/attention/prefill.cuh:2024-2413, mma.sync m16n8k16 tiles) and, for fp8 inputs,
// the FA3 path (hopper/quantization/mainloop_mma.cuh:21-237; semantics hopper/variants.cuh:64-102).
//
// Structure (one workgroup = 4 waves = 128 GQA-packed query rows x one kv head):
//   * GQA head-query fusion: packed row pr = qo_idx * G + head_in_group (ref: prefill.cuh:444-448), so
//     the G query heads of a kv head share every K/V tile.
//   * S^T = K . Q^T  ("swapped" product): A = K fragment read from LDS (ds_read_b128 of a swizzled
//     [64 kv][D] image), B = Q fragment held in registers for the whole kernel.  The 32x32 accumulator
//     then has the query row on the LANE and 16 kv positions in registers, so the online-softmax row
//     reductions are in-lane plus one lane<->lane+32 exchange.
//   * O^T += V^T . P^T: the S^T accumulator registers 8s..8s+7, rounded to 16 bit, ARE the B fragment of
//     k-step s (no LDS round trip for P); V^T fragments come from the row-major V image in LDS through
//     ds_read_b64_tr_b16 (hardware transpose), in the k order the accumulator layout dictates.
//   * K/V tiles (64 kv rows) are gathered page by page with 16-byte loads into registers while the
//     previous tile is being consumed, then written to the other LDS buffer (one barrier per tile); page
//     ids are fetched two tiles ahead.
//   * fp8 K/V/Q are upcast to the 16-bit compute type when staged (exact), so one MFMA pipeline serves
//     fp16 / bf16 / fp8-KV / fp8-QKV; with fp8 Q the probabilities are rounded through e4m3 (x448) before
//     P.V exactly as the reference does (hopper/variants.cuh:72, 84-90).
#pragma once
#include <type_traits>

#include "common.h"

namespace fi {

constexpr int kPrefillThreads = 256;
constexpr int kPrefillWaves = 4;
constexpr int kTileQ = 128;   // packed query rows per workgroup
constexpr int kTileKV = 64;   // kv rows per LDS tile
constexpr int kQkPrefetch = 4;  // K fragments in flight ahead of the QK^T MFMA chain

struct PrefillKernelParams {
  const void* q;
  void* o;
  float* lse;
  const void* k;
  const void* v;
  const int32_t* qo_indptr;      // NULL: single request with single_qo_len rows
  const int32_t* kv_indptr;      // NULL: single request, identity pages, single_kv_len
  const int32_t* kv_indices;
  const int32_t* kv_last_page_len;
  const int32_t* request_indices;  // work list (NULL: request 0, tile = work index)
  const int32_t* qo_tile_indices;
  // split-KV (ref: scheduler.cuh:495-614): work item = (request, q tile, kv chunk); partial states go to
  // tmp_o / tmp_lse at entry merge_indptr[qo row] + kv chunk and are folded by the n-way merge kernel
  const int32_t* kv_tile_indices;  // NULL: no split
  const int32_t* merge_indptr;     // [total qo rows + 1]
  float* tmp_o;
  float* tmp_lse;
  int32_t kv_chunk_size;           // tokens, a multiple of the 64-row kv tile (used when the pointer is null)
  // device copy of the chunk size in the int workspace, rewritten by every plan(): a captured run()
  // replayed after a re-plan reads the current value (ref: *kv_chunk_size_ptr, prefill.cuh:2058)
  const int32_t* kv_chunk_size_ptr;
  int32_t num_kv_chunks;           // single-request split (no work list): work = q tile * chunks + chunk
  const float* alibi_slopes;
  const float* scale_q;  // fp8: per qo head / kv head scales (NULL = 1)
  const float* scale_k;
  const float* scale_v;
  const uint8_t* custom_mask;    // packed bits (mask mode CUSTOM), NULL otherwise
  const int32_t* mask_indptr;    // per-request byte offsets into custom_mask (NULL: 0)
  // multi-item scoring (mask mode MULTIITEMSCORING, ref: prefill.cuh:795-858): a query past the request's
  // prefix sees the prefix and the tokens of its own item only
  const uint32_t* prefix_len_ptr;          // [batch], NULL: off
  const uint16_t* token_pos_in_items_ptr;  // [batch, token_pos_in_items_len]: position of a token in its item
  int32_t token_pos_in_items_len;
  int64_t q_stride_n, q_stride_h;
  int64_t kv_stride_page, kv_stride_n, kv_stride_h;  // host checks stride_page / stride_n < 2^31
  int32_t num_work;
  int32_t num_qo_heads, num_kv_heads, group_size;
  int32_t page_size;
  FastDiv page_div;
  FastDiv group_div;
  int32_t single_qo_len, single_kv_len;
  int32_t causal;
  int32_t window_left;  // < 0 off
  int32_t use_alibi;
  int32_t o_dtype;      // FI_DTYPE_F16 / BF16
  int32_t fp8_p_quant;  // round P through e4m3 (fp8 Q path)
  int32_t tile_q;       // packed query rows per workgroup the plan was cut for (128; 256: fp8-native 8-wave form)
  int32_t bf16_pv_mode;  // fi_batch_prefill_params_t.bf16_pv_mode (host-side kernel choice only)
  float logits_soft_cap;
  float sm_scale;
  float rope_rcp_scale, rope_rcp_theta;
};

template <int T16>
struct MfmaType;
template <>
struct MfmaType<FI_DTYPE_F16> {
  using elem = _Float16;
  using frag = __attribute__((ext_vector_type(8))) _Float16;
  using f32x16 = __attribute__((ext_vector_type(16))) float;
  static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint16_t from_f32(float x) { return f32_to_f16_bits(x); }
  static __device__ __forceinline__ float to_f32(uint16_t b) {
    return (float)__builtin_bit_cast(_Float16, b);
  }
};
template <>
struct MfmaType<FI_DTYPE_BF16> {
  using elem = __bf16;
  using frag = __attribute__((ext_vector_type(8))) __bf16;
  using f32x16 = __attribute__((ext_vector_type(16))) float;
  static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint16_t from_f32(float x) { return f32_to_bf16_bits(x); }
  static __device__ __forceinline__ float to_f32(uint16_t b) {
    return __builtin_bit_cast(float, (uint32_t)b << 16);
  }
};

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((address_space(3))) void lds_void;

// 16-bit MFMA as an asm statement with chosen register classes (head_dim 256: the O accumulators -- 128 registers --
// and the Q fragments -- 64 -- live in ACCUMULATOR registers, "a"; with the builtin the compiler kept shuttling ~400
// values per tile between the two files because the rare rescale branch touches O with vector instructions).
// CA: C / D in AGPRs, BA: the B operand in AGPRs.  The leading s_nop covers a vector write of an operand right in
// front (hipcc pads nothing inside asm); readers of the result are fenced by hand at each use (mfma16_settle).
template <int T16, bool CA, bool BA, typename Frag>
__device__ __forceinline__ void mfma16_asm(f32x16& c, const Frag& a, const Frag& b) {
#define FI_MFMA16_EMIT(NAME)                                                                                    \
  if constexpr (CA && BA) asm volatile("s_nop 1\n\t" NAME " %0, %1, %2, %0" : "+a"(c) : "v"(a), "a"(b));        \
  else if constexpr (CA) asm volatile("s_nop 1\n\t" NAME " %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));         \
  else if constexpr (BA) asm volatile("s_nop 1\n\t" NAME " %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));         \
  else asm volatile("s_nop 1\n\t" NAME " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  if constexpr (T16 == FI_DTYPE_BF16) { FI_MFMA16_EMIT("v_mfma_f32_32x32x16_bf16") }
  else { FI_MFMA16_EMIT("v_mfma_f32_32x32x16_f16") }
#undef FI_MFMA16_EMIT
}
// result latency of the asm MFMAs (8 passes + 2 wait states) before anything but an accumulate-chain MFMA touches them
#define FI_MFMA16_SETTLE_V2(x0, x1) asm volatile("s_nop 7\n\ts_nop 3" : "+v"(x0), "+v"(x1))
#define FI_MFMA16_SETTLE_A8(o) \
  asm volatile("s_nop 7\n\ts_nop 3" : "+a"(o[0]), "+a"(o[1]), "+a"(o[2]), "+a"(o[3]), "+a"(o[4]), "+a"(o[5]), "+a"(o[6]), "+a"(o[7]))

// value held by the partner lane (lane ^ 32) via v_permlane32_swap (VALU; no LDS round trip)
__device__ __forceinline__ float swap_halves(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  // r[0] = {x.lo, x.lo}, r[1] = {x.hi, x.hi} per 32-lane half: the partner's value is the other half's
  const bool upper = (threadIdx.x & 32) != 0;
  return __builtin_bit_cast(float, upper ? r[0] : r[1]);
}

// two f32 -> one dword of two 16-bit values (single v_cvt_pk_* instruction)
template <int T16>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  if constexpr (T16 == FI_DTYPE_BF16) {
    using bf2 = __attribute__((ext_vector_type(2))) __bf16;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf2));
  } else {
    using h2 = __attribute__((ext_vector_type(2))) _Float16;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, h2));
  }
}
using s16x4 = __attribute__((ext_vector_type(4))) short;

// 8 fp8 bytes -> 8 values of the 16-bit compute type, packed in a u32x4
template <int T16, int FP8_DT>
__device__ __forceinline__ u32x4 fp8x8_to_16(u32x2 raw) {
  u32x4 out;
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    f32x2 lo, hi;
    if constexpr (FP8_DT == FI_DTYPE_FP8_E4M3) {
      lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)raw[w], false);
      hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)raw[w], true);
    } else {
      lo = __builtin_amdgcn_cvt_pk_f32_bf8((int)raw[w], false);
      hi = __builtin_amdgcn_cvt_pk_f32_bf8((int)raw[w], true);
    }
    out[2 * w] = pack2<T16>(lo[0], lo[1]);
    out[2 * w + 1] = pack2<T16>(hi[0], hi[1]);
  }
  return out;
}

__device__ __forceinline__ float round_through_e4m3(float x) {
  // x is already scaled into the e4m3 range; returns the e4m3-rounded value as f32
  int packed = __builtin_amdgcn_cvt_pk_fp8_f32(x, 0.f, 0, false);
  return __builtin_amdgcn_cvt_pk_f32_fp8(packed, false)[0];
}

// T16: MFMA compute type; KVS: K/V storage dtype; QS: Q storage dtype; D: head_dim (64 / 128)
// GEN: logits / mask features compiled in, a bit mask -- 1 ALiBi, 2 logits soft cap, 4 custom bit mask, 8 multi-item
//   scoring.  0: the plain kernel.  A single bit: that feature only, unconditionally.  15: all four behind wave-uniform
//   runtime tests -- which the compiler if-converts, so every element pays for every feature (a tanh, a mask bit
//   extraction, two compares ...: 2.8x the vector instructions of the plain loop); it serves feature combinations
//   and the fused-RoPE instantiations, the single-bit forms the common cases.
// PMODE (bf16 q only; how P enters P.V -- a bf16 P carries 8 mantissa bits, which shows as ~2^-9 sum |p v| on
// cancelling rows, 4e-3 on unit-variance V):
//   0  P rounded once to bf16: the reference's arithmetic (prefill.cuh:962-985)
//   1  P as hi + lo bf16 halves, two MFMAs over the same V fragment (16 mantissa bits; +50 % P.V MFMAs)
//   2  P.V on the f16 MFMA: P rounded to f16 (11 bits, <= 2^6 by the deferred rescale) and V converted to f16 while
//      it is staged (exact for |v| < 65504 -- larger values saturate there -- and free for an fp8 cache, which is
//      converted anyway); QK^T stays on the bf16 MFMA.  The default: the 1e-3 bar at the cost of ~48 conversions
//      per tile instead of 16 MFMAs + 80 vector instructions.
template <int T16, int KVS, int QS, int D, bool ROPE, int GEN, int PMODE>
// head_dim 256 keeps 128 accumulator + 64 query-fragment registers per lane: one wave per SIMD (512 registers)
__global__ void __launch_bounds__(kPrefillThreads, D == 256 ? 1 : 2)
    batch_prefill_kernel(const PrefillKernelParams p) {
  using M = MfmaType<T16>;
  using frag_t = typename M::frag;
  constexpr bool KV_FP8 = (KVS == FI_DTYPE_FP8_E4M3 || KVS == FI_DTYPE_FP8_E5M2);
  constexpr bool Q_FP8 = (QS == FI_DTYPE_FP8_E4M3 || QS == FI_DTYPE_FP8_E5M2);
  constexpr bool GENERAL = GEN != 0;
  static_assert(GEN == 0 || GEN == 1 || GEN == 2 || GEN == 4 || GEN == 8 || GEN == 15, "feature mask");
  // feature f is on: compiled in, and (only in the all-features form) asked for at run time
  const bool f_alibi = (GEN & 1) && (GEN != 15 || p.use_alibi);
  const bool f_cap = (GEN & 2) && (GEN != 15 || p.logits_soft_cap > 0.f);
  const bool f_multi = (GEN & 8) && (GEN != 15 || p.prefix_len_ptr != nullptr);
  [[maybe_unused]] constexpr float kPScale = (QS == FI_DTYPE_FP8_E5M2) ? 57344.f : 448.f;
  constexpr bool P_HI_LO = PMODE == 1 && T16 == FI_DTYPE_BF16 && !Q_FP8;
  constexpr bool PV_F16 = PMODE == 2 && T16 == FI_DTYPE_BF16 && !Q_FP8;
  constexpr int TPV = PV_F16 ? FI_DTYPE_F16 : T16;  // operand type of the P.V MFMA
#ifndef FI_PF_ACC_AGPR
#define FI_PF_ACC_AGPR 1
#endif
#ifndef FI_PF_QK_PREFETCH_256
#define FI_PF_QK_PREFETCH_256 4
#endif
#ifndef FI_PF_PV_PREFETCH
#define FI_PF_PV_PREFETCH 3  // head_dim 256: V^T fragments in flight ahead of their P.V MFMA (0: read right before use)
#endif
  // head_dim 256 (one wave per SIMD, 512 registers = 256 vector + 256 accumulator): O and Q in accumulator registers
  constexpr bool ACC_AGPR = FI_PF_ACC_AGPR && D == 256 && !Q_FP8;
  using MPV = MfmaType<TPV>;
  [[maybe_unused]] constexpr int KV_BYTES = KV_FP8 ? 1 : 2;
  constexpr int ROWB = D * 2;             // bytes per row of the 16-bit LDS images
  constexpr int CPR = D / 8;              // 16-byte chunks per row
  constexpr int RPP = kPrefillThreads / CPR;  // rows staged per pass
  constexpr int NPASS = kTileKV / RPP;
  constexpr int KSTEPS = D / 16;          // MFMA k-steps over head_dim
  constexpr int DBLK = D / 32;            // 32-row blocks of O^T
  constexpr int TILE_BYTES = kTileKV * ROWB;
  static_assert(D == 64 || D == 128 || D == 256, "head_dim 64 / 128 / 256");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [stage][K tile | V tile]
  char* const lds_base = smem;
  // byte offset of every kv row of a tile inside the cache (page gather resolved once per workgroup by
  // one wave, three tiles ahead); slot = tile index mod 4
  __shared__ uint64_t row_off_tab[4][kTileKV];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 31;   // query column of this lane
  const int lh = lane >> 5;   // lane half

  // ---- which (request, q tile, kv head) ----
  // Linear block id -> XCD-contiguous logical id (blocks b and b+8 share an XCD): every XCD gets a
  // contiguous run of logical ids, and logical ids are ordered (kv head, work item) with the q tiles of
  // one request adjacent, so workgroups that stream the same K/V pages share an L2.
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
  const bool split = p.kv_tile_indices != nullptr || p.num_kv_chunks > 1;
  if (p.request_indices) {
    req = p.request_indices[work];
    q_tile = p.qo_tile_indices[work];
    if (req < 0) return;  // padding item of a fixed-shape (graph) launch; uniform for the workgroup
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
  const int row0 = q_tile * kTileQ + wave * 32;  // first packed row of this wave
  const int pr = row0 + lq;
  const bool row_valid = pr < packed_len;
  // A wave none of whose 32 rows exist (a decode-like request fills 1-8 of the tile's 128 rows) only helps staging
  // the K / V tiles: no MFMA, no softmax.  In a mixed batch its SIMD's matrix pipe is then free for the co-resident
  // workgroup's full tiles (wave-uniform, constant for the kernel's lifetime).
#ifndef FI_PF_WAVE_SKIP
#define FI_PF_WAVE_SKIP 1
#endif
  // (not at head_dim 256: the second path made that 512-register instantiation spill inside its loop)
  const bool wave_active = !FI_PF_WAVE_SKIP || D == 256 || row0 < packed_len;
  const int prc = row_valid ? pr : (packed_len > 0 ? packed_len - 1 : 0);
  const int qo_idx = (int)fast_div((uint32_t)prc, p.group_div);
  const int hg = prc - qo_idx * G;
  const int qo_head = kv_head * G + hg;
  // query position on the kv axis (ref: prefill.cuh:465-499, 782-786)
  const int q_pos = kv_len - qo_len + qo_idx;

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q, h) holds Q[q][16 ks + 8 h + 0..7] ----
  frag_t qf[KSTEPS];
  {
    const int64_t qb = (int64_t)(qo_start + qo_idx) * p.q_stride_n + (int64_t)qo_head * p.q_stride_h;
    u32x4 raw16[KSTEPS];
    if constexpr (Q_FP8) {
      u32x2 raw8[KSTEPS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
        raw8[ks] = *(const u32x2*)((const uint8_t*)p.q + qb + 16 * ks + 8 * lh);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) raw16[ks] = fp8x8_to_16<T16, QS>(raw8[ks]);
    } else {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
        raw16[ks] = *(const u32x4*)((const uint16_t*)p.q + qb + 16 * ks + 8 * lh);
    }
    if constexpr (ROPE) {
      // dims i and i + D/2 pair up: k-steps ks and ks + KSTEPS/2 of the same lane
#pragma unroll
      for (int ks = 0; ks < KSTEPS / 2; ++ks) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          uint32_t lo_w = raw16[ks][w], hi_w = raw16[ks + KSTEPS / 2][w];
          uint32_t out_lo = 0, out_hi = 0;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int i = 16 * ks + 8 * lh + 2 * w + e;  // < D/2
            const float freq =
                p.rope_rcp_scale * __powf(p.rope_rcp_theta, (float)(2 * i) / (float)D);
            float sn, cs;
            sincos_ool((float)q_pos * freq, &sn, &cs);
            const float a = M::to_f32((uint16_t)(lo_w >> (16 * e)));
            const float b = M::to_f32((uint16_t)(hi_w >> (16 * e)));
            out_lo |= (uint32_t)M::from_f32(a * cs - b * sn) << (16 * e);
            out_hi |= (uint32_t)M::from_f32(b * cs + a * sn) << (16 * e);
          }
          raw16[ks][w] = out_lo;
          raw16[ks + KSTEPS / 2][w] = out_hi;
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = __builtin_bit_cast(frag_t, raw16[ks]);
  }

  // ---- logits scale (ref: variants.cuh:47-53; fp8: hopper/variants.cuh:74-76) ----
  const bool soft_cap = f_cap;
  float qk_scale = p.sm_scale;
  if (p.scale_q) qk_scale *= p.scale_q[qo_head];
  if (p.scale_k) qk_scale *= p.scale_k[kv_head];
  const float c_log2 = qk_scale * kLog2e;
  const float inv_qk_scale = 1.0f / qk_scale;
  const float inv_cap = soft_cap ? 1.0f / p.logits_soft_cap : 0.f;
  const float slope = f_alibi ? p.alibi_slopes[qo_head] : 0.f;
  const uint8_t* const mask_bits =
      ((GEN & 4) && p.custom_mask) ? p.custom_mask + (p.mask_indptr ? p.mask_indptr[req] : 0) : nullptr;
  const uint64_t mask_row = (uint64_t)qo_idx * (uint64_t)kv_len;
  const uint64_t mask_bytes = ((uint64_t)qo_len * (uint64_t)kv_len + 7) >> 3;  // of this request
  // multi-item scoring: this row's item starts at q_pos - item_pos (ref: logits_mask_multi_item_scoring,
  // prefill.cuh:845-856: a query at p >= prefix_len sees kv_idx < prefix_len and kv_idx > p - token_pos[p - prefix])
  int mi_prefix = 0x7fffffff, mi_item_lo = 0;  // rows inside the prefix: plain causal
  if constexpr ((GEN & 8) != 0) {
    if (f_multi) {
      const int pl = (int)p.prefix_len_ptr[req];
      if (q_pos >= pl && q_pos < kv_len) {
        mi_prefix = pl;
        mi_item_lo = q_pos - (int)p.token_pos_in_items_ptr[(int64_t)req * p.token_pos_in_items_len + (q_pos - pl)];
      }
    }
  }

  // ---- kv range of this workgroup ----
  int kv_end = kv_len;
  if (p.causal) {
    // largest query position in the workgroup tile sees keys up to itself
    const int last_pr = min(q_tile * kTileQ + kTileQ, packed_len) - 1;
    const int last_qo = last_pr >= 0 ? (int)fast_div((uint32_t)last_pr, p.group_div) : 0;
    kv_end = min(kv_len, max(0, kv_len - qo_len + last_qo + 1));
  }
  // sliding window: keys left of the window of the tile's FIRST query row are visible to no row of the
  // tile; start at the kv tile that contains that boundary (ref: kv_start_idx, prefill.cuh:1794-1801)
  int kv_begin = 0;
  if (p.window_left >= 0) {
    const int first_pr = min(q_tile * kTileQ, max(packed_len - 1, 0));
    const int first_qo = (int)fast_div((uint32_t)first_pr, p.group_div);
    kv_begin = max(kv_len - qo_len + first_qo - p.window_left, 0) / kTileKV * kTileKV;
  }
  if (split) {
    const int kv_chunk_size = p.kv_chunk_size_ptr ? *p.kv_chunk_size_ptr : p.kv_chunk_size;
    kv_begin += kv_chunk * kv_chunk_size;
    kv_end = min(kv_end, kv_begin + kv_chunk_size);
  }
  const int tile_base = kv_begin / kTileKV;  // window starts and chunks sit on tile boundaries
  const int num_tiles = kv_end > kv_begin ? (kv_end - kv_begin + kTileKV - 1) / kTileKV : 0;
  // visible kv index range of this lane's query row (ref: prefill.cuh:782-786, variants.cuh:87-89).  A row
  // that sees no key at all (causal with qo_len > kv_len: q_pos < 0) gets a range no index can fall into.
  const int vis_hi_raw = p.causal ? min(kv_len - 1, q_pos) : kv_len - 1;
  const int vis_lo_raw = p.window_left >= 0 ? max(q_pos - p.window_left, 0) : 0;
  const bool sees_none = vis_hi_raw < vis_lo_raw;
  const int vis_lo = sees_none ? 0x40000000 : vis_lo_raw;
  const int vis_hi = sees_none ? 0x40000000 : vis_hi_raw;  // span 0 around an unreachable index
  // smallest query position of this WAVE (wave-uniform): tiles ending at or below it need no causal mask
  const int first_qo_wave = (int)fast_div((uint32_t)min(row0, max(packed_len - 1, 0)), p.group_div);
  const int min_qpos_wave = kv_len - qo_len + first_qo_wave;

  // ---- staging geometry ----
  const int st_row = tid / CPR;  // row within a pass
  const int st_ch = tid % CPR;   // 16-byte chunk (8 elements) within the row
  const int64_t head_off = (int64_t)kv_head * p.kv_stride_h;

  // this thread's chunk inside a kv row, folded into per-thread K / V base pointers
  const char* const k_thr = (const char*)p.k + (head_off + st_ch * 8) * KV_BYTES;
  const char* const v_thr = (const char*)p.v + (head_off + st_ch * 8) * KV_BYTES;
  const uint32_t stride_page32 = (uint32_t)p.kv_stride_page, stride_n32 = (uint32_t)p.kv_stride_n;
  // (page id, in-page entry) of kv row `lane` of a tile: evaluated by ONE wave per tile
  auto tab_lookup = [&](int tile, int& pg, int& en) {
    const int kvi = max(min(tile * kTileKV + lane, kv_len - 1), 0);
    const int pi = (int)fast_div((uint32_t)kvi, p.page_div);
    en = kvi - pi * p.page_size;
    pg = p.kv_indices ? p.kv_indices[page_begin + pi] : page_begin + pi;
  };
  auto tab_store = [&](int slot, int pg, int en) {
    // 32 x 32 -> 64-bit multiply-adds (strides fit in 31 bits, checked on the host)
    row_off_tab[slot][lane] =
        ((uint64_t)(uint32_t)pg * stride_page32 + (uint64_t)(uint32_t)en * stride_n32) * KV_BYTES;
  };
  auto read_offsets = [&](int slot, uint64_t (&roff)[NPASS]) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) roff[ps] = row_off_tab[slot][ps * RPP + st_row];
  };
  // One register set stages K and then V of the next tile (K is written to LDS -- the OTHER buffer, free
  // since the last barrier -- as soon as QK^T of the current tile has been issued, then the same registers
  // take the V rows).
  struct Stage {
    u32x4 r[NPASS];
  };
  auto issue_loads = [&](const char* base_thr, const uint64_t (&roff)[NPASS], Stage& st) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      if constexpr (KV_FP8) {
        const u32x2 rk = *(const u32x2*)(base_thr + roff[ps]);
        st.r[ps] = u32x4{rk[0], rk[1], 0, 0};
      } else {
        st.r[ps] = *(const u32x4*)(base_thr + roff[ps]);
      }
    }
  };
  // fused RoPE on K: this thread always stages the same 8 dims of rows RPP apart (pass to pass AND tile to tile),
  // so cos / sin of its angles advance by a fixed rotation per pass -- an angle-addition recurrence (4 FMAs per
  // dim) re-seeded with the hardware sin / cos every 8 tiles, instead of a range reduction + v_sin + v_cos per
  // element.  The per-pass rotation depends on the chunk only and sits in LDS (16 registers otherwise); the sign
  // of the sine term (- for the first half of the dims) is folded into the state.
  __shared__ __attribute__((aligned(16))) float rope_step[ROPE ? CPR * 16 : 4];
  [[maybe_unused]] float rope_c[ROPE ? 8 : 1], rope_s[ROPE ? 8 : 1];
  [[maybe_unused]] auto rope_freq = [&](int e) {
    const int i = (st_ch * 8 + e) % (D / 2);
    return p.rope_rcp_scale * __powf(p.rope_rcp_theta, (float)(2 * i) / (float)D);
  };
  if constexpr (ROPE) {
    const float rope_sgn = (st_ch < CPR / 2) ? -1.f : 1.f;
    if (st_row == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float sn, cs;
        fast_sincos((float)RPP * rope_freq(e), &sn, &cs);
        rope_step[st_ch * 16 + e] = cs;
        rope_step[st_ch * 16 + 8 + e] = rope_sgn * sn;
      }
    }
    __syncthreads();  // before the first advance reads it (every thread of the workgroup gets here)
  }
  // swizzles (see header comment): K image for ds_read_b128, V image for ds_read_b64_tr_b16
  auto k_lds_off = [&](int row, int ch) -> int {
    const int sw = (CPR >= 16) ? (row & 15) : ((row >> 1) & 7);
    return row * ROWB + ((ch ^ sw) << 4);
  };
  auto v_lds_off = [&](int row, int ch) -> int {
    const int f = (ROWB >= 256) ? (row & 3) : ((row >> 1) & 1);
    const int g64 = (ch >> 2) ^ f;
    return row * ROWB + (g64 << 6) + ((ch & 3) << 4);
  };
  // always_inline: called from two sites; outlined, its by-reference captures (the RoPE state) would live in scratch
  auto write_k = [&](int tile, int buf, const Stage& st) __attribute__((always_inline)) {
    char* kb = lds_base + buf * 2 * TILE_BYTES;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int row = ps * RPP + st_row;
      u32x4 kw;
      if constexpr (KV_FP8) kw = fp8x8_to_16<T16, KVS>(u32x2{st.r[ps][0], st.r[ps][1]});
      else kw = st.r[ps];
      if constexpr (ROPE) {
        // rotate K at its absolute position (ref: k_smem_inplace_apply_rotary, prefill.cuh:536-612).  The
        // partner chunk (dims +- D/2) sits CPR/2 lanes away: its packed registers are fetched with four
        // cross-lane moves.
        // (the chunk index passes through an empty asm: what is derived from it -- the sign, the table address -- is
        // recomputed here instead of occupying registers across the tile loop, where three of them spilled)
        int sc = st_ch;
        asm volatile("" : "+v"(sc));
        if (ps == 0 && ((tile - tile_base) & 7) == 0) {
          const float rope_sgn = (sc < CPR / 2) ? -1.f : 1.f;
          const int kvi = tile * kTileKV + row;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float sn, cs;
            fast_sincos((float)kvi * rope_freq(e), &sn, &cs);
            rope_c[e] = cs;
            rope_s[e] = rope_sgn * sn;
          }
        }
        u32x4 pw;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint32_t mine = kw[w];  // scalar copy first: bit_cast of a vector-element lvalue reads element 0
          pw[w] = __builtin_bit_cast(uint32_t, lane_xor<CPR / 2>(__builtin_bit_cast(float, mine)));
        }
        u32x4 outw;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float y[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float x = M::to_f32((uint16_t)(kw[w] >> (16 * e)));
            const float partner = M::to_f32((uint16_t)(pw[w] >> (16 * e)));
            y[e] = __builtin_fmaf(partner, rope_s[2 * w + e], x * rope_c[2 * w + e]);
          }
          outw[w] = pack2<T16>(y[0], y[1]);
        }
        kw = outw;
        // advance to the next pass (RPP rows on)
        const float* const tab = rope_step + sc * 16;  // read here every pass (kept, the table costs 16 registers)
#pragma unroll
        for (int e4 = 0; e4 < 8; e4 += 4) {
          const f32x4 dc4 = *(const f32x4*)(tab + e4), ds4 = *(const f32x4*)(tab + 8 + e4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float cn = rope_c[e4 + e] * dc4[e] - rope_s[e4 + e] * ds4[e];
            const float sn = rope_s[e4 + e] * dc4[e] + rope_c[e4 + e] * ds4[e];
            rope_c[e4 + e] = cn;
            rope_s[e4 + e] = sn;
          }
        }
      }
      *(u32x4*)(kb + k_lds_off(row, st_ch)) = kw;
    }
  };
  auto write_v = [&](int buf, const Stage& st) {
    char* vb = lds_base + buf * 2 * TILE_BYTES + TILE_BYTES;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int row = ps * RPP + st_row;
      u32x4 vw;
      if constexpr (KV_FP8) {
        vw = fp8x8_to_16<TPV, KVS>(u32x2{st.r[ps][0], st.r[ps][1]});
      } else if constexpr (PV_F16) {
        // bf16 -> f16, round to nearest even (v_cvt_pk_f16_f32): exact for 2^-14 <= |v| < 65504 (8 significant bits
        // fit 11); below that the value lands in f16's subnormal range and loses low bits (unbiased; relative 2^-10 at
        // 6e-5 growing to 2^-1 at 6e-8); |v| >= 65520 becomes +-inf, NOT a finite clamp -- the wrapper / plan option
        // `bf16_pv_exact_range=True` (hi + lo bf16 P on the bf16 MFMA, no range limit) is for caches that hold such
        // values
        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint32_t raw = st.r[ps][w];
          const f16x2_t h2 = {(_Float16)__builtin_bit_cast(float, raw << 16),
                              (_Float16)__builtin_bit_cast(float, raw & 0xffff0000u)};
          vw[w] = __builtin_bit_cast(uint32_t, h2);
        }
      } else {
        vw = st.r[ps];
      }
      *(u32x4*)(vb + v_lds_off(row, st_ch)) = vw;
    }
  };

  // ---- per-lane LDS read addresses ----
  // K fragment (A operand): row 32 kb + lq, chunk (2 ks + lh) ^ swizzle(row)
  // chunk 2 ks + lh = (2 ks) ^ lh (disjoint bits), so k_rd(ks) = k_rd_base ^ (ks << 5): ONE lane-constant
  // register; the tile body re-derives the per-step addresses (an asm barrier keeps them from being
  // hoisted: as loop invariants they cost KSTEPS + DBLK registers, and in the variants that spill every
  // in-loop reload -- a VMEM op -- drags a vmcnt(0) wait on the freshly issued K/V loads into the MFMAs)
  int k_rd_base = k_lds_off(lq, lh);
  auto k_rd = [&](int ks) { return k_rd_base ^ (ks << 5); };
  // V^T fragment via transposed read: within a 16-lane group, lane 4*q4 + p4 addresses row q4,
  // columns 4 p4 .. 4 p4 + 3 of a 4 x 16 block; the group's block is rows 16 s + 4 lh + (0..3) [+8],
  // columns 32 db + 16 gpar + (0..15)
  const int q4 = (lane & 15) >> 2, p4 = lane & 3, gpar = (lane >> 4) & 1;
  int v_rd_base;  // 64-byte group db ^ f: v_rd(db) = v_rd_base ^ (db << 6)
  {
    const int row = 4 * lh + q4;
    const int col_byte = (16 * gpar + 4 * p4) * 2;
    const int f = (ROWB >= 256) ? (row & 3) : ((row >> 1) & 1);
    v_rd_base = row * ROWB + (f << 6) + col_byte;
  }
  auto v_rd = [&](int db) { return v_rd_base ^ (db << 6); };

  // ---- running state ----
  f32x16 o_acc[DBLK];
#pragma unroll
  for (int db = 0; db < DBLK; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[db][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  [[maybe_unused]] uint32_t mask_dw[3] = {~0u, ~0u, ~0u};
  [[maybe_unused]] auto mask_fetch = [&](int t_rel) {
    const uint64_t bit0 = mask_row + (uint64_t)((tile_base + t_rel) * kTileKV);
    const uintptr_t a0 = ((uintptr_t)mask_bits + (bit0 >> 3)) & ~(uintptr_t)3;
    const uintptr_t last = ((uintptr_t)mask_bits + mask_bytes - 1) & ~(uintptr_t)3;
#pragma unroll
    for (int i = 0; i < 3; ++i) mask_dw[i] = *(const uint32_t*)(a0 + 4 * i < last ? a0 + 4 * i : last);
  };
  if (num_tiles > 0) {
    if constexpr (GENERAL)
      if (mask_bits) mask_fetch(0);
    // K rows are loaded TWO tiles ahead (during tile t the K rows of tile t+2 are in flight into kst; a tile
    // period is about one HBM round trip under load, the QK^T phase alone is not), V rows one tile ahead
    // (issued after QK^T, written to LDS before the closing barrier).  The row-offset table of tile t+3 is
    // produced during tile t by wave t % 4.
    uint64_t roff[NPASS];
    Stage kst, vst;
    if (wave < 3) {
      int pg0, en0;
      tab_lookup(tile_base + min(wave, num_tiles - 1), pg0, en0);
      tab_store(wave, pg0, en0);
    }
    __syncthreads();
    read_offsets(0, roff);
    issue_loads(k_thr, roff, kst);
    issue_loads(v_thr, roff, vst);
    write_k(tile_base, 0, kst);
    write_v(0, vst);
    read_offsets(1, roff);
    issue_loads(k_thr, roff, kst);
    __syncthreads();
    // The tile body is instantiated for the even and the odd LDS buffer so that every LDS address is a
    // lane-constant register plus an immediate.  Tiles past the end are clamped to the last one (re-staged
    // into the idle buffer): no branch around a load, so the compiler's vmcnt counts stay exact.
    auto tile_body = [&](auto buf_c, const int t) {
      constexpr int buf = decltype(buf_c)::value;
      if (!wave_active) {
        // staging only (same loads, LDS writes, table duty and barrier as below; a separate straight-line path, so
        // that the full body keeps the schedule it has without this test)
        const int t_nx = tile_base + min(t + 1, num_tiles - 1);
        write_k(t_nx, buf ^ 1, kst);
        read_offsets((t + 2) & 3, roff);
        issue_loads(k_thr, roff, kst);
        int pg = 0, en = 0;
        if (wave == (t & 3)) tab_lookup(tile_base + min(t + 3, num_tiles - 1), pg, en);
        read_offsets((t + 1) & 3, roff);
        issue_loads(v_thr, roff, vst);
        write_v(buf ^ 1, vst);
        if (wave == (t & 3)) tab_store((t + 3) & 3, pg, en);
        __syncthreads();
        return;
      }
      asm volatile("" : "+v"(k_rd_base), "+v"(v_rd_base));
      const int t_next = tile_base + min(t + 1, num_tiles - 1);
      write_k(t_next, buf ^ 1, kst);     // K rows of tile t+1 (loaded during tile t-1)
      read_offsets((t + 2) & 3, roff);
      issue_loads(k_thr, roff, kst);     // K rows of tile t+2
      const bool tab_wave = wave == (t & 3);
      int tab_pg = 0, tab_en = 0;
      if (tab_wave) tab_lookup(tile_base + min(t + 3, num_tiles - 1), tab_pg, tab_en);
      const char* kb = lds_base + buf * 2 * TILE_BYTES;
      const char* vb = kb + TILE_BYTES;
      const int tile0 = (tile_base + t) * kTileKV;

      // ---- S^T = K Q^T ----
      // The 2 x KSTEPS K fragments are read kQkPrefetch MFMAs ahead of their use (the LDS round trip is 2-4
      // MFMA slots); sched_group_barrier pins that order, which the scheduler otherwise collapses into
      // read -> wait -> MFMA pairs.
      f32x16 s_acc[2];
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[kbk][r] = 0.f;
      {
        constexpr int NK = 2 * KSTEPS;
        // feature / RoPE variants: fewer fragments in flight (registers); head_dim 256 (one wave per SIMD): deeper
        constexpr int PF = ROPE ? 1 : GENERAL ? 2 : (D == 256 ? FI_PF_QK_PREFETCH_256 : kQkPrefetch);
        u32x4 kf[NK];
        auto rd = [&](int i) {
          return *(const u32x4*)(kb + (i / KSTEPS) * 32 * ROWB + k_rd(i % KSTEPS));
        };
#pragma unroll
        for (int i = 0; i < PF; ++i) kf[i] = rd(i);
#pragma unroll
        for (int i = 0; i < NK; ++i) {
          if (i + PF < NK) kf[i + PF] = rd(i + PF);
          if constexpr (ACC_AGPR)
            mfma16_asm<T16, false, true>(s_acc[i / KSTEPS], __builtin_bit_cast(frag_t, kf[i]), qf[i % KSTEPS]);
          else
            s_acc[i / KSTEPS] = M::mfma(__builtin_bit_cast(frag_t, kf[i]), qf[i % KSTEPS], s_acc[i / KSTEPS]);
        }
        if constexpr (!ACC_AGPR) {  // (asm MFMAs keep their source order by themselves)
          __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
          for (int i = 0; i < NK - PF; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, PF, 0);
        } else {
          FI_MFMA16_SETTLE_V2(s_acc[0], s_acc[1]);
        }
      }
      // nothing of the staging below (it waits for the K loads issued at the top) may move up into QK^T
      __builtin_amdgcn_sched_barrier(0);

      read_offsets((t + 1) & 3, roff);
      issue_loads(v_thr, roff, vst);     // V rows of tile t+1

      // ---- logits transform + mask (ref: variants.cuh:67-91, prefill.cuh:782-786) ----
      // x holds c*logit (base-2 units) for the general path, or the RAW dot product on the plain path
      // (scale folded into the exp2 argument: p = 2^(s*c - m)).
      const bool need_mask = (tile0 + kTileKV > kv_len) ||
                             (p.causal && tile0 + kTileKV - 1 > min_qpos_wave) ||
                             (p.window_left >= 0);
      if constexpr (GENERAL) {
        // custom mask: the 64 bits of this lane's query row for the tile's keys (bit qo_idx * kv_len + kv_idx,
        // little-endian; ref: variants.cuh:80-86) sit in at most three aligned dwords; two 32-bit windows, one
        // per 32-key block, are cut out of them.  Dword addresses are clamped to the request's mask: bits past
        // its end belong to no visible key (an aligned dword never leaves the page of its first byte).
        uint32_t mask_win[2] = {~0u, ~0u};
        if (mask_bits) {
          // mask_dw: the three aligned dwords around this row's 64 bits, fetched during the previous tile
          const uint64_t bit0 = mask_row + (uint64_t)tile0;
          const uint32_t sh = (uint32_t)((((uintptr_t)mask_bits + (bit0 >> 3)) & 3) * 8 + (bit0 & 7));
          mask_win[0] = __builtin_amdgcn_alignbit(mask_dw[1], mask_dw[0], sh);  // ({dw1, dw0} >> sh) & 0xffffffff
          mask_win[1] = __builtin_amdgcn_alignbit(mask_dw[2], mask_dw[1], sh);
        }
        [[maybe_unused]] const float slope_c = slope * inv_qk_scale;
        [[maybe_unused]] const float alibi_rel0 = (float)(tile0 + 4 * lh - qo_idx);
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            [[maybe_unused]] const int kv_idx = tile0 + 32 * kbk + (r & 3) + 8 * (r >> 2) + 4 * lh;
            // ref: variants.cuh:67-76 -- alibi bias, then soft cap; kept in units of 1/c so that the
            // common exp2(fma(x, c, -m)) below applies
            float lg = s_acc[kbk][r];
            if constexpr (GEN == 1) {
              // ALiBi alone: (s c + slope rel) / c = s + (slope / c) rel -- one FMA on a lane-constant base
              lg = __builtin_fmaf(slope_c, alibi_rel0 + (float)(32 * kbk + (r & 3) + 8 * (r >> 2)), lg);
            } else if constexpr ((GEN & 3) != 0) {
              lg *= qk_scale;
              if (f_alibi) lg += slope * (float)(kv_idx - qo_idx);
              if (soft_cap) lg = p.logits_soft_cap * fast_tanh(lg * inv_cap);
              lg *= inv_qk_scale;
            }
            if constexpr ((GEN & 4) != 0)
              if (mask_bits) lg = ((mask_win[kbk] >> ((r & 3) + 8 * (r >> 2) + 4 * lh)) & 1) ? lg : -INFINITY;
            if constexpr ((GEN & 8) != 0)
              if (f_multi) lg = (kv_idx < mi_prefix || kv_idx > mi_item_lo) ? lg : -INFINITY;
            s_acc[kbk][r] = lg;
          }
        // the next tile's mask dwords go out HERE: younger than the K / V loads this tile issued (which complete
        // before the next tile's softmax anyway) and older than the ones the next tile issues, so the wait at
        // their use leaves the K / V prefetch in flight.  (Fetched right before use -- 9 byte loads -- the wait
        // drained the whole prefetch every tile: custom masks ran at 0.42 of the plain kernel's rate.)
        if (mask_bits) mask_fetch(min(t + 1, num_tiles - 1));
      }
      if (need_mask) {
        // visible kv range of this lane's query: [vis_lo, vis_hi]; one unsigned compare per element
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

      // ---- online softmax (base 2; ref: prefill.cuh:861-953) ----
      float mx = s_acc[0][0];
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[kbk][r]);
      mx = fmaxf(mx, swap_halves(mx));
      // Deferred rescale: the reference exponent m_run only moves when some row of the wave outgrew it by
      // more than kRescaleLog2 (then every row jumps to its true maximum); until then P is formed against
      // the stale m_run and may reach 2^kRescaleLog2 -- exact power-of-two scaling, cancelled by l_run.
      // Not used when P is rounded to e4m3 (fp8 q): that needs P <= 1.
      constexpr float kRescaleLog2 = Q_FP8 ? 0.f : 6.f;
      // PV_F16: P enters the f16 MFMA; it is formed 2^9 up (P <= 2^6 * 2^9 = 2^15 < 65504), so probabilities down to
      // 2^-23 of the reference exponent stay NORMAL f16 numbers (unscaled, everything below 6e-5 would lose bits in
      // f16's subnormal range).  The row sum carries the same factor: O / l is free of it, the lse subtracts it.
      constexpr float kPShift = PV_F16 ? 9.f : 0.f;
      const float m_true = fmaxf(m_run, mx * c_log2);  // c_log2 > 0
      if (__any(m_true - m_run > kRescaleLog2)) {
        const float alpha = fast_exp2(m_run - m_true);
        m_run = m_true;
        l_run *= alpha;
        if constexpr (ACC_AGPR) FI_MFMA16_SETTLE_A8(o_acc);
#pragma unroll
        for (int db = 0; db < DBLK; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o_acc[db][r] *= alpha;
      }
      float psum = 0.f;
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s_acc[kbk][r] = fast_exp2(__builtin_fmaf(s_acc[kbk][r], c_log2, kPShift - m_run));
          psum += s_acc[kbk][r];
        }
      l_run += psum;

      // ---- O^T += V^T P^T ----
      // P^T fragments: accumulator registers 8s..8s+7 of block kb, rounded to 16 bit, are the B operand of
      // k-step s (built right before use to keep registers free for LDS prefetch)
      using pv_ring_t = __attribute__((ext_vector_type(8))) short;
      [[maybe_unused]] pv_ring_t pv_ring[FI_PF_PV_PREFETCH > 0 ? FI_PF_PV_PREFETCH : 1];
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          u32x4 w;
          [[maybe_unused]] u32x4 w_lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float a = s_acc[kbk][8 * s2 + 2 * j], b = s_acc[kbk][8 * s2 + 2 * j + 1];
            if constexpr (Q_FP8) {
              // ref: hopper/variants.cuh:71-90 -- P * max(fp8 type) rounded to the q/k/v type (448 e4m3,
              // 57344 e5m2), mainloop_mma.cuh:173 convert_type<DTypeKV>
              if constexpr (QS == FI_DTYPE_FP8_E4M3) {
                const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a * kPScale, b * kPScale, 0, false);
                const f32x2 back = __builtin_amdgcn_cvt_pk_f32_fp8(pk, false);
                a = back[0];
                b = back[1];
              } else {
                const int pk = __builtin_amdgcn_cvt_pk_bf8_f32(a * kPScale, b * kPScale, 0, false);
                const f32x2 back = __builtin_amdgcn_cvt_pk_f32_bf8(pk, false);
                a = back[0];
                b = back[1];
              }
            }
            w[j] = pack2<TPV>(a, b);
            if constexpr (P_HI_LO) {
              // A bf16 P carries 8 mantissa bits: with |o| << |v| (cancelling rows) the rounding shows as
              // ~2^-9 |p v| absolute, 4e-3 on unit-variance V.  The residual p - bf16(p) is exact in f32 and
              // its own bf16 rounding leaves 2^-17 relative.
              const float a_hi = __builtin_bit_cast(float, w[j] << 16);
              const float b_hi = __builtin_bit_cast(float, w[j] & 0xffff0000u);
              w_lo[j] = pack2<T16>(a - a_hi, b - b_hi);
            }
          }
          using pv_frag_t = typename MPV::frag;
          const pv_frag_t pfrag = __builtin_bit_cast(pv_frag_t, w);
          if constexpr (ACC_AGPR && FI_PF_PV_PREFETCH > 0) {
            // One wave per SIMD (head_dim 256): nobody hides an LDS round trip, and with {read, read, wait, MFMA} per
            // block the matrix pipe idled through 32 of them per tile.  The V^T fragments run FI_PF_PV_PREFETCH
            // blocks ahead of their MFMA (across the four (kbk, s2) groups: vfrag_at flattens the index).
            using s16x8 = __attribute__((ext_vector_type(8))) short;
            constexpr int PFV = FI_PF_PV_PREFETCH;
            auto vfrag_at = [&](int idx) {
              const char* base = vb + (32 * (idx / (2 * DBLK)) + 16 * ((idx / DBLK) & 1)) * ROWB + v_rd(idx % DBLK);
              const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
              const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 8 * ROWB));
              return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            const int g0 = (2 * kbk + s2) * DBLK;  // first flattened index of this group
            if (g0 == 0) {
#pragma unroll
              for (int i = 0; i < PFV; ++i) pv_ring[i] = vfrag_at(i);
            }
#pragma unroll
            for (int db = 0; db < DBLK; ++db) {
              const int idx = g0 + db;
              const s16x8 a8 = pv_ring[idx % PFV];
              if (idx + PFV < 4 * DBLK) pv_ring[idx % PFV] = vfrag_at(idx + PFV);
              mfma16_asm<TPV, true, false>(o_acc[db], __builtin_bit_cast(pv_frag_t, a8), pfrag);
              if constexpr (P_HI_LO)
                mfma16_asm<T16, true, false>(o_acc[db], __builtin_bit_cast(frag_t, a8), __builtin_bit_cast(frag_t, w_lo));
            }
            continue;
          }
#pragma unroll
          for (int db = 0; db < DBLK; ++db) {
            const char* base = vb + (32 * kbk + 16 * s2) * ROWB + v_rd(db);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(base));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(base + 8 * ROWB));
            using s16x8 = __attribute__((ext_vector_type(8))) short;
            const s16x8 a8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if constexpr (ACC_AGPR) {
              mfma16_asm<TPV, true, false>(o_acc[db], __builtin_bit_cast(pv_frag_t, a8), pfrag);
              if constexpr (P_HI_LO)
                mfma16_asm<T16, true, false>(o_acc[db], __builtin_bit_cast(frag_t, a8), __builtin_bit_cast(frag_t, w_lo));
            } else {
              o_acc[db] = MPV::mfma(__builtin_bit_cast(pv_frag_t, a8), pfrag, o_acc[db]);
              if constexpr (P_HI_LO)
                o_acc[db] = M::mfma(__builtin_bit_cast(frag_t, a8), __builtin_bit_cast(frag_t, w_lo), o_acc[db]);
            }
          }
        }
      }

      write_v(buf ^ 1, vst);
      if (tab_wave) tab_store((t + 3) & 3, tab_pg, tab_en);
      __syncthreads();
    };
    int t = 0;
    for (; t + 1 < num_tiles; t += 2) {
      tile_body(std::integral_constant<int, 0>{}, t);
      tile_body(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < num_tiles) tile_body(std::integral_constant<int, 0>{}, t);
    if constexpr (ACC_AGPR) FI_MFMA16_SETTLE_A8(o_acc);
  }

  // ---- finalize (ref: prefill.cuh:2378-2403; fp8: attention_updater.cuh:221-240) ----
  l_run += swap_halves(l_run);
  const bool empty = !(l_run > 0.f);
  float inv = empty ? 0.f : 1.0f / l_run;
  if constexpr (Q_FP8) inv *= (p.scale_v ? p.scale_v[kv_head] : 1.f) / kPScale;
  else if (p.scale_v) inv *= p.scale_v[kv_head];
  if (row_valid && split) {
    // partial state of this kv chunk: normalised f32 o + base-2 lse (ref: prefill.cuh:2378-2403)
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
      p.tmp_lse[entry * p.num_qo_heads + qo_head] = empty ? FI_NEG_INF : m_run + fast_log2(l_run) - (PV_F16 ? 9.f : 0.f);
  } else if (row_valid) {
    const int64_t ob = ((int64_t)(qo_start + qo_idx) * p.num_qo_heads + qo_head) * D;
#pragma unroll
    for (int db = 0; db < DBLK; ++db) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d0 = 32 * db + 8 * r4 + 4 * lh;
        // the output type equals the compute type (checked on the host)
        const uint32_t w0 = pack2<T16>(o_acc[db][4 * r4 + 0] * inv, o_acc[db][4 * r4 + 1] * inv);
        const uint32_t w1 = pack2<T16>(o_acc[db][4 * r4 + 2] * inv, o_acc[db][4 * r4 + 3] * inv);
        *(u32x2*)((uint16_t*)p.o + ob + d0) = u32x2{w0, w1};
      }
    }
    if (p.lse && lh == 0)
      p.lse[(int64_t)(qo_start + qo_idx) * p.num_qo_heads + qo_head] =
          empty ? FI_NEG_INF : m_run + fast_log2(l_run) - (PV_F16 ? 9.f : 0.f);
  }
}

}  // namespace fi

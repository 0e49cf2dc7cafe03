// fp8-native batch paged prefill for gfx950 (BASELINE config C3), second structure.
//
// Arithmetic (unchanged; ref FA3 fp8: hopper/variants.cuh:64-102, attention_updater.cuh:167-256,
// quantization/mainloop_mma.cuh:21-237): Q, K, V e4m3; S = Q K^T and O = P8 V on
// v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales; P = 2^(c S - m), row sum from the UNROUNDED P,
// P8 = e4m3(P x const) in registers as the B operand of the second product; O *= scale_v / rowsum at the end;
// lse = m + log2(rowsum) in base 2.
//
// Decomposition (as prefill_kernel.h): workgroup = 4 waves = 128 GQA-packed query rows x one kv head, two
// workgroups per CU (2 waves per SIMD: the softmax is vector-issue bound, and a SIMD issues vector
// instructions from two waves at twice the rate of one); 64-key tiles; S^T = K Q^T with the query row on the
// lane; O^T += V^T P^T with the S^T accumulator registers (as e4m3) as the B operand.
//
// What changed against the first structure (r1: register-staged K / V, one softmax behind its own QK^T; 29 % of the
// MFMA peak, 25 vector instructions per MFMA, matrix pipe idle two thirds of the time; removed in r3):
//   * K / V tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) three
//     tiles ahead through a ring of four 16 KB stages: no staging registers, no ds_write, no vector load in
//     the loop at all (hand-counted vmcnt + raw s_barrier; every LDS object lives in ONE array).  Both images
//     keep their XOR swizzles: the DMA destination is lane-linear, so the swizzle is applied to the per-lane
//     SOURCE chunk.  Page ids sit in LDS (1024 at a time), one wave per tile turns (page id, entry) of the 64
//     rows into byte offsets -- the steady state issues no ordinary global load.
//   * software pipeline inside a wave: iteration t issues the four QK^T MFMAs of tile t+1 interleaved with
//     the exp2 / sum / e4m3 packing of tile t, then the four P.V MFMAs of tile t interleaved with the row
//     maximum of tile t+1.  A wave issues in order, so vector work has to sit BETWEEN its MFMAs in program
//     order to run under them.
//   * deferred rescale: the reference exponent m only moves when a row of the wave outgrows it by more than
//     2^kF8Thr; P8 = e4m3(P x 448 / 2^kF8Thr) keeps the e4m3 range (P <= 2^kF8Thr).  The rescale of the 64
//     O registers leaves the common path.
//   * the loop body exists for each ring stage (all LDS addresses are a lane constant plus an immediate) and
//     without / with the mask code (tiles on the causal diagonal or past kv_len run the masked body).
//   * output rows leave through LDS: 16-byte stores of whole 256-byte rows instead of 8-byte pieces.
#pragma once
#include <type_traits>

#include "prefill_kernel.h"

#ifndef FI_PF8_KO
#define FI_PF8_KO 0  // timing experiments only (results are wrong), bit mask: 1 no exp2, 2 no DMA in the steps,
                     // 4 no barrier / vmcnt wait, 8 no P.V MFMAs, 16 no QK^T MFMAs, 32 no rescale check,
                     // 64 one workgroup per CU (LDS padded)
#endif

#ifndef FI_PF8_ASM_MFMA
#define FI_PF8_ASM_MFMA 1  // the steps' MFMAs as asm volatile statements: they stay where the source puts them
#endif

namespace fi {

using i32x8 = __attribute__((ext_vector_type(8))) int;
using i32x2 = __attribute__((ext_vector_type(2))) int;

// MFMA as an asm statement (v_mfma_scale_f32_32x32x64_f8f6f4, e4m3 x e4m3, unit block scales).  The compiler moves
// builtin MFMAs freely -- across the step barrier and into back-to-back bursts -- while the pipeline below wants each
// one between two blocks of vector work.  What the compiler no longer does for these statements (guide 5.7): hazard
// padding -- the leading s_nop covers a vector write of an operand right before the statement; every reader of a
// result sits at least one MFMA (64 cycles) later in program order (noted at each use).
// BF8: both operands e5m2 (cbsz / blgp = 1) instead of e4m3
template <bool BF8>
__device__ __forceinline__ void mfma_fp8_k64_asm(f32x16& c, const i32x8& a, const i32x8& b, int unit) {
  if constexpr (BF8)
    asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:1 blgp:1"
                 : "+v"(c) : "v"(a), "v"(b), "v"(unit));
  else
    asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                 : "+v"(c) : "v"(a), "v"(b), "v"(unit));
}
template <bool BF8>
__device__ __forceinline__ void mfma_fp8_k64_asm_zero(f32x16& c, const i32x8& a, const i32x8& b, int unit) {
  if constexpr (BF8)
    asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0] cbsz:1 blgp:1"
                 : "=v"(c) : "v"(a), "v"(b), "v"(unit));
  else
    asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]"
                 : "=v"(c) : "v"(a), "v"(b), "v"(unit));
}
template <bool BF8>
__device__ __forceinline__ f32x16 mfma_fp8_k64_fmt(i32x8 a, i32x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, BF8 ? 1 : 0, BF8 ? 1 : 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

constexpr int kF8Stages = 4;        // ring depth (K and V each): tile t+1 / t read, t+2 landed, t+3 in flight
constexpr int kF8Ids = 1024;                             // page ids held in LDS
// LDS layout for head_dim D: [K ring: 4 stages][V ring: 4 stages][uint64 row-offset tables [4][64]][page ids];
// a stage is 64 rows of max(D, 128) bytes (head_dim 64 keeps the 8 KB stride)
template <int D>
struct F8Lds {
  static constexpr int kTile = kTileKV * (D > 128 ? D : 128);
  static constexpr int kVOff = kF8Stages * kTile;
  static constexpr int kTabOff = 2 * kF8Stages * kTile;
  static constexpr int kIdsOff = kTabOff + 4 * kTileKV * 8;
  static constexpr int kSmem = kIdsOff + kF8Ids * 4;  // 71 680 B at D <= 128 (two workgroups per CU), 137 216 B at 256
};
constexpr float kF8Thr = 3.0f;                           // log2 headroom of the deferred rescale
constexpr float kF8Log2Scale = 8.807354922057604f - kF8Thr;   // log2(448) - headroom    (e4m3)
constexpr float kBF8Log2Scale = 15.807354922057604f - kF8Thr;  // log2(57344) - headroom  (e5m2)

// transposed 8-bit LDS read, issued from an asm statement: as an intrinsic the compiler cannot tell it from the
// LDS-DMA targets and drains vmcnt in front of it.  Its completion is awaited by hand (counted lgkmcnt; LDS
// operations of a wave return in order, and the compiler's own counted waits only get stricter).
template <int OFF>
__device__ __forceinline__ i32x2 lds_tr8(int addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  i32x2 w;
  asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(w) : "v"(addr), "i"(OFF));
  return w;
}

typedef __attribute__((address_space(3))) void f8_lds_void;
typedef const __attribute__((address_space(1))) void f8_gbl_void;

// UNI: GQA group size 1, 2 or 4 -- every wave then holds rows of ONE query head (wave w: head w % G, 32 consecutive
// tokens), so the logit scale c = sm_scale * scale_q[head] * scale_k[kv head] * log2 e is wave-uniform and sits in a
// scalar register: exp2's argument fma(s, c, -m) then reads two vector registers instead of three (a vector
// instruction with three distinct vector sources issues at half rate: tools/ubench_issue2 measurements).
// NW: waves per workgroup.  4: 128 packed query rows, two workgroups per CU.  8 (plans cut with cta_tile_q = 256):
// 256 rows share every K/V tile, so a wave issues two LDS-DMA pieces per 64-key step instead of four and the
// page-id / row-offset tables are built once per 256 rows; one workgroup per CU, the same 8 waves.
// BF8: q, k, v e5m2 -- P is scaled by 57344 / 2^kF8Thr and rounded to e5m2 (ref: hopper/variants.cuh:71-73)
// D: head_dim 128, or 64 (NW = 4 only): rows of 64 bytes in the same ring (a stage keeps its 8 KB stride), one k step
// per QK^T block and two P.V blocks -- two MFMAs each per 64-key step against the same softmax work, so that form is
// vector-bound by construction and its step is written plainly (builtin MFMAs, no hand interleave).
// D = 256 (NW = 4; ref instantiation: hopper/quantization/prefill_sm90.cuh:459-470): 256-byte rows, 16 KB K / V tiles,
// one workgroup per CU (128 accumulator + 32 query registers per lane: the 512-register budget of one wave per SIMD),
// four k steps per QK^T block and eight P.V blocks per 64-key step -- 16 MFMAs against the same softmax, written plainly
// like the head_dim 64 form.
template <int OUT16, bool UNI, int NW, bool BF8, int D = 128>
__global__ void __launch_bounds__(NW * 64, (NW == 8 || D == 256) ? 1 : 2) batch_prefill_fp8_kernel(const PrefillKernelParams p) {
  static_assert(D == 128 || ((D == 64 || D == 256) && NW == 4), "head_dim 128, or 64 / 256 with four waves");
  constexpr int kF8KTile = F8Lds<D>::kTile, kF8VOff = F8Lds<D>::kVOff, kF8TabOff = F8Lds<D>::kTabOff;
  constexpr int kF8IdsOff = F8Lds<D>::kIdsOff, kF8Smem = F8Lds<D>::kSmem;
  constexpr int ROWB = D;             // bytes per K / V row in LDS
  constexpr int SLOTS = ROWB / 16;    // 16-byte slots per row
  constexpr int KBLK = 32 * ROWB;     // byte offset of the second 32-row block of a tile
  constexpr int TRR = 16 * ROWB;      // row step of the four transposed V reads
  constexpr int kThreads = NW * 64;
  constexpr int kTQ = NW * 32;  // packed query rows per workgroup
  constexpr float kLog2Scale = BF8 ? kBF8Log2Scale : kF8Log2Scale;
  constexpr float kPMax = BF8 ? 57344.f : 448.f;  // largest value of the type P is rounded to
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  constexpr int DBLK = D / 32;
  // ONE static array for every LDS object: the compiler separates an LDS-DMA target from an LDS read by
  // constant offsets (and index ranges) inside one object; with a second object, or unbounded indices, it
  // puts s_waitcnt vmcnt(0) in front of the reads and the prefetch is gone
  __shared__ __attribute__((aligned(1024))) char smem[kF8Smem + ((FI_PF8_KO & 64) ? 40960 : 0)];

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
  const bool split = p.kv_tile_indices != nullptr || p.num_kv_chunks > 1;
  if (p.request_indices) {
    req = p.request_indices[work];
    q_tile = p.qo_tile_indices[work];
    if (req < 0) return;
    if (p.kv_tile_indices) kv_chunk = p.kv_tile_indices[work];
  } else if (p.num_kv_chunks > 1) {
    q_tile = work / p.num_kv_chunks;
    kv_chunk = work - q_tile * p.num_kv_chunks;
  }
  int qo_start = 0, qo_len, kv_len, page_begin = 0, num_pages = 0;
  if (p.qo_indptr) {
    qo_start = p.qo_indptr[req];
    qo_len = p.qo_indptr[req + 1] - qo_start;
  } else {
    qo_len = p.single_qo_len;
  }
  if (p.kv_indptr) {
    page_begin = p.kv_indptr[req];
    num_pages = p.kv_indptr[req + 1] - page_begin;
    kv_len = p.kv_last_page_len ? (num_pages > 0 ? (num_pages - 1) * p.page_size + p.kv_last_page_len[req] : 0)
                                : num_pages * p.page_size;
  } else {
    kv_len = p.single_kv_len;
    num_pages = (kv_len + p.page_size - 1) / p.page_size;
  }
  const int G = p.group_size;
  const int packed_len = qo_len * G;
  // packed row (qo_idx * G + head) of this lane inside the workgroup's 128-row tile
  const int row_in_tile = UNI ? ((wave / G) * 32 + lq) * G + (wave % G) : wave * 32 + lq;
  const int row0 = q_tile * kTQ + (UNI ? (wave / G) * 32 * G : wave * 32);  // the wave's first packed row
  const int pr = q_tile * kTQ + row_in_tile;
  const bool row_valid = pr < packed_len;
  const int prc = row_valid ? pr : (packed_len > 0 ? packed_len - 1 : 0);
  const int qo_idx = (int)fast_div((uint32_t)prc, p.group_div);
  const int hg = prc - qo_idx * G;
  const int qo_head = kv_head * G + hg;
  const int q_pos = kv_len - qo_len + qo_idx;

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q, h) holds bytes [64 kk + 32 h, +32) of its row ----
  constexpr int KK = D / 64;  // k steps of QK^T
  i32x8 qf[KK < 2 ? 2 : KK];
  qf[1] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
  {
    const uint8_t* qrow = (const uint8_t*)p.q + (int64_t)(qo_start + qo_idx) * p.q_stride_n +
                          (int64_t)qo_head * p.q_stride_h;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const u32x4 lo = *(const u32x4*)(qrow + 64 * kk + 32 * lh);
      const u32x4 hi = *(const u32x4*)(qrow + 64 * kk + 32 * lh + 16);
      qf[kk] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    }
  }

  float qk_scale = p.sm_scale;
  if (p.scale_q) qk_scale *= p.scale_q[qo_head];
  if (p.scale_k) qk_scale *= p.scale_k[kv_head];
  const float c_lane = qk_scale * kLog2e;
  // wave-uniform copy (valid with UNI): a scalar register
  const float c_uni = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, c_lane)));

  int kv_end = kv_len;
  if (p.causal) {
    const int last_pr = min(q_tile * kTQ + kTQ, packed_len) - 1;
    const int last_qo = last_pr >= 0 ? (int)fast_div((uint32_t)last_pr, p.group_div) : 0;
    kv_end = min(kv_len, max(0, kv_len - qo_len + last_qo + 1));
  }
  int kv_begin = 0;
  if (split) {
    const int kv_chunk_size = p.kv_chunk_size_ptr ? *p.kv_chunk_size_ptr : p.kv_chunk_size;
    kv_begin = kv_chunk * kv_chunk_size;
    kv_end = min(kv_end, kv_begin + kv_chunk_size);
  }
  const int tile_base = kv_begin / kTileKV;
  const int num_tiles = kv_end > kv_begin ? (kv_end - kv_begin + kTileKV - 1) / kTileKV : 0;
  const int vis_hi_raw = p.causal ? min(kv_len - 1, q_pos) : kv_len - 1;
  const int vis_lo = vis_hi_raw < 0 ? 0x40000000 : 0;
  const int vis_hi = vis_hi_raw < 0 ? 0x40000000 : vis_hi_raw;
  const int first_qo_wave = (int)fast_div((uint32_t)min(row0, max(packed_len - 1, 0)), p.group_div);
  const int min_qpos_wave = kv_len - qo_len + first_qo_wave;
  // a tile needs the mask code when it reaches past kv_len or past the smallest query position of the wave
  auto tile_needs_mask = [&](int t) {
    const int tile0 = (tile_base + t) * kTileKV;
    return (tile0 + kTileKV > kv_len) || (p.causal && tile0 + kTileKV - 1 > min_qpos_wave);
  };

  // ---- lane constants of the LDS images (same swizzles as the first structure) ----
  // DMA: thread -> (row = tid / 8 [+ 32], 16-byte slot tid % 8); the slot receives global chunk slot ^ swizzle(row)
  // (D = 64: four slots per row, a piece is 16 rows, 256 threads cover the 64 rows in one pass; K rows are
  // swizzled by (row >> 2) & 3 -- the 16 rows one ds_read_b128 lane group touches then fill 256 bytes of banks -- and
  // V rows by bit 3 of the row on slot bit 1, which does the same for the 8 rows x 32 bytes of a transposed read)
  const int st_row = tid / SLOTS, st_slot = tid % SLOTS;
  // (D = 256: a row is a whole 256-byte bank row, so the 16 rows of a ds_read_b128 lane group need all 16 chunk
  // positions -- chunk ^ (row & 15) -- and the 8 rows x 32 bytes of a transposed read, rows {0..3, 8..11} (+ 4), take
  // row bits 0, 1 and 3 onto chunk bits 1..3)
  auto k_swz = [](int row) { return D == 256 ? (row & 15) : D == 128 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  auto v_swz = [](int row) {
    return D == 256 ? (((row & 3) | (((row >> 3) & 1) << 2)) << 1)
         : D == 128 ? ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2))
                    : (((row >> 3) & 1) << 1);
  };
  const int kc = (st_slot ^ k_swz(st_row)) << 4;
  const int vc = (st_slot ^ v_swz(st_row)) << 4;
  const int64_t head_off = (int64_t)kv_head * p.kv_stride_h;
  const char* const k_thr = (const char*)p.k + head_off + kc;
  const char* const v_thr = (const char*)p.v + head_off + vc;
  const uint32_t stride_page32 = (uint32_t)p.kv_stride_page, stride_n32 = (uint32_t)p.kv_stride_n;
  // K fragment (A operand): row lq (+32 kb), chunks 4 kk + 2 lh + e, chunk ^ ((row >> 1) & 7)
  int k_rd[KK < 2 ? 2 : KK][2];
#pragma unroll
  for (int kk = 0; kk < (KK < 2 ? 2 : KK); ++kk)
#pragma unroll
    for (int e = 0; e < 2; ++e)
      k_rd[kk][e] = D == 64 ? lq * 64 + (((2 * lh + e) ^ k_swz(lq)) << 4)
                            : lq * ROWB + (((4 * kk + 2 * lh + e) ^ k_swz(lq)) << 4);
  // V^T fragment.  P is the B operand in accumulator order (lane (q, h) holds the 32 probabilities kv = 32 kb + 8 g
  // + 4 h + e, kb < 2, g < 4, e < 4), so V must be the A operand with that k order along each head_dim row.  The V
  // tile stays ROW-MAJOR in LDS and is transposed on the way out by ds_read_b64_tr_b8: in a 16-lane group, lanes 2b
  // and 2b + 1 address the 16 bytes of "row b" (any row) and lane j receives byte j of rows 0..7 -- each lane
  // gathers, per read, the 8 kv rows of its k order for its own head_dim column: lane i of a group addresses row
  // b = i >> 1, bytes 8 (i & 1) .. +8 of chunk 2 db + g
  int v_rd[DBLK < 4 ? 4 : DBLK];  // [db]; head_dim 64 uses the first two
  {
    const int i16 = lane & 15, g = (lane >> 4) & 1, b = i16 >> 1;
    const int row = 4 * lh + (b & 3) + 8 * (b >> 2);
    const int sw = v_swz(row);
    const int base = row * ROWB + ((g ^ sw) << 4) + 8 * (i16 & 1);
#pragma unroll
    for (int db = 0; db < (DBLK < 4 ? 4 : DBLK); ++db) v_rd[db] = base ^ ((db & (DBLK - 1)) << 5);  // ring / stage / row-block offsets: immediates
  }
  uint64_t* const tab = (uint64_t*)(smem + kF8TabOff);
  int32_t* const ids = (int32_t*)(smem + kF8IdsOff);

  // ---- running state ----
  f32x16 o_acc[DBLK < 4 ? 4 : DBLK];  // [db]; head_dim 64 uses the first two (the others are dead and cost nothing)
#pragma unroll
  for (int db = 0; db < (DBLK < 4 ? 4 : DBLK); ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[db][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;
  float m_adj = m_run - kLog2Scale;  // exp2 argument offset: P x 448 / 2^kF8Thr = 2^(c s - m_adj)

  if (num_tiles > 0) {
    // ---- page ids of this kv range -> LDS (refilled when the walk leaves the window) ----
    int ids_base = (int)fast_div((uint32_t)min(tile_base * kTileKV, max(kv_len - 1, 0)), p.page_div);
    auto fill_ids = [&]() {
      if (p.kv_indices) {
        for (int i = tid; i < kF8Ids; i += kThreads) {
          const int pg = ids_base + i;
          ids[i] = pg < num_pages ? p.kv_indices[page_begin + pg] : 0;
        }
      }
    };
    // byte offsets of the 64 rows of tile `t_rel` (clamped to the last tile): one row per lane
    auto make_tab = [&](int t_rel, int slot) {
      const int tile = tile_base + min(t_rel, num_tiles - 1);
      const int kvi = max(min(tile * kTileKV + lane, kv_len - 1), 0);
      const int pi = (int)fast_div((uint32_t)kvi, p.page_div);
      const int en = kvi - pi * p.page_size;
      const int pg = p.kv_indices ? ids[(pi - ids_base) & (kF8Ids - 1)] : page_begin + pi;  // index range visible
      tab[slot * kTileKV + lane] = (uint64_t)(uint32_t)pg * stride_page32 + (uint64_t)(uint32_t)en * stride_n32;
    };
    // page window check for the tile whose table is made next (uniform)
    auto ids_cover = [&](int t_rel) {
      const int tile = tile_base + min(t_rel, num_tiles - 1);
      const int k0 = max(min(tile * kTileKV, kv_len - 1), 0), k1 = max(min(tile * kTileKV + kTileKV - 1, kv_len - 1), 0);
      const int p0 = (int)fast_div((uint32_t)k0, p.page_div), p1 = (int)fast_div((uint32_t)k1, p.page_div);
      return p0 >= ids_base && p1 - ids_base < kF8Ids;
    };
    // DMA of a tile's K and V rows into ring stage `stage`: the row offsets are read from table slot `slot` first
    // (dma_offsets, early in a step) and the four pieces are issued later (dma_issue), so that the LDS latency of
    // the table read does not sit at the top of the step
    // a pass = the rows the workgroup's threads cover with one 16-byte chunk each (NW * 64 / SLOTS rows); the tile's
    // 64 rows take PASSES of them (head_dim 256: four), one K and one V piece per wave and pass
    constexpr int kRowsPerPass = kThreads / SLOTS;
    constexpr int PASSES = kTileKV / kRowsPerPass;
    static_assert(PASSES == 1 || PASSES == 2 || PASSES == 4, "passes per tile");
    struct DmaOffs { uint64_t o[PASSES]; };
    auto dma_offsets = [&](int slot, DmaOffs& f) {
#pragma unroll
      for (int j = 0; j < PASSES; ++j) f.o[j] = tab[slot * kTileKV + st_row + j * kRowsPerPass];
    };
    auto dma_issue = [&](int stage, const DmaOffs& f) {
      char* const kdst = smem + stage * kF8KTile + wave * 1024;
      char* const vdst = kdst + kF8VOff;
#pragma unroll
      for (int j = 0; j < PASSES; ++j)
        __builtin_amdgcn_global_load_lds((f8_gbl_void*)(k_thr + f.o[j]), (f8_lds_void*)(kdst + j * NW * 1024), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < PASSES; ++j)
        __builtin_amdgcn_global_load_lds((f8_gbl_void*)(v_thr + f.o[j]), (f8_lds_void*)(vdst + j * NW * 1024), 16, 0, 0);
    };
    auto dma_tile = [&](int slot, int stage) {
      DmaOffs f;
      dma_offsets(slot, f);
      dma_issue(stage, f);
    };

    fill_ids();
    __syncthreads();
    if (wave < 4) make_tab(wave, wave);  // tables of tiles 0..3 (clamped)
    __syncthreads();
    dma_tile(0, 0);
    dma_tile(1, 1);
    dma_tile(2, 2);
    // tiles 0 and 1 have landed (tile 2: this wave's 4 (NW = 8: 2) pieces in flight); wait + barrier as one
    // statement (see the step)
    // (LDS-DMA pieces per wave and tile: 2 * PASSES)
    if constexpr (PASSES == 4) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else if constexpr (PASSES == 2) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");

    // ---- building blocks ----
    const i32x8 q0 = qf[0], q1 = qf[1];
    auto k_frag2 = [&](const char* kb, int kbk, int kk) {  // 32-key block kbk, k step kk of the K tile at kb
      const u32x4 lo = *(const u32x4*)(kb + kbk * KBLK + k_rd[kk][0]);
      const u32x4 hi = *(const u32x4*)(kb + kbk * KBLK + k_rd[kk][1]);
      return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    };
    auto k_frag = [&](const char* kb, int i) {  // fragment i = 2 kbk + kk of the K tile at kb
      const u32x4 lo = *(const u32x4*)(kb + (i >> 1) * KBLK + k_rd[i & 1][0]);
      const u32x4 hi = *(const u32x4*)(kb + (i >> 1) * KBLK + k_rd[i & 1][1]);
      return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    };
    auto apply_mask = [&](int t_rel, f32x16 (&s)[2]) {
      const int tile0 = (tile_base + t_rel) * kTileKV;
      const unsigned span = (unsigned)(vis_hi - vis_lo);
      const int base_idx = tile0 + 4 * lh - vis_lo;
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned rel = (unsigned)(base_idx + 32 * kbk + (r & 3) + 8 * (r >> 2));
          s[kbk][r] = rel <= span ? s[kbk][r] : -INFINITY;
        }
    };
    // exact row maximum of a score tile (rare path only)
    auto row_max = [&](const f32x16 (&s)[2]) {
#define FI_F8_MAX8(v, r) fmaxf(fmaxf(fmaxf(v[r], v[r + 1]), fmaxf(v[r + 2], v[r + 3])), fmaxf(fmaxf(v[r + 4], v[r + 5]), fmaxf(v[r + 6], v[r + 7])))
      const float mx = fmaxf(fmaxf(FI_F8_MAX8(s[0], 0), FI_F8_MAX8(s[0], 8)), fmaxf(FI_F8_MAX8(s[1], 0), FI_F8_MAX8(s[1], 8)));
#undef FI_F8_MAX8
      return fmaxf(mx, swap_halves(mx));
    };
    // Deferred reference exponent WITHOUT a per-tile row maximum: the probabilities of a tile are formed against
    // the current exponent m_run as P'' = 2^(c s - m_run) x 448 / 2^kF8Thr; e4m3 holds them as long as P'' <= 448.
    // The sums over 8 probabilities (needed for the row sum anyway) bound every term, so "some chunk sum > 448" is
    // the (conservative) overflow test.  Only then -- the first tile of a row, or a row maximum that grew by more
    // than ~2^kF8Thr -- the exact maximum of the tile is taken, O and the row sum are rescaled to it (rescale_to)
    // and the tile's probabilities are formed again.
    auto rescale_to = [&](float m_true) {
      const float alpha = fast_exp2(m_run - m_true);
      m_run = m_true;
      m_adj = m_true - kLog2Scale;
      l_run *= alpha;
      // whole-vector statements, no loop: in a block the compiler treats as cold a loop may stay rolled, and a
      // runtime index would push the accumulators into scratch memory for the whole kernel
      o_acc[0] = o_acc[0] * alpha;
      o_acc[1] = o_acc[1] * alpha;
      if constexpr (D >= 128) {
        o_acc[2] = o_acc[2] * alpha;
        o_acc[3] = o_acc[3] * alpha;
      }
      if constexpr (D == 256) {
        o_acc[4] = o_acc[4] * alpha;
        o_acc[5] = o_acc[5] * alpha;
        o_acc[6] = o_acc[6] * alpha;
        o_acc[7] = o_acc[7] * alpha;
      }
    };
    const float c_exp = UNI ? c_uni : c_lane;
    // exp2 argument c s - m_adj.  UNI: c from a SCALAR register, so the instruction reads two vector registers
    // and issues at full rate
    // (the readfirstlane is an asm statement: as a builtin the compiler pushes it through the multiplications that
    // form c, and the product -- there is no scalar f32 multiply -- ends up in a vector register again)
    float c_sreg = c_lane;
    if constexpr (UNI) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(c_sreg) : "v"(c_lane));
    auto exp_arg = [&](float sv) { return __builtin_fmaf(sv, UNI ? c_sreg : c_lane, -m_adj); };
    // exp2 of registers r0 .. r0 + 7 of a score block: their sum and their e4m3 image (two words of the B operand)
    auto exp_chunk = [&](const f32x16& s, int r0, int& w0, int& w1, float& cs) {
#define FI_F8_EXP(i) ((FI_PF8_KO & 1) ? exp_arg(s[r0 + i]) : fast_exp2(exp_arg(s[r0 + i])))
      const float x[8] = {FI_F8_EXP(0), FI_F8_EXP(1), FI_F8_EXP(2), FI_F8_EXP(3), FI_F8_EXP(4), FI_F8_EXP(5), FI_F8_EXP(6), FI_F8_EXP(7)};
#undef FI_F8_EXP
      // two chains of single adds (packed f32 adds beside MFMAs cost more than the pairs they replace; the
      // empty asm keeps the chain inside this chunk's MFMA gap instead of being sunk to the end of the tile)
      float e = x[0] + x[1], o = x[2] + x[3];
      e += x[4];
      o += x[5];
      e += x[6];
      o += x[7];
      cs = e + o;
      asm volatile("" : "+v"(cs));
      // both halves of each word are written by the two conversions: the initial content is don't-care
      int u0, u1;  // volatile: identical empty statements would otherwise be merged into one value (and copied)
      asm volatile("" : "=v"(u0));
      asm volatile("" : "=v"(u1));
      if constexpr (BF8) {
        w0 = __builtin_amdgcn_cvt_pk_bf8_f32(x[2], x[3], __builtin_amdgcn_cvt_pk_bf8_f32(x[0], x[1], u0, false), true);
        w1 = __builtin_amdgcn_cvt_pk_bf8_f32(x[6], x[7], __builtin_amdgcn_cvt_pk_bf8_f32(x[4], x[5], u1, false), true);
      } else {
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], u0, false), true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[6], x[7], __builtin_amdgcn_cvt_pk_fp8_f32(x[4], x[5], u1, false), true);
      }
    };

    // ---- prologue: S^T of tile 0, its mask and row maximum ----
    f32x16 s_a[2], s_b[2];
    {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if constexpr (D == 128) {
        s_a[0] = mfma_fp8_k64_fmt<BF8>(k_frag(smem, 1), q1, mfma_fp8_k64_fmt<BF8>(k_frag(smem, 0), q0, zero));
        s_a[1] = mfma_fp8_k64_fmt<BF8>(k_frag(smem, 3), q1, mfma_fp8_k64_fmt<BF8>(k_frag(smem, 2), q0, zero));
      } else {
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
          s_a[kbk] = mfma_fp8_k64_fmt<BF8>(k_frag2(smem, kbk, 0), qf[0], zero);
#pragma unroll
          for (int kk = 1; kk < KK; ++kk) s_a[kbk] = mfma_fp8_k64_fmt<BF8>(k_frag2(smem, kbk, kk), qf[kk], s_a[kbk]);
        }
      }
      if (tile_needs_mask(0)) apply_mask(0, s_a);
    }

    // One pipeline step: tile t (scores in sc) is finished while the scores of tile t+1 are produced in sn.
    // ST = t & 3 (ring stage of tile t), MASK: tile t+1 takes the mask code.  A wave issues in order, so the
    // vector work is placed BETWEEN the MFMAs in program order (sched_barrier keeps the groups apart):
    //   region A: 4 x { K fragment of the next MFMA, QK^T MFMA of tile t+1, exp2 / sum / e4m3 of 8 scores of tile t }
    //   region B: 4 x { V^T fragment two MFMAs ahead, P.V MFMA of tile t, running maximum over 8 scores of tile t+1 }
    auto step = [&](auto stage_c, auto mask_c, f32x16 (&sc)[2], f32x16 (&sn)[2], const int t) {
      constexpr int ST = decltype(stage_c)::value;
      constexpr bool MASK = decltype(mask_c)::value;
      // table of tile t+4 -> the slot tile t's table used (read for the last time at step t-3)
      if (wave == ST) make_tab(t + 4, ST);
      if constexpr (D != 128) {
        // head_dim 64 / 256: the same step, plainly written (builtin MFMAs, the compiler's schedule between the
        // fences) -- 2 KK QK^T MFMAs of tile t+1 with the four exp2 chunks of tile t between them, then DBLK P.V MFMAs
        DmaOffs f;
        dma_offsets((ST + 3) & 3, f);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const char* const kb = smem + ((ST + 1) & 3) * kF8KTile;
        constexpr int VB = kF8VOff + ST * kF8KTile;
        int p8w[8];
        float cs0, cs1, cs2, cs3;
        __builtin_amdgcn_sched_barrier(0);
        exp_chunk(sc[0], 0, p8w[0], p8w[1], cs0);
        dma_issue((ST + 3) & 3, f);
        __builtin_amdgcn_sched_barrier(0);
        sn[0] = mfma_fp8_k64_fmt<BF8>(k_frag2(kb, 0, 0), qf[0], zero);
#pragma unroll
        for (int kk = 1; kk < KK; ++kk) sn[0] = mfma_fp8_k64_fmt<BF8>(k_frag2(kb, 0, kk), qf[kk], sn[0]);
        exp_chunk(sc[0], 8, p8w[2], p8w[3], cs1);
        __builtin_amdgcn_sched_barrier(0);
        sn[1] = mfma_fp8_k64_fmt<BF8>(k_frag2(kb, 1, 0), qf[0], zero);
#pragma unroll
        for (int kk = 1; kk < KK; ++kk) sn[1] = mfma_fp8_k64_fmt<BF8>(k_frag2(kb, 1, kk), qf[kk], sn[1]);
        exp_chunk(sc[1], 0, p8w[4], p8w[5], cs2);
        exp_chunk(sc[1], 8, p8w[6], p8w[7], cs3);
        __builtin_amdgcn_sched_barrier(0);
        if (__builtin_expect(__any(!(fmaxf(fmaxf(cs0, cs1), fmaxf(cs2, cs3)) <= kPMax)), 0)) {
          rescale_to(fmaxf(m_run, row_max(sc) * c_exp));
          exp_chunk(sc[0], 0, p8w[0], p8w[1], cs0);
          exp_chunk(sc[0], 8, p8w[2], p8w[3], cs1);
          exp_chunk(sc[1], 0, p8w[4], p8w[5], cs2);
          exp_chunk(sc[1], 8, p8w[6], p8w[7], cs3);
        }
        l_run += (cs0 + cs1) + (cs2 + cs3);
        const i32x8 p8 = {p8w[0], p8w[1], p8w[2], p8w[3], p8w[4], p8w[5], p8w[6], p8w[7]};
        // V^T fragments two head_dim blocks at a time (the stage base goes into the address register: at head_dim 256
        // the V ring starts beyond the 16-bit offset field of the LDS instructions)
        // (one workgroup per CU at head_dim 256, i.e. one wave per SIMD: the pair after next is read BEFORE this pair's
        // MFMAs are issued, otherwise every pair stands through an LDS round trip with the matrix pipe empty)
        i32x2 vfa[2][4], vfb[2][4];
        auto v_pair_read = [&](int db, int slot) {
          const int a0 = v_rd[db] + VB, a1 = v_rd[db + 1] + VB;
          vfa[slot][0] = lds_tr8<0 * TRR>(a0);
          vfa[slot][1] = lds_tr8<1 * TRR>(a0);
          vfa[slot][2] = lds_tr8<2 * TRR>(a0);
          vfa[slot][3] = lds_tr8<3 * TRR>(a0);
          vfb[slot][0] = lds_tr8<0 * TRR>(a1);
          vfb[slot][1] = lds_tr8<1 * TRR>(a1);
          vfb[slot][2] = lds_tr8<2 * TRR>(a1);
          vfb[slot][3] = lds_tr8<3 * TRR>(a1);
        };
        v_pair_read(0, 0);
#pragma unroll
        for (int db = 0; db < DBLK; db += 2) {
          const int cur = (db >> 1) & 1;
          i32x2(&va)[4] = vfa[cur];
          i32x2(&vb)[4] = vfb[cur];
          if (db + 2 < DBLK) {
            v_pair_read(db + 2, cur ^ 1);
            asm volatile("s_waitcnt lgkmcnt(8)"
                         : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]), "+v"(vb[0]), "+v"(vb[1]), "+v"(vb[2]), "+v"(vb[3]));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]), "+v"(vb[0]), "+v"(vb[1]), "+v"(vb[2]), "+v"(vb[3]));
          }
          __builtin_amdgcn_sched_barrier(0);  // an MFMA is no memory operation: keep it behind the wait (guide 5.4 rule 18)
          o_acc[db] = mfma_fp8_k64_fmt<BF8>(
              i32x8{va[0][0], va[0][1], va[1][0], va[1][1], va[2][0], va[2][1], va[3][0], va[3][1]}, p8, o_acc[db]);
          o_acc[db + 1] = mfma_fp8_k64_fmt<BF8>(
              i32x8{vb[0][0], vb[0][1], vb[1][0], vb[1][1], vb[2][0], vb[2][1], vb[3][0], vb[3][1]}, p8, o_acc[db + 1]);
        }
        if constexpr (MASK) apply_mask(t + 1, sn);
        // in flight afterwards: V of tile t+2 and both operands of tile t+3 = 3 PASSES pieces of this wave
        if constexpr (PASSES == 4) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else {
      // row offsets of tile t+3 (its DMA is issued inside region A, after the table read has returned)
      DmaOffs f;
      dma_offsets((ST + 3) & 3, f);
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const char* const kb = smem + ((ST + 1) & 3) * kF8KTile;  // tile t+1 (past the end: the last tile again)
      int p8w[8];
      float cs0, cs1, cs2, cs3;
      // V^T fragments of the first two P.V MFMAs: V of tile t landed before the last barrier, so they are read
      // now and wait in registers through region A
      constexpr int VB = kF8VOff + ST * kF8KTile;  // V tile of stage ST; transposed reads r = 0..3 at rows 0 / 16 / 32 / 48
      i32x2 va[4], vb[4];
#define FI_F8_VREAD(dst, db)                        \
  dst[0] = lds_tr8<VB + 0 * TRR>(v_rd[db]);        \
  dst[1] = lds_tr8<VB + 1 * TRR>(v_rd[db]);        \
  dst[2] = lds_tr8<VB + 2 * TRR>(v_rd[db]);        \
  dst[3] = lds_tr8<VB + 3 * TRR>(v_rd[db]);
#define FI_F8_VWAIT(n, w) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]))
#define FI_F8_VFRAG(w) (i32x8{w[0][0], w[0][1], w[1][0], w[1][1], w[2][0], w[2][1], w[3][0], w[3][1]})
      FI_F8_VREAD(va, 0)
      FI_F8_VREAD(vb, 1)
      // ---- region A ----
      // group g: MFMA g-1 of QK^T (tile t+1) first, then the K fragment two MFMAs ahead, then the exp2 chunk g
      // of tile t; the first chunk runs under the LDS latency of the first two fragments
#define FI_F8_GROUP(nds)                                         \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             \
  __builtin_amdgcn_sched_group_barrier(0x100, nds, 0);           \
  __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);             \
  __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);             \
  __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);             \
  __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);             \
  __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);            \
  __builtin_amdgcn_sched_barrier(0);
      int unit = 0x7F7F7F7F;
      asm volatile("" : "+v"(unit));
#if FI_PF8_ASM_MFMA
#define FI_F8_QK0(dst, a, b) if (!(FI_PF8_KO & 16)) mfma_fp8_k64_asm_zero<BF8>(dst, a, b, unit)
#define FI_F8_QK1(dst, a, b) if (!(FI_PF8_KO & 16)) mfma_fp8_k64_asm<BF8>(dst, a, b, unit)
#define FI_F8_PV(dst, a, b) if (!(FI_PF8_KO & 8)) mfma_fp8_k64_asm<BF8>(dst, a, b, unit)
#else
#define FI_F8_QK0(dst, a, b) if (!(FI_PF8_KO & 16)) dst = mfma_fp8_k64_fmt<BF8>(a, b, zero)
#define FI_F8_QK1(dst, a, b) if (!(FI_PF8_KO & 16)) dst = mfma_fp8_k64_fmt<BF8>(a, b, dst)
#define FI_F8_PV(dst, a, b) if (!(FI_PF8_KO & 8)) dst = mfma_fp8_k64_fmt<BF8>(a, b, dst)
#endif
      i32x8 kf0 = k_frag(kb, 0);
      i32x8 kf1 = k_frag(kb, 1);
      __builtin_amdgcn_sched_barrier(0);  // every LDS read of the step's head is in flight before the first chunk
      exp_chunk(sc[0], 0, p8w[0], p8w[1], cs0);
      // new rows for the ring: tile t+3 into the stage tile t-1 left (every wave passed the barrier of t-1)
      if (!(FI_PF8_KO & 2)) dma_issue((ST + 3) & 3, f);
      __builtin_amdgcn_sched_barrier(0);
      FI_F8_QK0(sn[0], kf0, q0);
      kf0 = k_frag(kb, 2);
      exp_chunk(sc[0], 8, p8w[2], p8w[3], cs1);
      FI_F8_GROUP(2)
      FI_F8_QK1(sn[0], kf1, q1);
      kf1 = k_frag(kb, 3);
      exp_chunk(sc[1], 0, p8w[4], p8w[5], cs2);
      FI_F8_GROUP(2)
      FI_F8_QK0(sn[1], kf0, q0);
      exp_chunk(sc[1], 8, p8w[6], p8w[7], cs3);
      FI_F8_GROUP(0)
#undef FI_F8_GROUP
      FI_F8_QK1(sn[1], kf1, q1);
      if (FI_PF8_KO & 16) asm volatile("" : "+v"(kf0), "+v"(kf1));
      __builtin_amdgcn_sched_barrier(0);
      if (__builtin_expect(!(FI_PF8_KO & 32) && __any(!(fmaxf(fmaxf(cs0, cs1), fmaxf(cs2, cs3)) <= kPMax)), 0)) {
        // rare: some probability may have left the e4m3 range -- exact maximum of tile t, move the exponent, again
        rescale_to(fmaxf(m_run, row_max(sc) * c_exp));
        exp_chunk(sc[0], 0, p8w[0], p8w[1], cs0);
        exp_chunk(sc[0], 8, p8w[2], p8w[3], cs1);
        exp_chunk(sc[1], 0, p8w[4], p8w[5], cs2);
        exp_chunk(sc[1], 8, p8w[6], p8w[7], cs3);
      }
      l_run += (cs0 + cs1) + (cs2 + cs3);
      const i32x8 p8 = {p8w[0], p8w[1], p8w[2], p8w[3], p8w[4], p8w[5], p8w[6], p8w[7]};
      // ---- region B ----  (the fourth QK^T MFMA runs under the wait for the first V^T fragment)
      FI_F8_VWAIT(4, va);
      { const i32x8 vf = FI_F8_VFRAG(va); FI_F8_PV(o_acc[0], vf, p8); }
      FI_F8_VREAD(va, 2)
      __builtin_amdgcn_sched_barrier(0);
      FI_F8_VWAIT(4, vb);
      { const i32x8 vf = FI_F8_VFRAG(vb); FI_F8_PV(o_acc[1], vf, p8); }
      FI_F8_VREAD(vb, 3)
      __builtin_amdgcn_sched_barrier(0);
      FI_F8_VWAIT(4, va);
      { const i32x8 vf = FI_F8_VFRAG(va); FI_F8_PV(o_acc[2], vf, p8); }
      __builtin_amdgcn_sched_barrier(0);
      FI_F8_VWAIT(0, vb);
      { const i32x8 vf = FI_F8_VFRAG(vb); FI_F8_PV(o_acc[3], vf, p8); }
      if (FI_PF8_KO & 8) asm volatile("" :: "v"(p8), "v"(va[0]), "v"(vb[0]));
      if constexpr (MASK) apply_mask(t + 1, sn);
#undef FI_F8_VREAD
#undef FI_F8_VWAIT
#undef FI_F8_VFRAG
#undef FI_F8_QK0
#undef FI_F8_QK1
#undef FI_F8_PV
      // K of tile t+2 (and everything older, V of tile t+1 included) of this wave's pieces landed; then the
      // workgroup barrier.  ONE asm statement: the s_barrier builtin alone is no memory fence for the compiler,
      // which would hoist the next step's LDS reads between the wait and the barrier.
      // (in flight afterwards: V of tile t+2 and both operands of tile t+3 -- 6 pieces, 3 with NW = 8)
      if (!(FI_PF8_KO & 4)) {
        if constexpr (PASSES == 2) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      }  // D == 128
    };
    auto refill_if_needed = [&](int t_first, int t_last) {
      if (p.kv_indices && !(ids_cover(t_first + 4) && ids_cover(t_last + 4))) {  // uniform; rare
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int tile = tile_base + min(t_first + 4, num_tiles - 1);
        ids_base = (int)fast_div((uint32_t)max(min(tile * kTileKV, kv_len - 1), 0), p.page_div);
        fill_ids();
        __syncthreads();
      }
    };
    using std::integral_constant;
    using std::false_type;
    using std::true_type;
    // unmasked prefix: tiles 1 .. n_plain need no mask code (both conditions of tile_needs_mask are monotone in the
    // tile index), so steps 0 .. n_plain - 1 run the plain body; found once, the loop conditions are one compare
    int n_plain = 0;
    while (n_plain + 1 < num_tiles && !tile_needs_mask(n_plain + 1)) ++n_plain;
    const int bulk_end = n_plain & ~3;  // four plain steps per trip
    // page-id window: the first step whose table (tile t + 4) leaves the window, checked once per trip
    int t = 0;
    for (; t < bulk_end; t += 4) {
      refill_if_needed(t, t + 3);
      step(integral_constant<int, 0>{}, false_type{}, s_a, s_b, t);
      step(integral_constant<int, 1>{}, false_type{}, s_b, s_a, t + 1);
      step(integral_constant<int, 2>{}, false_type{}, s_a, s_b, t + 2);
      step(integral_constant<int, 3>{}, false_type{}, s_b, s_a, t + 3);
    }
    // remaining tiles (the diagonal / the partial last tile): mask code always in; the stage stays a compile-time
    // constant (t is a multiple of 4 here)
    for (; t < num_tiles; t += 4) {
      refill_if_needed(t, t + 3);
      step(integral_constant<int, 0>{}, true_type{}, s_a, s_b, t);
      if (t + 1 < num_tiles) step(integral_constant<int, 1>{}, true_type{}, s_b, s_a, t + 1);
      if (t + 2 < num_tiles) step(integral_constant<int, 2>{}, true_type{}, s_a, s_b, t + 2);
      if (t + 3 < num_tiles) step(integral_constant<int, 3>{}, true_type{}, s_b, s_a, t + 3);
    }
    // the clamped extra tiles still land in the ring; the s_nops cover the last P.V MFMAs (asm statements: the
    // compiler does not pad their result latency) before the epilogue reads the accumulators
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  }

  // ---- finalize: l_run carries the constant factor of P8, so O / l_run is free of it ----
  l_run += swap_halves(l_run);
  const bool empty = !(l_run > 0.f);
  float inv = empty ? 0.f : 1.0f / l_run;
  if (p.scale_v) inv *= p.scale_v[kv_head];
  const float lse_val = empty ? FI_NEG_INF : m_run + fast_log2(l_run) - kLog2Scale;
  if (split) {
    if (row_valid) {
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
      if (lh == 0) p.tmp_lse[entry * p.num_qo_heads + qo_head] = lse_val;
    }
    return;
  }
  // Output through LDS: the wave's 32 x 128 block as 16-bit rows of 256 bytes (16-byte chunks XOR-swizzled
  // by the row), read back as whole rows -- 16 bytes per lane, 4 rows per store instruction.  Four packed
  // rows of one token are adjacent heads, i.e. adjacent 256-byte rows of the output tensor.
  {
    char* const region = smem + wave * (D > 128 ? 64 * D : 8192);  // the K ring is idle now
    int64_t* const rowtab = (int64_t*)(smem + kF8TabOff) + wave * 32;
    if (lh == 0)
      rowtab[lq] = row_valid ? ((int64_t)(qo_start + qo_idx) * p.num_qo_heads + qo_head) * D : (int64_t)-1;
#pragma unroll
    for (int db = 0; db < DBLK; ++db) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const uint32_t w0 = pack2<OUT16>(o_acc[db][4 * r4 + 0] * inv, o_acc[db][4 * r4 + 1] * inv);
        const uint32_t w1 = pack2<OUT16>(o_acc[db][4 * r4 + 2] * inv, o_acc[db][4 * r4 + 3] * inv);
        *(u32x2*)(region + lq * (2 * D) + (((4 * db + r4) ^ (lq & (D / 8 - 1))) << 4) + 8 * lh) = u32x2{w0, w1};
      }
    }
    __builtin_amdgcn_wave_barrier();  // LDS operations of one wave execute in order
    constexpr int CPRO = D / 8;        // 16-byte chunks per output row
    constexpr int RPS = 64 / CPRO;     // rows per store instruction
    const int rr = lane / CPRO, ch = lane % CPRO;
#pragma unroll
    for (int ps = 0; ps < 32 / RPS; ++ps) {
      const int row = RPS * ps + rr;
      const u32x4 w = *(const u32x4*)(region + row * (2 * D) + ((ch ^ (row & (CPRO - 1))) << 4));
      const int64_t ob = rowtab[row];
      if (ob >= 0) *(u32x4*)((uint16_t*)p.o + ob + 8 * ch) = w;
    }
    if (p.lse && lh == 0 && row_valid)
      p.lse[(int64_t)(qo_start + qo_idx) * p.num_qo_heads + qo_head] = lse_val;
  }
}

}  // namespace fi

// fp8 (e4m3 / e5m2) groupwise-scaled GEMM and grouped GEMM for gfx950.
//
// What it replaces: the reference's CUTLASS blockwise-scaled kernels
// (include/flashinfer/gemm/gemm_groupwise_sm100.cuh, group_gemm_fp8_groupwise_sm100.cuh:35-72, 76-250;
// bindings csrc/gemm_groupwise_sm100.cu:89-120, csrc/group_gemm_fp8_groupwise_sm100.cu:89-124).
//   D[g] = (A[g] . sA) (B[g] . sB)^T ;  A (cum_m, k) fp8 row-major, B (G, n, k) fp8 ("nt"), f32 accumulate,
//   scales per (1 or 128 rows of A, 128 k) and per (128 rows of B, 128 k)   (flashinfer/gemm.py:2657-2719).
//
// Structure: one workgroup (4 waves) owns a 128 (m) x 128 (n) output tile of one group and walks K in
// 128-wide blocks = the scale granularity.  A and B tiles are staged through LDS ([128 rows][128 B] fp8
// images, 16-byte chunks XOR-swizzled so that ds_read_b128 is conflict-free) with register prefetch of the
// next block.  The product is computed TRANSPOSED (D^T = B A^T, mfma_f32_32x32x16_fp8_fp8): the m index
// then sits on the lane, so the per-row A scale is one value per lane and the per-block B scale is wave
// uniform; each K block's partial product is folded into the running f32 accumulator with one fma per
// element (two-level accumulation, exactly the block-scaled definition).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "gemm_common.h"

namespace fi {

template <bool A_E5M2, bool B_E5M2>
__device__ __forceinline__ f32x16g mfma_fp8(long a, long b, f32x16g c) {
  if constexpr (!A_E5M2 && !B_E5M2) return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
  else if constexpr (!A_E5M2 && B_E5M2) return __builtin_amdgcn_mfma_f32_32x32x16_fp8_bf8(a, b, c, 0, 0, 0);
  else if constexpr (A_E5M2 && !B_E5M2) return __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(a, b, c, 0, 0, 0);
}

// MA_E5M2 / MB_E5M2 refer to the MFMA A operand (= matrix B of the GEMM) and MFMA B operand (= matrix A)

// MX: use the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 with unit (E8M0 = 127) scales -- the plain fp8
// product at twice the rate of the non-scaled fp8 MFMA (MI355X_MICROARCH.md, matrix cores).
template <bool MA_E5M2, bool MB_E5M2, bool MX>
__global__ void __launch_bounds__(kGemmThreads, 2) group_gemm_fp8_kernel(const GemmParams p) {
  // the 256 x 256 kernel's hardware-scale variant already did this call (launch_gemm: mid-size problems)
  if (fi_scales_are_pow2(p.pow2_flag)) return;
  __shared__ __attribute__((aligned(16))) uint8_t smem[2][2][kBM * kBK];  // [stage][A|B]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int lq = lane & 31, lh = lane >> 5;

  // ---- tile assignment: XCD-contiguous logical id ----
  const int total = p.num_m_tiles_bound * p.n_tiles;
  int logical;
  {
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int qn = total >> 3, rn = total & 7;
    logical = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
  }
  // Banded order inside the XCD's range: kBandM m tiles x all n tiles per band, m fastest.  The ~64
  // workgroups an XCD runs at a time then cover an 8 x 8 block of output tiles (16 operand tiles in L2 for
  // 64 products) instead of one m row (65 operand tiles): at C4 the B matrix of a group is streamed 4x
  // instead of 32x.
  constexpr int kBandM = 8;
  const int band_tiles = kBandM * p.n_tiles;
  const int band = logical / band_tiles;
  const int in_band = logical - band * band_tiles;
  const int band_m = min(kBandM, p.num_m_tiles_bound - band * kBandM);
  const int nt = in_band / band_m;
  const int mt_global = band * kBandM + (in_band - nt * band_m);
  int g = 0, m_begin = 0, m_end = p.m_total, mt = mt_global;
  if (p.m_indptr) {
    if (!find_group_tile<kBM>(p.m_indptr, p.num_groups, mt_global, lane, g, m_begin, m_end, mt)) return;
  } else if (mt_global * kBM >= p.m_total) {
    return;
  }
  const int m0 = m_begin + mt * kBM;
  const int n0 = nt * kBN;
  const int K = p.k, N = p.n;
  const int kblocks = K / kBK;
  const uint8_t* Bg = p.b + (int64_t)g * N * K;

  // ---- staging geometry: 4 passes of 32 rows x 8 chunks ----
  const int st_row = tid >> 3, st_ch = tid & 7;
  auto lds_off = [](int row, int ch) { return row * kBK + ((ch ^ ((row >> 1) & 7)) << 4); };
  // global loads run one k block ahead of the MFMAs, through registers; the k block past the end re-loads
  // the last one (no branch around a load)
  struct StageRegs {
    u32x4 a[4], b[4];
  };
  StageRegs rs;
  // uniform tile bases + 32-bit per-thread row offsets (128 rows x K bytes < 2^32, checked on the host)
  const uint8_t* const a_tile = p.a + (int64_t)m0 * K;
  const uint8_t* const b_tile = Bg + (int64_t)n0 * K;
  uint32_t a_off[4], b_off[4];
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int row = ps * 32 + st_row;
    a_off[ps] = (uint32_t)(min(m0 + row, m_end - 1) - m0) * (uint32_t)K + st_ch * 16;
    b_off[ps] = (uint32_t)(min(n0 + row, N - 1) - n0) * (uint32_t)K + st_ch * 16;
  }
  auto issue = [&](int kb, StageRegs& r) {
    const uint32_t koff = (uint32_t)(min(kb, kblocks - 1) * kBK);
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      r.a[ps] = *(const u32x4*)(a_tile + (a_off[ps] + koff));
      r.b[ps] = *(const u32x4*)(b_tile + (b_off[ps] + koff));
    }
  };
  auto commit = [&](int buf, const StageRegs& r) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int row = ps * 32 + st_row;
      *(u32x4*)(&smem[buf][0][lds_off(row, st_ch)]) = r.a[ps];
      *(u32x4*)(&smem[buf][1][lds_off(row, st_ch)]) = r.b[ps];
    }
  };
  // scales: a per lane (its m column), b per workgroup n tile
  auto a_scale_at = [&](int kb, int m) -> float {
    const int mi = p.a_gran_m == 1 ? m : m / p.a_gran_m;
    const int m_cnt = p.a_gran_m == 1 ? p.m_total : (p.m_total + p.a_gran_m - 1) / p.a_gran_m;
    return p.scale_k_major ? p.a_scale[(int64_t)mi * kblocks + kb] : p.a_scale[(int64_t)kb * m_cnt + mi];
  };
  const int n_sblocks = (N + 127) / 128;
  auto b_scale_at = [&](int kb) -> float {
    const int nb = n0 / 128;
    return p.scale_k_major ? p.b_scale[((int64_t)g * n_sblocks + nb) * kblocks + kb]
                           : p.b_scale[((int64_t)g * kblocks + kb) * n_sblocks + nb];
  };

  f32x16g acc[2][2];  // [n block][m block]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int m_lane[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) m_lane[mb] = min(m0 + 64 * wm + 32 * mb + lq, m_end - 1);

  float sa_nxt[2], sb_nxt;
  auto load_scales = [&](int kb) {
    const int kc = min(kb, kblocks - 1);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) sa_nxt[mb] = a_scale_at(kc, m_lane[mb]);
    sb_nxt = b_scale_at(kc);
  };
  load_scales(0);
  issue(0, rs);
  commit(0, rs);
  __syncthreads();
  auto k_step = [&](auto par_c, const int kb) {
    constexpr int buf = decltype(par_c)::value;  // == kb & 1
    // scales of block kb were requested one step ago, the next ones go out BEFORE this step's operand
    // loads (vmcnt retires in order: a scale load behind them would make the fold wait for the prefetch)
    const float sa[2] = {sa_nxt[0], sa_nxt[1]};
    const float sb = sb_nxt;
    load_scales(kb + 1);
    issue(kb + 1, rs);

    if constexpr (MX) {
      // A fragments of the whole k block stay in registers; the two 32-column halves of the wave's tile
      // are then processed one after the other, each with its own 32-register partial product, so that
      // folding half 0 into the accumulator (vector pipe) runs beside the MFMAs of half 1 (matrix pipe)
      i32x8g fa[2][2];  // [kk][mb]: this lane holds bytes [64 kk + 32 lh, +32) of its row
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const u32x4 lo = *(const u32x4*)(&smem[buf][0][lds_off(64 * wm + 32 * mb + lq, 4 * kk + 2 * lh)]);
          const u32x4 hi = *(const u32x4*)(&smem[buf][0][lds_off(64 * wm + 32 * mb + lq, 4 * kk + 2 * lh + 1)]);
          fa[kk][mb] = i32x8g{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        f32x16g part[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int r = 0; r < 16; ++r) part[mb][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const u32x4 lo = *(const u32x4*)(&smem[buf][1][lds_off(64 * wn + 32 * nb + lq, 4 * kk + 2 * lh)]);
          const u32x4 hi = *(const u32x4*)(&smem[buf][1][lds_off(64 * wn + 32 * nb + lq, 4 * kk + 2 * lh + 1)]);
          const i32x8g fb = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            part[mb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                fb, fa[kk][mb], part[mb], MA_E5M2 ? 1 : 0, MB_E5M2 ? 1 : 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const float s = sa[mb] * sb;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[nb][mb][r] += s * part[mb][r];
        }
      }
    } else {
    f32x16g part[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {  // 32 bytes of k per step pair: bytes [32 kp + 16 lh, +16)
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
        fa[mb] = *(const u32x4*)(&smem[buf][0][lds_off(64 * wm + 32 * mb + lq, 2 * kp + lh)]);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        fb[nb] = *(const u32x4*)(&smem[buf][1][lds_off(64 * wn + 32 * nb + lq, 2 * kp + lh)]);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const long bfrag = (long)fb[nb][2 * half] | ((long)fb[nb][2 * half + 1] << 32);
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            const long afrag = (long)fa[mb][2 * half] | ((long)fa[mb][2 * half + 1] << 32);
            part[nb][mb] = mfma_fp8<MA_E5M2, MB_E5M2>(bfrag, afrag, part[nb][mb]);
          }
        }
      }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float s = sa[mb] * sb;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][mb][r] += s * part[nb][mb][r];
      }
    }
    commit(buf ^ 1, rs);
    __syncthreads();
  };
  {
    int kb = 0;
    for (; kb + 1 < kblocks; kb += 2) {
      k_step(std::integral_constant<int, 0>{}, kb);
      k_step(std::integral_constant<int, 1>{}, kb + 1);
    }
    if (kb < kblocks) k_step(std::integral_constant<int, 0>{}, kb);
  }

  // ---- epilogue ----
  // The accumulators are transposed (m on the lane): a direct store writes 8-byte pieces.  Each wave turns
  // its 64 x 64 block through its own 9 KB of LDS ([m][64 n] 16-bit rows, 144-byte stride; every LDS stage
  // read finished at the loop's last barrier) and stores whole 128-byte rows, 16 bytes per lane.
  {
    constexpr int kOutStride = 144;
    uint8_t* const scratch = &smem[0][0][0] + wave * (64 * kOutStride);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          uint32_t w[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const uint32_t lo = f32_to_16bit(acc[nb][mb][4 * r4 + 2 * e], p.out_dtype);
            const uint32_t hi = f32_to_16bit(acc[nb][mb][4 * r4 + 2 * e + 1], p.out_dtype);
            w[e] = lo | (hi << 16);
          }
          *(u32x2*)(scratch + (32 * mb + lq) * kOutStride + (32 * nb + 8 * r4 + 4 * lh) * 2) = u32x2{w[0], w[1]};
        }
    const bool d_aligned16 = (((uintptr_t)p.d) & 15) == 0;
#pragma unroll
    for (int i2 = 0; i2 < 8; ++i2) {
      const int idx = lane + 64 * i2;
      const int r = idx >> 3, c = idx & 7;
      const u32x4 v = *(const u32x4*)(scratch + r * kOutStride + c * 16);
      const int m = m0 + 64 * wm + r;
      const int n = n0 + 64 * wn + 8 * c;
      if (m >= m_end || n >= N) continue;  // n is a multiple of 8 and so is N
      uint16_t* dst = (uint16_t*)p.d + (int64_t)m * N + n;
      if (d_aligned16) {
        *(u32x4*)dst = v;
      } else {
        *(u32x2*)dst = u32x2{v[0], v[1]};
        *(u32x2*)(dst + 4) = u32x2{v[2], v[3]};
      }
    }
  }
}

// ---- 256 x 128 tile persistent kernel for large problems (LDS-DMA staging) --------------------------------------
// One 512-thread workgroup per CU owns a 256 (m) x 128 (n) tile (or 128 x 256, TM): 8 waves of 64 x 64, two per SIMD;
// the tile needs 25 % fewer operand bytes per flop from L2 than 128 x 128 (170 flop/B).  (r1's register-staged form
// of this kernel was removed in r3: the DMA form below replaced it as the default in r1 and nothing selected it.)
constexpr int kWsThreads = 512;
constexpr int kWsBM = 256;

// ---- the same 256 x 128 persistent kernel with LDS-DMA staging (default for large problems) --------------
// Operands go global -> LDS directly (`global_load_lds_dwordx4`, 1 KiB per wave instruction), two k blocks
// ahead through a ring of three 52 KB LDS stages (156 of the 160 KB): no staging registers, no
// VGPR -> LDS store transfer.  Synchronisation is by hand: every wave waits `vmcnt(8)` (its own pieces of the
// NEXT block landed, the 8 DMA instructions of the block after it still in flight) and then a raw `s_barrier`
// -- `__syncthreads()` would drain vmcnt.
template <bool MA_E5M2, bool MB_E5M2, int TM>
__global__ void __launch_bounds__(kWsThreads, 1) group_gemm_fp8_dma_kernel(const GemmParams p) {
  if (fi_scales_are_pow2(p.pow2_flag)) return;  // see group_gemm_fp8_kernel
  // TM x TN output tile, TM + TN = 384 rows of operands per k block: 256 x 128, or 128 x 256 for groups of few
  // rows (a group of <= 128 rows wastes half of a 256-row tile, and B -- streamed once from HBM when every group
  // has a single m tile -- gets twice the bytes in flight)
  constexpr int TN = kWsBM + kBN - TM;
  constexpr int WM = TM / 64;  // waves along m; 8 / WM along n
  static_assert(TM == 256 || TM == 128, "tile");
  // [stage][A 32 KB | B 16 KB | per wave: 64 A scales, 64 x the B scale]  (one array: the compiler tells a DMA
  // target from an LDS read by constant offsets inside ONE object; a second array made it wait vmcnt(0))
  constexpr int kScOff = (kWsBM + kBN) * kBK;
  __shared__ __attribute__((aligned(1024))) uint8_t smem[3][kScOff + (kWsThreads / 64) * 512];
  constexpr int kBOff = TM * kBK;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int lq = lane & 31, lh = lane >> 5;
  const int K = p.k, N = p.n;
  const int kblocks = K / kBK;
  auto lds_off = [](int row, int ch) { return row * kBK + ((ch ^ ((row >> 1) & 7)) << 4); };
  const int m_cnt = p.a_gran_m == 1 ? p.m_total : (p.m_total + p.a_gran_m - 1) / p.a_gran_m;
  const int a_sc_stride = p.scale_k_major ? 1 : m_cnt;
  const int n_sblocks = (N + 127) / 128;
  const int b_sc_stride = p.scale_k_major ? 1 : n_sblocks;
  // LDS read addresses: every fragment address is one per-lane base XOR a literal (the k half flips chunk
  // bit 2, the second 16 bytes chunk bit 0) plus a literal; the bases pass through an empty asm in the loop
  // so that the derived addresses are recomputed there instead of living in 16 registers
  uint32_t a_rd_base = (uint32_t)lds_off(64 * wm + lq, 2 * lh);
  uint32_t b_rd_base = (uint32_t)(kBOff + lds_off(64 * wn + lq, 2 * lh));

  // ---- persistent workgroup: the XCD-contiguous tile range of this XCD, strided by its workgroups ----
  const int n_tiles = (N + TN - 1) / TN;
  const int total = p.num_m_tiles_bound * n_tiles;
  int logical, logical_end;
  const int logical_step = gridDim.x >> 3;
  {
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int qn = total >> 3, rn = total & 7;
    const int start = xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
    logical = start + slot;
    logical_end = start + qn + (xcd < rn ? 1 : 0);
  }

  // Up to 64 groups: lane i keeps group i's row range and first tile for the whole kernel, so the per-tile
  // (group, m tile) search is a ballot + three readlanes instead of loads (a round trip per tile, with
  // nothing else in flight to hide it).  More groups: the looping search per tile.
  const bool groups_cached = p.m_indptr != nullptr && p.num_groups <= 64;
  int gc_lo = 0, gc_hi = 0, gc_start = 0;
  if (groups_cached) {
    const bool in = lane < p.num_groups;
    gc_lo = in ? p.m_indptr[lane] : 0;
    gc_hi = in ? p.m_indptr[lane + 1] : 0;
    const int tiles = (gc_hi - gc_lo + TM - 1) / TM;
    int incl = tiles;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    gc_start = incl - tiles;
  }

  f32x16g acc[2][2];  // [n block][m block]
  int out_m0 = 0, out_n0 = 0, out_m_end = 0;  // tile whose accumulators are waiting to be stored
  bool have_out = false;
  // Output: the accumulators are transposed (m on the lane), so a direct store writes 8-byte pieces.  Each
  // wave instead turns its 64 x 64 block, one 32-column half at a time, through 5 KB of LDS stage 2 -- free at
  // a tile border while the next tile's first two blocks are already landing in stages 0 and 1 -- and stores
  // 64-byte row pieces, 16 bytes per lane.
  constexpr int kOutStride = 80;
  const bool d_aligned16 = (((uintptr_t)p.d) & 15) == 0;
  auto store_tile = [&]() {
    uint8_t* const scratch = &smem[2][0] + wave * (64 * kOutStride);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          uint32_t w[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const uint32_t lo = f32_to_16bit(acc[nb][mb][4 * r4 + 2 * e], p.out_dtype);
            const uint32_t hi = f32_to_16bit(acc[nb][mb][4 * r4 + 2 * e + 1], p.out_dtype);
            w[e] = lo | (hi << 16);
          }
          *(u32x2*)(scratch + (32 * mb + lq) * kOutStride + (8 * r4 + 4 * lh) * 2) = u32x2{w[0], w[1]};
        }
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2) {
        const int idx = lane + 64 * i2;
        const int r = idx >> 2, c = idx & 3;
        const u32x4 v = *(const u32x4*)(scratch + r * kOutStride + c * 16);
        const int m = out_m0 + 64 * wm + r;
        const int n = out_n0 + 64 * wn + 32 * nb + 8 * c;
        if (m >= out_m_end || n >= N) continue;  // n is a multiple of 8 and so is N
        uint16_t* dst = (uint16_t*)p.d + (int64_t)m * N + n;
        if (d_aligned16) {
          *(u32x4*)dst = v;
        } else {
          *(u32x2*)dst = u32x2{v[0], v[1]};
          *(u32x2*)(dst + 4) = u32x2{v[2], v[3]};
        }
      }
    }
  };

  for (; logical < logical_end; logical += logical_step) {
    constexpr int kBandM = 1024 / TM;  // 1024 rows x all n per band, as in the 128 x 128 kernel
    const int band_tiles = kBandM * n_tiles;
    const int band = logical / band_tiles;
    const int in_band = logical - band * band_tiles;
    const int band_m = min(kBandM, p.num_m_tiles_bound - band * kBandM);
    const int nt = in_band / band_m;
    const int mt_global = band * kBandM + (in_band - nt * band_m);
    int g = 0, m_begin = 0, m_end = p.m_total, mt = mt_global;
    bool found = true;
    if (p.m_indptr) {
      if (groups_cached) {
        const int tiles = (gc_hi - gc_lo + TM - 1) / TM;
        const uint64_t hit = __ballot(mt_global >= gc_start && mt_global < gc_start + tiles);
        found = hit != 0;
        if (found) {
          g = __builtin_ctzll(hit);
          m_begin = __builtin_amdgcn_readlane(gc_lo, g);
          m_end = __builtin_amdgcn_readlane(gc_hi, g);
          mt = mt_global - __builtin_amdgcn_readlane(gc_start, g);
        }
      } else {
        found = find_group_tile<TM>(p.m_indptr, p.num_groups, mt_global, lane, g, m_begin, m_end, mt);
      }
    } else if (mt_global * TM >= p.m_total) {
      found = false;
    }
    if (!found) continue;  // the grid bound counts one partial tile per group; uniform per workgroup
    const int m0 = m_begin + mt * TM;
    const int n0 = nt * TN;
    const uint8_t* Bg = p.b + (int64_t)g * N * K;

    // LDS-DMA geometry: the stage image is [384 rows][128 B] (A rows 0-255, B rows 256-383) and a piece =
    // one wave instruction = 8 rows = 1 KiB of it, written linearly (lane i -> byte 16 i of the piece).  The
    // XOR swizzle is applied on the GLOBAL side: lane i fetches chunk (i & 7) ^ key(row) of row (i >> 3).
    // Wave w owns pieces 6 w .. 6 w + 5 of the 48.
    const uint8_t* const a_tile = p.a + (int64_t)m0 * K;
    const uint8_t* const b_tile = Bg + (int64_t)n0 * K;
    uint32_t d_off[6];
#pragma unroll
    for (int j2 = 0; j2 < 6; ++j2) {
      const int row = 8 * (6 * wave + j2) + (lane >> 3);
      const int ch = (lane & 7) ^ ((row >> 1) & 7);
      const int r = row < TM ? min(m0 + row, m_end - 1) - m0 : min(n0 + row - TM, N - 1) - n0;
      d_off[j2] = (uint32_t)r * (uint32_t)K + ch * 16;
    }
    // scale sources for the DMA: one A-scale pointer per lane (row 64 wm + lane of the tile), B scale uniform
    const int nsb = min((n0 + 64 * wn) / 128, n_sblocks - 1);  // this wave's 128-wide scale block of B
    const float* const b_sc = p.scale_k_major ? p.b_scale + ((int64_t)g * n_sblocks + nsb) * kblocks
                                              : p.b_scale + (int64_t)g * kblocks * n_sblocks + nsb;
    const float* a_sc_dma;
    {
      const int m = min(m0 + 64 * wm + lane, m_end - 1);
      const int mi = p.a_gran_m == 1 ? m : m / p.a_gran_m;
      a_sc_dma = p.scale_k_major ? p.a_scale + (int64_t)mi * kblocks : p.a_scale + mi;
    }
    auto dma = [&](int kb, int stage) {
      const uint32_t koff = (uint32_t)(kb * kBK);
#pragma unroll
      for (int j2 = 0; j2 < 6; ++j2) {
        const int q = 6 * wave + j2;
        const uint8_t* src_p = (q < TM / 8 ? a_tile : b_tile) + (d_off[j2] + koff);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_p,
                                         (__attribute__((address_space(3))) void*)(&smem[stage][q * 1024]), 16, 0, 0);
      }
      // the block's scales travel the same way (4 bytes per lane): lane L's A scale is the one of row
      // 64 wm + L of the tile, the B scale is fetched by every lane.  They are then ordinary LDS reads; an
      // ordinary global load in flight beside the DMA would cost a vmcnt(0) at its first use.
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_sc_dma + (int64_t)kb * a_sc_stride),
                                       (__attribute__((address_space(3))) void*)(&smem[stage][kScOff + wave * 512]), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_sc + (int64_t)kb * b_sc_stride),
                                       (__attribute__((address_space(3))) void*)(&smem[stage][kScOff + wave * 512 + 256]), 4, 0, 0);
    };
    // the next tile's first two blocks go out BEFORE the finished tile is converted and stored (stages 0
    // and 1; the store's scratch is stage 2): with one workgroup per CU nothing else would cover the round trip
    dma(0, 0);
    if (kblocks > 1) dma(1, 1);
    const bool stored = have_out;
    if (have_out) store_tile();
    out_m0 = m0;
    out_n0 = n0;
    out_m_end = m_end;
    have_out = true;
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i2][j2][r] = 0.f;
    // D(0) landed.  The output stores are younger than both blocks and vmcnt retires in order, so with
    // stores in the queue the only count that is sure to cover D(0) is 0.
    // wait + barrier as ONE asm statement: the s_barrier builtin is no memory fence for the compiler, which may
    // hoist the next LDS reads between a separate wait and the barrier (seen in prefill_fp8_kernel.h)
    // (also: every wave is done with the stage 2 scratch before step 0's DMA)
    if (kblocks > 1 && !stored) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    f32x16g p_carry;  // block (1, 1) of the previous k step, folded at the start of the next one
#pragma unroll
    for (int r = 0; r < 16; ++r) p_carry[r] = 0.f;
    float s_carry = 0.f;

    auto k_step = [&](auto par_c, const int kb) {
      constexpr int buf = decltype(par_c)::value;  // == kb % 3
      if (kb + 2 < kblocks) dma(kb + 2, (buf + 2) % 3);  // D(kb + 2) into the stage read in step kb - 1
      const float* const sc = (const float*)(&smem[buf][kScOff + wave * 512]);
      const float sa[2] = {sc[lq], sc[32 + lq]};
      const float sb = sc[64 + lane];
      asm volatile("" : "+v"(a_rd_base), "+v"(b_rd_base));
      const uint8_t* const stage = &smem[buf][0];
      // One 32 x 32 block at a time, cut into sched_barrier regions: left alone the scheduler hoists every
      // LDS read and all four partial products to the top and spills ~30 registers.
      auto frag = [&](uint32_t base, int kk, int blk) {
        const u32x4 lo = *(const u32x4*)(stage + ((base ^ (kk << 6)) + blk * 32 * kBK));
        const u32x4 hi = *(const u32x4*)(stage + ((base ^ (kk << 6) ^ 16) + blk * 32 * kBK));
        return i32x8g{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      };
      auto mfma0 = [&](const i32x8g& b, const i32x8g& a) {
        f32x16g z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, z, MA_E5M2 ? 1 : 0, MB_E5M2 ? 1 : 0, 0,
                                                               0x7F7F7F7F, 0, 0x7F7F7F7F);
      };
      auto mfma1 = [&](const i32x8g& b, const i32x8g& a, const f32x16g& c) {
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c, MA_E5M2 ? 1 : 0, MB_E5M2 ? 1 : 0, 0,
                                                               0x7F7F7F7F, 0, 0x7F7F7F7F);
      };
      auto fold = [&](int nb, int mb, const f32x16g& part, float s) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][mb][r] += s * part[r];
        asm volatile("" : "+v"(acc[nb][mb]));  // IR-level sinking ignores sched_barrier: pin the fold here
      };
      const float s0 = sa[0] * sb, s1 = sa[1] * sb;
      i32x8g fa0[2], fa1[2], fb0[2], fb1[2];
      // The two MFMAs of a block are dependent (same accumulator): the second cannot issue for the 64 cycles
      // the first one runs, and neither can anything behind it in this wave.  So the previous block's fold
      // sits BETWEEN the two, in the shadow of the first.
      // region 0: fragment reads, the fold carried over from the previous k step hides their latency
      fa0[0] = frag(a_rd_base, 0, 0);
      fb0[0] = frag(b_rd_base, 0, 0);
      fa0[1] = frag(a_rd_base, 1, 0);
      fb0[1] = frag(b_rd_base, 1, 0);
      fa1[0] = frag(a_rd_base, 0, 1);
      fa1[1] = frag(a_rd_base, 1, 1);
      fold(1, 1, p_carry, s_carry);
      __builtin_amdgcn_sched_barrier(0);
      f32x16g p00 = mfma0(fb0[0], fa0[0]);
      __builtin_amdgcn_sched_barrier(0);
      fb1[0] = frag(b_rd_base, 0, 1);
      fb1[1] = frag(b_rd_base, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      p00 = mfma1(fb0[1], fa0[1], p00);
      // region 1
      f32x16g p01 = mfma0(fb0[0], fa1[0]);
      __builtin_amdgcn_sched_barrier(0);
      fold(0, 0, p00, s0);
      __builtin_amdgcn_sched_barrier(0);
      p01 = mfma1(fb0[1], fa1[1], p01);
      // region 2
      f32x16g p10 = mfma0(fb1[0], fa0[0]);
      __builtin_amdgcn_sched_barrier(0);
      fold(0, 1, p01, s1);
      __builtin_amdgcn_sched_barrier(0);
      p10 = mfma1(fb1[1], fa0[1], p10);
      // region 3
      f32x16g p11 = mfma0(fb1[0], fa1[0]);
      __builtin_amdgcn_sched_barrier(0);
      fold(1, 0, p10, s0);
      __builtin_amdgcn_sched_barrier(0);
      p11 = mfma1(fb1[1], fa1[1], p11);
      p_carry = p11;
      s_carry = s1;
      // D(kb + 1) must have landed (all waves' pieces: barrier); the 8 pieces of D(kb + 2) stay in flight
      if (kb + 2 < kblocks) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);  // no mixing of two k steps either
    };
    {
      int kb = 0;
      for (; kb + 2 < kblocks; kb += 3) {
        k_step(std::integral_constant<int, 0>{}, kb);
        k_step(std::integral_constant<int, 1>{}, kb + 1);
        k_step(std::integral_constant<int, 2>{}, kb + 2);
      }
      if (kb < kblocks) k_step(std::integral_constant<int, 0>{}, kb);
      if (kb + 1 < kblocks) k_step(std::integral_constant<int, 1>{}, kb + 1);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[1][1][r] += s_carry * p_carry[r];
  }
  if (have_out) store_tile();
}

static hipError_t launch_gemm(const GemmParams& p_in, hipStream_t stream) {
  GemmParams p = p_in;
  const int grid = p.num_m_tiles_bound * p.n_tiles;
  if (grid <= 0) return hipSuccess;
  // MFMA A operand = GEMM matrix B, MFMA B operand = GEMM matrix A
  const int sel = (p.b_is_e5m2 ? 2 : 0) | (p.a_is_e5m2 ? 1 : 0);
  static const bool use_mx = [] {
    const char* e = getenv("FI_GEMM_MX");
    return e ? atoi(e) != 0 : true;
  }();
  // producer/consumer kernel: MX path, enough 256 x 128 tiles to give every CU a few
  static const int ws_min_tiles = [] {
    const char* e = getenv("FI_GEMM_WS_MIN_TILES");
    return e ? atoi(e) : 2 * fi_num_compute_units();
  }();
  const int ws_tiles = p.num_m_tiles_bound_ws * p.n_tiles;
  const int ws_grid = (fi_num_compute_units() / 8) * 8;  // persistent: one workgroup per CU, XCD-aligned
  const bool use_ws = use_mx && ws_min_tiles >= 0 && ws_tiles >= ws_min_tiles && ws_grid >= 8;
  constexpr bool use_dma = true;
  // 128 x 256 tiles for grouped problems whose groups have few rows (a <= 128-row group fills half of a
  // 256-row tile); FI_GEMM_DMA_TM = 128 / 256 forces the shape
  static const int forced_tm = [] {
    const char* e = getenv("FI_GEMM_DMA_TM");
    return e ? atoi(e) : 0;
  }();
  const bool few_rows = p.m_indptr != nullptr && p.m_total <= 160 * (int64_t)p.num_groups;
  const bool tall = forced_tm == 128 || (forced_tm != 256 && few_rows);
  const int tall_tiles = p.num_m_tiles_bound * ceil_div(p.n, 2 * kBN);
  // 256 x 256 tiles (gemm_big.hip); FI_GEMM_BIG=0 keeps 256 x 128 / 128 x 128.  r3 (tools/bench_gemm_threshold.py):
  //  * power-of-two scales (the reference quantiser's; decided on the device): the hardware-scale variant beats the
  //    256 x 128 LDS-DMA kernel on every shape that reaches either (8 x 1024 x 4096 x 7168: 2.32 against 1.37 PFLOP/s;
  //    256 tiles = one per CU: 1.27 against 1.10) and the 128 x 128 kernel from 128 tiles on (2048 x 4096 x 4096: 1.12
  //    against 0.83, 3072 x 4096 x 4096: 1.65 against 1.13; 64 tiles: the 128 x 128 kernel's 256 workgroups win) -> from
  //    half a tile per CU on (for scales that do NOT qualify the check kernel and the variant that returns at once
  //    cost 5-7 us, ~7 % of the smallest such calls);
  //  * arbitrary scales (the fold variant): mixed below four tiles per CU (8 x 1024 x 4096 x 7168 1.57 against 1.32,
  //    4 x 1024 x 7168 x 2048 0.88 against 1.18) -> from 4 x CUs tiles on, as in r2.
  // In between only the hardware-scale variant is launched (it returns at once when the scales do not qualify) and
  // the kernel chosen below gets the flag word and returns at once when they do.
  static const int big_hws_min_tiles = [] {
    const char* e = getenv("FI_GEMM_BIG");
    if (e && atoi(e) == 0) return -1;
    const char* t = getenv("FI_GEMM_BIG_MIN_TILES");
    return t ? atoi(t) : fi_num_compute_units() / 2;
  }();
  static const int big_fold_min_tiles = [] {
    const char* t = getenv("FI_GEMM_BIG_FOLD_MIN_TILES");
    if (t) return atoi(t);
    const char* h = getenv("FI_GEMM_BIG_MIN_TILES");  // a forced threshold (tests, A/B runs) applies to both variants
    return h ? atoi(h) : 4 * fi_num_compute_units();
  }();
  const int big_tiles = p.num_m_tiles_bound_ws * ceil_div(p.n, 2 * kBN);
  // groups of few rows: from ~96 rows per group on the 256 x 256 kernel with its half-empty row tiles is ahead of the
  // 128 x 256 kernel (256 groups x 128 x 4096 x 7168: 1.03 against 0.81 PFLOP/s with power-of-two scales, 0.92 against
  // 0.82 folded; 160 rows: 1.23 against 0.76; 64 rows: equal; 96 rows 0.69 against 0.54 / 0.53 against 0.55)
  const bool rows_ok = forced_tm == 256 || (forced_tm == 0 && (!few_rows || p.m_total > 96 * (int64_t)p.num_groups));
  const bool big_ok = use_mx && ws_min_tiles >= 0 && ws_grid >= 8 && use_dma && big_hws_min_tiles >= 0 && rows_ok;
  if (big_ok && big_tiles >= big_hws_min_tiles) {
    GemmParams q = p;
    q.num_m_tiles_bound = p.num_m_tiles_bound_ws;
    if (big_tiles >= big_fold_min_tiles) return launch_gemm_big(q, ws_grid, stream, nullptr);
    uint32_t* flag = nullptr;
    hipError_t e = launch_gemm_big(q, ws_grid, stream, &flag);
    if (e != hipSuccess) return e;
    p.pow2_flag = flag;  // null (no flag ring yet under a stream capture, FI_GEMM_HW_SCALES=0): nothing was launched
  }
  if (use_dma && use_mx && ws_min_tiles >= 0 && ws_grid >= 8 && tall && tall_tiles >= ws_min_tiles) {
    switch (sel) {  // num_m_tiles_bound already counts 128-row tiles
      case 0: group_gemm_fp8_dma_kernel<false, false, 128><<<dim3(ws_grid), dim3(kWsThreads), 0, stream>>>(p); break;
      case 1: group_gemm_fp8_dma_kernel<false, true, 128><<<dim3(ws_grid), dim3(kWsThreads), 0, stream>>>(p); break;
      case 2: group_gemm_fp8_dma_kernel<true, false, 128><<<dim3(ws_grid), dim3(kWsThreads), 0, stream>>>(p); break;
      default: group_gemm_fp8_dma_kernel<true, true, 128><<<dim3(ws_grid), dim3(kWsThreads), 0, stream>>>(p); break;
    }
    return hipGetLastError();
  }
  if (use_ws && use_dma) {
    GemmParams q = p;
    q.num_m_tiles_bound = p.num_m_tiles_bound_ws;
    switch (sel) {
      case 0: group_gemm_fp8_dma_kernel<false, false, 256><<<dim3(ws_grid), dim3(kWsThreads), 0, stream>>>(q); break;
      case 1: group_gemm_fp8_dma_kernel<false, true, 256><<<dim3(ws_grid), dim3(kWsThreads), 0, stream>>>(q); break;
      case 2: group_gemm_fp8_dma_kernel<true, false, 256><<<dim3(ws_grid), dim3(kWsThreads), 0, stream>>>(q); break;
      default: group_gemm_fp8_dma_kernel<true, true, 256><<<dim3(ws_grid), dim3(kWsThreads), 0, stream>>>(q); break;
    }
    return hipGetLastError();
  }
#define FI_GEMM_LAUNCH(A, B)                                                                   \
  if (use_mx)                                                                                  \
    group_gemm_fp8_kernel<A, B, true><<<dim3(grid), dim3(kGemmThreads), 0, stream>>>(p);       \
  else                                                                                         \
    group_gemm_fp8_kernel<A, B, false><<<dim3(grid), dim3(kGemmThreads), 0, stream>>>(p);
  switch (sel) {
    case 0: FI_GEMM_LAUNCH(false, false) break;
    case 1: FI_GEMM_LAUNCH(false, true) break;
    case 2: FI_GEMM_LAUNCH(true, false) break;
    default: FI_GEMM_LAUNCH(true, true) break;
  }
#undef FI_GEMM_LAUNCH
  return hipGetLastError();
}

static int fill_and_check(GemmParams& p, const char* who, const void* a, const void* b, const void* sa,
                          const void* sb, void* d, int m_total, int n, int k, int gm, int gn, int gk,
                          int scale_k_major, int a_dt, int b_dt, int d_dt) {
  FI_REQUIRE(a && b && sa && sb && d, "%s: null tensor", who);
  FI_REQUIRE((a_dt == FI_DTYPE_FP8_E4M3 || a_dt == FI_DTYPE_FP8_E5M2) &&
                 (b_dt == FI_DTYPE_FP8_E4M3 || b_dt == FI_DTYPE_FP8_E5M2),
             "%s: a and b must be fp8 (e4m3 / e5m2)", who);
  FI_REQUIRE(d_dt == FI_DTYPE_F16 || d_dt == FI_DTYPE_BF16, "%s: output dtype must be f16/bf16", who);
  FI_REQUIRE((gm == 1 || gm == 128) && gn == 128 && gk == 128,
             "%s: scale granularity (%d,%d,%d) unsupported; (1,128,128) or (128,128,128)", who, gm, gn, gk);
  FI_REQUIRE(n % 8 == 0 && k % 16 == 0, "%s: n must be a multiple of 8 and k of 16", who);
  FI_REQUIRE(k % 128 == 0, "%s: k must be a multiple of the 128-wide scale block", who);
  FI_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && ((uintptr_t)d % 8) == 0,
             "%s: a/b must be 16-byte aligned", who);
  memset(&p, 0, sizeof(p));
  p.a = (const uint8_t*)a;
  p.b = (const uint8_t*)b;
  p.a_scale = (const float*)sa;
  p.b_scale = (const float*)sb;
  p.d = d;
  p.m_total = m_total;
  p.n = n;
  p.k = k;
  p.a_gran_m = gm;
  p.scale_k_major = scale_k_major;
  p.out_dtype = d_dt;
  p.n_tiles = ceil_div(n, kBN);
  p.a_is_e5m2 = a_dt == FI_DTYPE_FP8_E5M2;
  p.b_is_e5m2 = b_dt == FI_DTYPE_FP8_E5M2;
  return 0;
}

}  // namespace fi

using namespace fi;

extern "C" FI_API int fi_gemm_fp8_nt_groupwise(const void* a, const void* b, const void* a_scale,
                                               const void* b_scale, void* d, int32_t m, int32_t n,
                                               int32_t k, int32_t gran_m, int32_t gran_n,
                                               int32_t gran_k, int32_t scale_k_major, int32_t a_dtype,
                                               int32_t b_dtype, int32_t d_dtype, fi_stream_t stream) {
  if (m == 0 || n == 0) return 0;
  GemmParams p;
  if (fill_and_check(p, "gemm_fp8_nt_groupwise", a, b, a_scale, b_scale, d, m, n, k, gran_m, gran_n,
                     gran_k, scale_k_major, a_dtype, b_dtype, d_dtype))
    return 1;
  p.num_groups = 1;
  p.m_indptr = nullptr;
  p.pow2_flag = nullptr;
  p.num_m_tiles_bound = ceil_div(m, kBM);
  p.num_m_tiles_bound_ws = ceil_div(m, kWsBM);
  FI_HIP_CALL(launch_gemm(p, (hipStream_t)stream));
  return 0;
}

extern "C" FI_API int fi_group_gemm_fp8_nt_groupwise(const void* a, const void* b, const void* a_scale,
                                                     const void* b_scale, void* d,
                                                     const int32_t* m_indptr, int32_t num_groups,
                                                     int32_t cum_m, int32_t n, int32_t k,
                                                     int32_t gran_m, int32_t gran_n, int32_t gran_k,
                                                     int32_t scale_k_major, int32_t a_dtype,
                                                     int32_t b_dtype, int32_t d_dtype,
                                                     fi_stream_t stream) {
  if (cum_m == 0 || n == 0 || num_groups == 0) return 0;
  GemmParams p;
  if (fill_and_check(p, "group_gemm_fp8_nt_groupwise", a, b, a_scale, b_scale, d, cum_m, n, k, gran_m,
                     gran_n, gran_k, scale_k_major, a_dtype, b_dtype, d_dtype))
    return 1;
  FI_REQUIRE(m_indptr, "group_gemm_fp8_nt_groupwise: null m_indptr");
  p.num_groups = num_groups;
  p.m_indptr = m_indptr;
  p.pow2_flag = nullptr;
  // every group adds at most one partial tile on top of cum_m / 128
  p.num_m_tiles_bound = cum_m / kBM + num_groups;
  p.num_m_tiles_bound_ws = cum_m / kWsBM + num_groups;
  FI_HIP_CALL(launch_gemm(p, (hipStream_t)stream));
  return 0;
}

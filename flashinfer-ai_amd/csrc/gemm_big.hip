// fp8 groupwise (grouped) GEMM, 256 x 256 output tile per workgroup: the kernel for large problems (C4).
//
// What it replaces: the reference's 2-SM 256-row CUTLASS tile for the same op
// (include/flashinfer/gemm/group_gemm_fp8_groupwise_sm100.cuh:107-108); arithmetic as in gemm.hip (D^T = B A^T
// on `v_mfma_scale_f32_32x32x64_f8f6f4` with unit hardware scales, every 128-wide k block's partial product
// folded into the f32 accumulator with its two scales).
//
// Why a second shape: the 256 x 128 kernel of gemm.hip asks L2 for 48 KB per 8.4 MFLOP and spends 38 % of its
// wave time waiting for them (profiles/r01_gemm_pmc.txt: 25 GB of L2 requests per C4 launch).  A 256 x 256
// tile needs 64 KB per 16.8 MFLOP -- two thirds of the bytes per flop -- and its 16 MFMAs per wave and k block
// read 24 fragments from LDS instead of 32.
//   * one persistent 512-thread workgroup per CU, 8 waves as 4 (m) x 2 (n), wave tile 64 (m) x 128 (n):
//     128 accumulator registers, the A fragments of a k block held (32), the B fragments streamed (2 x 16),
//     two partial products in flight (32);
//   * operands global -> LDS by DMA (`global_load_lds_dwordx4`), stage = [A 256 rows | B 256 rows] x 128 B +
//     the block's scales, a ring of TWO stages (136 KB): block kb + 1 is issued at the top of step kb and
//     awaited (`vmcnt(0)` + barrier, one asm statement) at its end;
//   * output through LDS (stage 1, free at a tile border) as whole 128-byte row pieces.
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <type_traits>

#include "gemm_common.h"

namespace fi {

constexpr int kBigThreads = 512;
constexpr int kBigTM = 256, kBigTN = 256;
constexpr int kBigScOff = (kBigTM + kBigTN) * kBK;              // 64 KB of operands, then the scales
constexpr int kBigStage = kBigScOff + (kBigThreads / 64) * 512;  // per wave: 64 A scales, 64 x the B scale

#ifndef FI_GEMM_BIG_NT_STORE
#define FI_GEMM_BIG_NT_STORE 1  // output stores non-temporal: 128 KB per tile that nobody reads again stay out of the L2's way
#endif
#ifndef FI_GEMM_BIG_CARRY
#define FI_GEMM_BIG_CARRY 2  // 0 none, 1 the held-back pair behind the DMA issue, 2 in front of it
#endif
#ifndef FI_GEMM_BIG_BRANCHFREE
#define FI_GEMM_BIG_BRANCHFREE 3  // bit 0: hardware-scale path, bit 1: fold path (see k_step)
#endif
#ifndef FI_GEMM_BIG_BAND
#define FI_GEMM_BIG_BAND 1024
#endif
#ifndef FI_GEMM_BIG_INTERLEAVE
#define FI_GEMM_BIG_INTERLEAVE 1
#endif
#ifndef FI_GEMM_BIG_KO
#define FI_GEMM_BIG_KO 0  // experiments only, bit mask: 1 no output stores, 2 no fold, 4 no DMA in the k loop, 8 / 32 (timing only, wrong results): the step awaits none / all but the newest block of its DMA
#endif

// flag[block] = 1 when the block saw a scale that is not a positive normal power of two, else 0 (kPow2Words blocks:
// every word of the call's slot is rewritten, nothing to reset).  HBM-bound, a few microseconds: C4's a_scale is 4 MB.
__global__ void __launch_bounds__(256) scales_pow2_check_kernel(const uint32_t* a_scale, int64_t na, const uint32_t* b_scale,
                                                                  int64_t nb, uint32_t* flag) {
  __shared__ uint32_t any_bad;
  if (threadIdx.x == 0) any_bad = 0;
  __syncthreads();
  bool bad = false;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += stride) {
    const uint32_t w = i < na ? a_scale[i] : b_scale[i - na];
    const uint32_t e = w >> 23;  // sign | exponent
    bad |= (w & 0x007fffffu) != 0 || e == 0 || e >= 255;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) any_bad = 1;  // benign race: every writer stores 1
  __syncthreads();
  if (threadIdx.x == 0) flag[blockIdx.x] = any_bad;
}

// HWS: the scales are powers of two (checked on the device, p.pow2_flag): their exponents go to the MFMA as E8M0
// block scales -- v_mfma_scale computes sum (a 2^sa)(b 2^sb) -- and every k block accumulates straight into the
// tile's accumulators: no partial products, no fold (128 FMAs per 16 MFMAs and wave in the general path, 16 % of
// the C4 launch).  Both instantiations are launched; the one the flag does not select returns at once.
template <bool MA_E5M2, bool MB_E5M2, bool HWS>
__global__ void __launch_bounds__(kBigThreads, 1) group_gemm_fp8_big_kernel(const GemmParams p) {
  {
    const bool pow2 = fi_scales_are_pow2(p.pow2_flag);
    if (pow2 != HWS) return;
  }
  // ONE array: the compiler tells a DMA target from an LDS read by constant offsets inside one object
  __shared__ __attribute__((aligned(1024))) uint8_t smem[2 * kBigStage + 768];  // 2 stages, then the group table
  constexpr int kGroupTab = 2 * kBigStage;
  constexpr int kBOff = kBigTM * kBK;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int lq = lane & 31, lh = lane >> 5;
  const int K = p.k, N = p.n;
  const int kblocks = K / kBK;
  auto lds_off = [](int row, int ch) { return row * kBK + ((ch ^ ((row >> 1) & 7)) << 4); };
  const int m_cnt = p.a_gran_m == 1 ? p.m_total : (p.m_total + p.a_gran_m - 1) / p.a_gran_m;
  const int a_sc_stride = p.scale_k_major ? 1 : m_cnt;
  const int n_sblocks = (N + 127) / 128;
  const int b_sc_stride = p.scale_k_major ? 1 : n_sblocks;
  uint32_t a_rd_base = (uint32_t)lds_off(64 * wm + lq, 2 * lh);
  uint32_t b_rd_base = (uint32_t)(kBOff + lds_off(128 * wn + lq, 2 * lh));

  // ---- persistent workgroup: the tiles of this XCD, strided by its workgroups ----
  // Tiles are ordered in bands of kBandM m tiles x all n tiles (m fastest inside a band, see gemm.hip).  The bands are
  // dealt to the 8 XCDs INTERLEAVED: XCD x takes bands x, x + 8, x + 16 ... -- at any time the chip works on 8
  // CONSECUTIVE bands, i.e. on one or two groups, whose B matrices (C4: 59 MB each) stay in the 256 MB Infinity
  // Cache while every XCD streams them, instead of eight groups' worth (470 MB) going to HBM each time (r2 gave
  // XCD x a contiguous tile range, which at C4 is group x for the whole launch).
  const int n_tiles = (N + kBigTN - 1) / kBigTN;
  const int total = p.num_m_tiles_bound * n_tiles;
  constexpr int kBandM = FI_GEMM_BIG_BAND / kBigTM;  // 1024 rows x all n per band
  const int band_tiles = kBandM * n_tiles;
  // whole rounds of 8 bands are dealt one band per XCD; what is left (fewer than 8 bands, the last one possibly
  // short) is split into contiguous ranges so that every XCD gets the same number of tiles (+- 1)
  const int n_inter = FI_GEMM_BIG_INTERLEAVE ? (p.num_m_tiles_bound / (8 * kBandM)) * band_tiles : 0;  // per XCD
  const int logical_step = gridDim.x >> 3;
  const int xcd_id = blockIdx.x & 7;
  int logical = blockIdx.x >> 3, logical_end, rem_base;
  {
    const int rem_total = total - 8 * n_inter;
    const int qn = rem_total >> 3, rn = rem_total & 7;
    rem_base = 8 * n_inter + (xcd_id < rn ? xcd_id * (qn + 1) : rn * (qn + 1) + (xcd_id - rn) * qn);
    logical_end = n_inter + qn + (xcd_id < rn ? 1 : 0);
  }

  // up to 64 groups: group i's row range and first tile live in LDS (a register copy per lane would stay
  // live across the k loop), read back by lane i at every tile border -- no global round trip per tile
  const bool groups_cached = p.m_indptr != nullptr && p.num_groups <= 64;
  int32_t* const group_tab = (int32_t*)&smem[kGroupTab];  // [3][64]: lo, hi, first tile
  if (groups_cached && wave == 0) {
    const bool in = lane < p.num_groups;
    const int lo = in ? p.m_indptr[lane] : 0;
    const int hi = in ? p.m_indptr[lane + 1] : 0;
    const int tiles = (hi - lo + kBigTM - 1) / kBigTM;
    int incl = tiles;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    group_tab[lane] = lo;
    group_tab[64 + lane] = hi;
    group_tab[128 + lane] = incl - tiles;
  }
  __syncthreads();

  f32x16g acc[4][2];  // [n block][m block]; register r of a block: n = 8 (r / 4) + 4 lh + r % 4, m = lq
  int out_m0 = 0, out_n0 = 0, out_m_end = 0;  // tile whose accumulators are waiting to be stored
  bool have_out = false;
  const bool d_aligned16 = (((uintptr_t)p.d) & 15) == 0;
  // Output: the accumulators are transposed (m on the lane).  Each wave turns its 64 x 128 block, 64 columns at
  // a time, through its 8 KB of stage 1's operand area ([64 rows][128 B], 16-byte chunks swizzled like the
  // operand images) and stores whole 128-byte row pieces, 16 bytes per lane.
  auto store_tile = [&]() {
    uint8_t* const scratch = &smem[kBigStage] + wave * 8192;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int nbl = 0; nbl < 2; ++nbl)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            uint32_t w[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const uint32_t lo = f32_to_16bit(acc[2 * h + nbl][mb][4 * r4 + 2 * e], p.out_dtype);
              const uint32_t hi = f32_to_16bit(acc[2 * h + nbl][mb][4 * r4 + 2 * e + 1], p.out_dtype);
              w[e] = lo | (hi << 16);
            }
            const int row = 32 * mb + lq;
            *(u32x2*)(scratch + lds_off(row, 4 * nbl + r4) + 8 * lh) = u32x2{w[0], w[1]};
          }
#pragma unroll
      for (int i2 = 0; i2 < 8; ++i2) {
        const int idx = lane + 64 * i2;
        const int r = idx >> 3, c = idx & 7;
        const u32x4 v = *(const u32x4*)(scratch + lds_off(r, c));
        const int m = out_m0 + 64 * wm + r;
        const int n = out_n0 + 128 * wn + 64 * h + 8 * c;
        if (m >= out_m_end || n >= N) continue;  // n is a multiple of 8 and so is N
        uint16_t* dst = (uint16_t*)p.d + (int64_t)m * N + n;
        if (d_aligned16) {
          if (FI_GEMM_BIG_NT_STORE) __builtin_nontemporal_store(v, (u32x4*)dst);
          else *(u32x4*)dst = v;
        } else {
          *(u32x2*)dst = u32x2{v[0], v[1]};
          *(u32x2*)(dst + 4) = u32x2{v[2], v[3]};
        }
      }
    }
  };

  for (; logical < logical_end; logical += logical_step) {
    int band, in_band;
    if (logical < n_inter) {  // the XCD's own band of round band_l
      const int band_l = logical / band_tiles;
      in_band = logical - band_l * band_tiles;
      band = band_l * 8 + xcd_id;
    } else {
      const int gl = rem_base + (logical - n_inter);
      band = gl / band_tiles;
      in_band = gl - band * band_tiles;
    }
    const int band_m = min(kBandM, p.num_m_tiles_bound - band * kBandM);
    const int nt = in_band / band_m;
    const int mt_global = band * kBandM + (in_band - nt * band_m);
    int g = 0, m_begin = 0, m_end = p.m_total, mt = mt_global;
    bool found = true;
    if (p.m_indptr) {
      if (groups_cached) {
        const int gc_lo = group_tab[lane], gc_hi = group_tab[64 + lane], gc_start = group_tab[128 + lane];
        const int tiles = (gc_hi - gc_lo + kBigTM - 1) / kBigTM;
        const uint64_t hit = __ballot(mt_global >= gc_start && mt_global < gc_start + tiles);
        found = hit != 0;
        if (found) {
          g = __builtin_ctzll(hit);
          m_begin = __builtin_amdgcn_readlane(gc_lo, g);
          m_end = __builtin_amdgcn_readlane(gc_hi, g);
          mt = mt_global - __builtin_amdgcn_readlane(gc_start, g);
        }
      } else {
        found = find_group_tile<kBigTM>(p.m_indptr, p.num_groups, mt_global, lane, g, m_begin, m_end, mt);
      }
    } else if (mt_global * kBigTM >= p.m_total) {
      found = false;
    }
    if (!found) continue;  // the grid bound counts one partial tile per group; uniform per workgroup
    const int m0 = m_begin + mt * kBigTM;
    const int n0 = nt * kBigTN;
    const uint8_t* Bg = p.b + (int64_t)g * N * K;

    // LDS-DMA geometry: the stage image is [512 rows][128 B] (A rows 0-255, B rows 256-511); a piece = one wave
    // instruction = 8 rows = 1 KiB, written linearly (lane i -> byte 16 i of the piece), the XOR swizzle applied
    // on the GLOBAL side: lane i fetches chunk (i & 7) ^ key(row) of row (i >> 3).  Wave w owns pieces
    // 8 w .. 8 w + 7 = image rows 64 w .. 64 w + 63: waves 0-3 fetch A, waves 4-7 fetch B.
    // The source is a buffer descriptor over the tile's valid rows: rows past the group's end (or past N) fail
    // the hardware range check and arrive as zeros -- no clamped per-piece offsets to keep in registers.  The
    // check covers the VGPR offset only, so the row goes there and the k offset into the scalar offset.
    // key(row) = (row >> 1) & 7 = 4 (j & 1) | (lane >> 4) for piece j: odd pieces flip byte-offset bit 6.
    const int rows_valid = wave < 4 ? min(m_end - m0, kBigTM) : min(N - n0, kBigTN);
    const uint8_t* const src_tile = wave < 4 ? p.a + (int64_t)m0 * K : Bg + (int64_t)n0 * K;
    const auto src_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)src_tile, 0, __builtin_amdgcn_readfirstlane(rows_valid * K), 0x00020000);
    const uint32_t v_par0 = (uint32_t)(64 * (wave & 3) + (lane >> 3)) * (uint32_t)K + (((lane & 7) ^ (lane >> 4)) << 4);
    // the block's scales travel the same way (4 bytes per lane; see gemm.hip): lane L fetches the A scale of
    // row 64 wm + L of the tile, every lane the B scale of this wave's 128-wide block
    const int nsb = min((n0 + 128 * wn) / 128, n_sblocks - 1);
    const auto b_sc_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.scale_k_major ? p.b_scale + ((int64_t)g * n_sblocks + nsb) * kblocks
                                : p.b_scale + (int64_t)g * kblocks * n_sblocks + nsb),
        0, 0x7fffffff, 0x00020000);
    const auto a_sc_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.a_scale, 0, 0x7fffffff, 0x00020000);
    uint32_t a_sc_voff;
    {
      const int m = min(m0 + 64 * wm + lane, m_end - 1);
      const int mi = p.a_gran_m == 1 ? m : m / p.a_gran_m;
      a_sc_voff = (uint32_t)(p.scale_k_major ? mi * kblocks : mi) * 4u;
    }
    // oob = 0x80000000 pushes every lane's offset past the descriptor's range: the pieces are then written as zeros
    // without a memory access (see k_step)
    auto dma_pieces = [&](int kb, int stage, auto first_c, auto count_c, uint32_t oob = 0) {
      constexpr int first = decltype(first_c)::value, count = decltype(count_c)::value;
      const int koff = kb * kBK;
      uint32_t vb = v_par0 + oob;
      asm volatile("" : "+v"(vb));  // the piece offsets are recomputed here, not kept across the k loop
#pragma unroll
      for (int j2 = first; j2 < first + count; ++j2) {
        const uint32_t voff = (vb ^ ((j2 & 1) << 6)) + (uint32_t)(8 * j2) * (uint32_t)K;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            src_rsrc, (__attribute__((address_space(3))) void*)(&smem[stage * kBigStage + (8 * wave + j2) * 1024]), 16, voff,
            koff, 0, 0);
      }
    };
    auto dma_scales = [&](int kb, int stage, uint32_t oob = 0) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          a_sc_rsrc, (__attribute__((address_space(3))) void*)(&smem[stage * kBigStage + kBigScOff + wave * 512]), 4,
          a_sc_voff + oob, kb * a_sc_stride * 4, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          b_sc_rsrc, (__attribute__((address_space(3))) void*)(&smem[stage * kBigStage + kBigScOff + wave * 512 + 256]),
          4, oob, kb * b_sc_stride * 4, 0, 0);
    };
    auto dma = [&](int kb, int stage, uint32_t oob = 0) {
      dma_pieces(kb, stage, std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{}, oob);
      dma_scales(kb, stage, oob);
    };
    // block 0 of the next tile goes out BEFORE the finished tile is converted and stored (stage 0; the store's
    // scratch is stage 1, which block 1 enters only in step 0, behind the barrier below)
    dma(0, 0);
    if (have_out && !(FI_GEMM_BIG_KO & 1)) store_tile();
    out_m0 = m0;
    out_n0 = n0;
    out_m_end = m_end;
    have_out = true;
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i2][j2][r] = 0.f;
    // block 0 landed (the output stores are younger and vmcnt retires in order: 0 is the only safe count);
    // every wave is done with its stage 1 scratch
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // Hardware-scale path (FI_GEMM_BIG_CARRY): the last two MFMAs of a step (n block 3, second k half) are held
    // back and issued BEHIND the barrier, in front of the next step's first MFMA -- the matrix pipe has work while
    // the new step's DMA instructions issue and its first fragment reads are in flight.  Their operands (the A
    // fragments' second k halves, one B fragment, the step's scale exponents) stay in the registers they are in;
    // the new step's reads go to the first-k-half registers first.  Zero operands in front of step 0.
    i32x8g hfa[2][2];  // [m block][k half]
    i32x8g hfb[2][2];  // [n block parity][k half]
    int he_b = 127, he_a0 = 127, he_a1 = 127;
    if constexpr (HWS && FI_GEMM_BIG_CARRY) {
#pragma unroll
      for (int r = 0; r < 8; ++r) hfa[0][1][r] = hfa[1][1][r] = hfb[1][1][r] = 0;
    }
    f32x16g p_carry;  // block (3, 1) of the previous k step, folded at the start of the next one
#pragma unroll
    for (int r = 0; r < 16; ++r) p_carry[r] = 0.f;
    float s_carry = 0.f;

    auto k_step = [&](auto par_c, const int kb) {
      constexpr int buf = decltype(par_c)::value;  // == kb % 2
      // Block kb + 1 goes into the stage read in step kb - 1.  No branch on "is there a block kb + 1": a branch
      // here splits the step's scheduling region (the allocator then spills around it, and the last step's copy
      // of the loop body cost 7 % of C4).  The LAST step issues the same instructions with every lane's offset
      // pushed out of the descriptor's range: zeros, written without a memory access into the stage nobody reads.
      auto dma_next = [&]() {
        const uint32_t oob = kb + 1 < kblocks ? 0u : 0x80000000u;
        if (!(FI_GEMM_BIG_KO & 4) || kb == 0) dma(min(kb + 1, kblocks - 1), buf ^ 1, oob);
      };
      constexpr bool kDmaBehindCarry = HWS && FI_GEMM_BIG_CARRY == 2;
      if (FI_GEMM_BIG_BRANCHFREE & (HWS ? 1 : 2)) {
        if (!kDmaBehindCarry) dma_next();
      } else if (kb + 1 < kblocks && (!(FI_GEMM_BIG_KO & 4) || kb == 0)) {
        dma(kb + 1, buf ^ 1);
      }
      asm volatile("" : "+v"(a_rd_base), "+v"(b_rd_base));
      const uint8_t* const stage = &smem[buf * kBigStage];
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
        if (FI_GEMM_BIG_KO & 2) {
          acc[nb][mb][0] += s * part[0];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[nb][mb][r] += s * part[r];
        }
        asm volatile("" : "+v"(acc[nb][mb]));  // IR-level sinking ignores sched_barrier: pin the fold here
      };
      i32x8g fa[2][2];  // [m block][k half], held for the step
      i32x8g fb[2][2];  // [n block parity][k half], streamed one n block ahead
      // region 0: fragment reads first, the block's scales behind them (first needed at the first fold); the
      // fold carried over from the previous k step runs while the reads are in flight (carrying both blocks of
      // the last n block spilled inside the loop)
      fa[0][0] = frag(a_rd_base, 0, 0);
      fb[0][0] = frag(b_rd_base, 0, 0);
      fa[0][1] = frag(a_rd_base, 1, 0);
      fb[0][1] = frag(b_rd_base, 1, 0);
      fa[1][0] = frag(a_rd_base, 0, 1);
      fa[1][1] = frag(a_rd_base, 1, 1);
      const float* const sc = (const float*)(&smem[buf * kBigStage + kBigScOff + wave * 512]);
      const float sa[2] = {sc[lq], sc[32 + lq]};
      const float sb = sc[64 + lane];
      if constexpr (HWS && FI_GEMM_BIG_CARRY) {
        const float* const sc = (const float*)(&smem[buf * kBigStage + kBigScOff + wave * 512]);
        const float sa0 = sc[lq], sa1 = sc[32 + lq], sb = sc[64 + lane];
        hfa[0][0] = frag(a_rd_base, 0, 0);
        hfb[0][0] = frag(b_rd_base, 0, 0);
        hfa[1][0] = frag(a_rd_base, 0, 1);
        auto mfma_c = [&](const i32x8g& b, const i32x8g& a, const f32x16g& c, int eb, int ea) {
          return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c, MA_E5M2 ? 1 : 0, MB_E5M2 ? 1 : 0, 0, eb, 0, ea);
        };
        __builtin_amdgcn_sched_barrier(0);
        // the previous step's held-back pair
        acc[3][0] = mfma_c(hfb[1][1], hfa[0][1], acc[3][0], he_b, he_a0);
        acc[3][1] = mfma_c(hfb[1][1], hfa[1][1], acc[3][1], he_b, he_a1);
        __builtin_amdgcn_sched_barrier(0);
        if (kDmaBehindCarry) {
          dma_next();
          __builtin_amdgcn_sched_barrier(0);
        }
        hfa[0][1] = frag(a_rd_base, 1, 0);
        hfb[0][1] = frag(b_rd_base, 1, 0);
        hfa[1][1] = frag(a_rd_base, 1, 1);
        he_b = (int)(__builtin_bit_cast(uint32_t, sb) >> 23);
        he_a0 = (int)(__builtin_bit_cast(uint32_t, sa0) >> 23);
        he_a1 = (int)(__builtin_bit_cast(uint32_t, sa1) >> 23);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int cur = nb & 1;
          acc[nb][0] = mfma_c(hfb[cur][0], hfa[0][0], acc[nb][0], he_b, he_a0);
          if (nb < 3) {
            hfb[cur ^ 1][0] = frag(b_rd_base, 0, nb + 1);
            hfb[cur ^ 1][1] = frag(b_rd_base, 1, nb + 1);
          }
          acc[nb][1] = mfma_c(hfb[cur][0], hfa[1][0], acc[nb][1], he_b, he_a1);
          if (nb < 3) {
            acc[nb][0] = mfma_c(hfb[cur][1], hfa[0][1], acc[nb][0], he_b, he_a0);
            acc[nb][1] = mfma_c(hfb[cur][1], hfa[1][1], acc[nb][1], he_b, he_a1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        return;
      }
      if constexpr (HWS) {
        // E8M0 = the f32 exponent field of a power of two.  MFMA A operand = B matrix rows (one scale per wave: sb),
        // MFMA B operand = A matrix rows, the lane's row (sa)
        const int e_b = (int)(__builtin_bit_cast(uint32_t, sb) >> 23);
        const int e_a0 = (int)(__builtin_bit_cast(uint32_t, sa[0]) >> 23), e_a1 = (int)(__builtin_bit_cast(uint32_t, sa[1]) >> 23);
        auto mfma_s = [&](const i32x8g& b, const i32x8g& a, const f32x16g& c, int ea) {
          return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c, MA_E5M2 ? 1 : 0, MB_E5M2 ? 1 : 0, 0, e_b, 0, ea);
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int cur = nb & 1;
          acc[nb][0] = mfma_s(fb[cur][0], fa[0][0], acc[nb][0], e_a0);
          if (nb < 3) {
            fb[cur ^ 1][0] = frag(b_rd_base, 0, nb + 1);
            fb[cur ^ 1][1] = frag(b_rd_base, 1, nb + 1);
          }
          acc[nb][1] = mfma_s(fb[cur][0], fa[1][0], acc[nb][1], e_a1);
          acc[nb][0] = mfma_s(fb[cur][1], fa[0][1], acc[nb][0], e_a0);
          acc[nb][1] = mfma_s(fb[cur][1], fa[1][1], acc[nb][1], e_a1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (FI_GEMM_BIG_KO & 8) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (FI_GEMM_BIG_KO & 32) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        return;
      }
      fold(3, 1, p_carry, s_carry);
      __builtin_amdgcn_sched_barrier(0);
      const float s0 = sa[0] * sb, s1 = sa[1] * sb;
      f32x16g p1 = p_carry;
      // The two MFMAs of a block are dependent (same accumulator): the previous block's fold and the next n
      // block's fragment reads sit between them.
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int cur = nb & 1;
        f32x16g p0 = mfma0(fb[cur][0], fa[0][0]);
        __builtin_amdgcn_sched_barrier(0);
        if (nb < 3) {
          fb[cur ^ 1][0] = frag(b_rd_base, 0, nb + 1);
          fb[cur ^ 1][1] = frag(b_rd_base, 1, nb + 1);
        }
        if (nb > 0) fold(nb - 1, 1, p1, s1);
        __builtin_amdgcn_sched_barrier(0);
        p0 = mfma1(fb[cur][1], fa[0][1], p0);
        p1 = mfma0(fb[cur][0], fa[1][0]);
        __builtin_amdgcn_sched_barrier(0);
        fold(nb, 0, p0, s0);
        __builtin_amdgcn_sched_barrier(0);
        p1 = mfma1(fb[cur][1], fa[1][1], p1);
      }
      p_carry = p1;
      s_carry = s1;
      // block kb + 1 landed (every wave's pieces: barrier), nobody still reads this stage
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    {
      int kb = 0;
      for (; kb + 1 < kblocks; kb += 2) {
        k_step(std::integral_constant<int, 0>{}, kb);
        k_step(std::integral_constant<int, 1>{}, kb + 1);
      }
      if (kb < kblocks) k_step(std::integral_constant<int, 0>{}, kb);
    }
    if constexpr (!HWS) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[3][1][r] += s_carry * p_carry[r];
    } else if constexpr (FI_GEMM_BIG_CARRY) {
      acc[3][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(hfb[1][1], hfa[0][1], acc[3][0], MA_E5M2 ? 1 : 0,
                                                                  MB_E5M2 ? 1 : 0, 0, he_b, 0, he_a0);
      acc[3][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(hfb[1][1], hfa[1][1], acc[3][1], MA_E5M2 ? 1 : 0,
                                                                  MB_E5M2 ? 1 : 0, 0, he_b, 0, he_a1);
    }
  }
  if (have_out && (!(FI_GEMM_BIG_KO & 1) || p.k == 12345)) store_tile();
}

// flag words for the power-of-two check: a ring of slots per device (a call's three kernels read / write ITS slot;
// calls in flight on other streams use other slots), allocated at the first call outside a stream capture
static uint32_t* pow2_flag_slot(hipStream_t stream) {
  constexpr int kSlots = 1024;  // x kPow2Words words
  static uint32_t* ring[64] = {nullptr};
  static std::atomic<unsigned> next{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (ring[dev] == nullptr) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (ring[dev] == nullptr) {
      uint32_t* ptr = nullptr;
      if (hipMalloc(&ptr, (size_t)kSlots * kPow2Words * sizeof(uint32_t)) != hipSuccess) return nullptr;
      ring[dev] = ptr;
    }
  }
  return ring[dev] + (size_t)(next.fetch_add(1) % kSlots) * kPow2Words;
}

template <bool HWS>
static void launch_big_variant(const GemmParams& p, int sel, int grid, hipStream_t stream) {
  switch (sel) {
    case 0: group_gemm_fp8_big_kernel<false, false, HWS><<<dim3(grid), dim3(kBigThreads), 0, stream>>>(p); break;
    case 1: group_gemm_fp8_big_kernel<false, true, HWS><<<dim3(grid), dim3(kBigThreads), 0, stream>>>(p); break;
    case 2: group_gemm_fp8_big_kernel<true, false, HWS><<<dim3(grid), dim3(kBigThreads), 0, stream>>>(p); break;
    default: group_gemm_fp8_big_kernel<true, true, HWS><<<dim3(grid), dim3(kBigThreads), 0, stream>>>(p); break;
  }
}

hipError_t launch_gemm_big(const GemmParams& p_in, int grid, hipStream_t stream, uint32_t** hws_only_flag) {
  // MFMA A operand = GEMM matrix B, MFMA B operand = GEMM matrix A
  GemmParams p = p_in;
  const int sel = (p.b_is_e5m2 ? 2 : 0) | (p.a_is_e5m2 ? 1 : 0);
  // FI_GEMM_HW_SCALES=0: never take the hardware-scale path (A/B runs)
  static const bool hw_scales = [] {
    const char* e = getenv("FI_GEMM_HW_SCALES");
    return !(e && atoi(e) == 0);
  }();
  uint32_t* flag = hw_scales ? pow2_flag_slot(stream) : nullptr;
  p.pow2_flag = flag;
  if (flag != nullptr) {
    const int kblocks = p.k / kBK;
    const int64_t m_cnt = p.a_gran_m == 1 ? p.m_total : (p.m_total + p.a_gran_m - 1) / p.a_gran_m;
    const int64_t na = m_cnt * kblocks;
    const int64_t nb = (int64_t)(p.m_indptr ? p.num_groups : 1) * kblocks * ((p.n + 127) / 128);
    scales_pow2_check_kernel<<<dim3(kPow2Words), dim3(256), 0, stream>>>((const uint32_t*)p.a_scale, na, (const uint32_t*)p.b_scale,
                                                                  nb, flag);
    launch_big_variant<true>(p, sel, grid, stream);
  }
  if (hws_only_flag != nullptr) *hws_only_flag = flag;
  else launch_big_variant<false>(p, sel, grid, stream);
  return hipGetLastError();
}

}  // namespace fi

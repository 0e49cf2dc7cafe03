// Shared host/device helpers for libfi_mi355.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fi_mi355.h"

namespace fi {

// ---- host-side error channel (ref: include/flashinfer/exception.h:23, 74-87) ----
int set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
const char* last_error();

#define FI_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) return ::fi::set_error(__VA_ARGS__); \
  } while (0)

#define FI_HIP_CALL(expr)                                                                 \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess)                                                                \
      return ::fi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                             __LINE__);                                                   \
  } while (0)

inline size_t dtype_size(int dt) {
  switch (dt) {
    case FI_DTYPE_F16:
    case FI_DTYPE_BF16:
      return 2;
    case FI_DTYPE_FP8_E4M3:
    case FI_DTYPE_FP8_E5M2:
      return 1;
    case FI_DTYPE_F32:
      return 4;
    default:
      return 0;
  }
}

template <typename T>
inline T ceil_div(T a, T b) {
  return (a + b - 1) / b;
}

// 16-byte aligned bump allocator over a caller-owned workspace; returns byte offsets.
// ref: AlignedAllocator, include/flashinfer/allocator.h:32-59.
struct OffsetAllocator {
  size_t cap;
  size_t used = 0;
  bool ok = true;
  explicit OffsetAllocator(size_t capacity) : cap(capacity) {}
  int64_t alloc(size_t bytes, size_t align = 16) {
    size_t start = (used + align - 1) / align * align;
    if (start + bytes > cap) {
      ok = false;
      return 0;
    }
    used = start + bytes;
    return (int64_t)start;
  }
};

// Division by a runtime-constant 32-bit divisor using a precomputed multiplier (round-up method,
// Granlund & Montgomery 1994): q = (umulhi(n, m) + n) >> l, valid for n < 2^31.
struct FastDiv {
  uint32_t d, m, l;
  FastDiv() : d(1), m(0), l(0) {}
  explicit FastDiv(uint32_t divisor) : d(divisor) {
    l = 0;
    while ((1ull << l) < d) ++l;
    m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  }
};

}  // namespace fi

#if defined(__HIPCC__)
namespace fi {

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv& fd) {
  return (__umulhi(n, fd.m) + n) >> fd.l;
}

constexpr float kLog2e = 1.44269504088896340736f;

// accurate sin/cos kept out of line: only used to seed rotations / in the fused-RoPE variants
__device__ __attribute__((noinline)) static void sincos_ool(float x, float* sn, float* cs) {
  sincosf(x, sn, cs);
}

// sin / cos of an angle in radians: x = k * 2pi + y with |y| <= pi (two-term Cody-Waite reduction), then
// the hardware v_sin / v_cos on y / 2pi.  Absolute error ~1e-6 for |x| up to ~1e5 (RoPE angles).
__device__ __forceinline__ void fast_sincos(float x, float* sn, float* cs) {
  const float k = rintf(x * 0.15915494309189535f);
  float y = __builtin_fmaf(-k, 6.2831854820251465f, x);       // 2pi high part (float)
  y = __builtin_fmaf(-k, -1.7484555e-07f, y);                 // 2pi - float(2pi)
  const float rev = y * 0.15915494309189535f;
  *sn = __builtin_amdgcn_sinf(rev);
  *cs = __builtin_amdgcn_cosf(rev);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// tanh(x) = 1 - 2 / (2^(2 x log2e) + 1)   (ref uses tanh.approx: include/flashinfer/math.cuh:120-150)
__device__ __forceinline__ float fast_tanh(float x) {
  float e = fast_exp2(x * 2.885390081777927f);
  return 1.0f - 2.0f * fast_rcp(e + 1.0f);
}

// ---- cross-lane moves ----
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;   // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;   // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_ROR8 = 0x128;

// value of lane (lane ^ MASK)
template <int MASK>
__device__ __forceinline__ float lane_xor(float x) {
  if constexpr (MASK == 1)
    return dpp_mov<DPP_QUAD_XOR1>(x);
  else if constexpr (MASK == 2)
    return dpp_mov<DPP_QUAD_XOR2>(x);
  else if constexpr (MASK == 8)
    return dpp_mov<DPP_ROW_ROR8>(x);
  else
    return __shfl_xor(x, MASK, 64);
}

// Sum over groups of N consecutive lanes (N a power of two, groups aligned); every lane of the group
// receives the total.  Steps inside a 16-lane row are DPP adds (no LDS traffic).
template <int N>
__device__ __forceinline__ float group_sum(float x) {
  if constexpr (N >= 2) x += dpp_mov<DPP_QUAD_XOR1>(x);
  if constexpr (N >= 4) x += dpp_mov<DPP_QUAD_XOR2>(x);
  if constexpr (N >= 8) x += dpp_mov<DPP_ROW_HALF_MIRROR>(x);
  if constexpr (N >= 16) x += dpp_mov<DPP_ROW_MIRROR>(x);
  if constexpr (N >= 32) x += __shfl_xor(x, 16, 64);
  if constexpr (N >= 64) x += __shfl_xor(x, 32, 64);
  return x;
}

// ---- storage-type unpack: 16 bytes -> VEC floats ----
template <int DT>
struct KVTraits;

template <>
struct KVTraits<FI_DTYPE_BF16> {
  static constexpr int BYTES = 2, VEC = 8;
  static __device__ __forceinline__ void unpack(const u32x4& r, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __builtin_bit_cast(float, r[i] << 16);
      f[2 * i + 1] = __builtin_bit_cast(float, r[i] & 0xffff0000u);
    }
  }
};
template <>
struct KVTraits<FI_DTYPE_F16> {
  static constexpr int BYTES = 2, VEC = 8;
  static __device__ __forceinline__ void unpack(const u32x4& r, float* f) {
    using h2 = __attribute__((ext_vector_type(2))) _Float16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t w = r[i];  // copy first: bit_cast of a vector-element lvalue reads element 0
      h2 h = __builtin_bit_cast(h2, w);
      f[2 * i] = (float)h[0];
      f[2 * i + 1] = (float)h[1];
    }
  }
};
template <>
struct KVTraits<FI_DTYPE_FP8_E4M3> {
  static constexpr int BYTES = 1, VEC = 16;
  static __device__ __forceinline__ void unpack(const u32x4& r, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[i], false);
      f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[i], true);
      f[4 * i] = lo[0];
      f[4 * i + 1] = lo[1];
      f[4 * i + 2] = hi[0];
      f[4 * i + 3] = hi[1];
    }
  }
};
template <>
struct KVTraits<FI_DTYPE_FP8_E5M2> {
  static constexpr int BYTES = 1, VEC = 16;
  static __device__ __forceinline__ void unpack(const u32x4& r, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x2 lo = __builtin_amdgcn_cvt_pk_f32_bf8((int)r[i], false);
      f32x2 hi = __builtin_amdgcn_cvt_pk_f32_bf8((int)r[i], true);
      f[4 * i] = lo[0];
      f[4 * i + 1] = lo[1];
      f[4 * i + 2] = hi[0];
      f[4 * i + 3] = hi[1];
    }
  }
};

// 16-bit float load / store with a runtime dtype (FI_DTYPE_F16 / FI_DTYPE_BF16).
__device__ __forceinline__ float load_f16_or_bf16(const void* p, int64_t idx, int dt) {
  if (dt == FI_DTYPE_BF16) {
    uint32_t u = ((const uint16_t*)p)[idx];
    return __builtin_bit_cast(float, u << 16);
  }
  return (float)((const _Float16*)p)[idx];
}
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float x) {
  return __builtin_bit_cast(uint16_t, (__bf16)x);
}
__device__ __forceinline__ uint16_t f32_to_f16_bits(float x) {
  return __builtin_bit_cast(uint16_t, (_Float16)x);
}
__device__ __forceinline__ uint16_t f32_to_16bit(float x, int dt) {
  return dt == FI_DTYPE_BF16 ? f32_to_bf16_bits(x) : f32_to_f16_bits(x);
}
__device__ __forceinline__ float load_any_float(const void* p, int64_t idx, int dt) {
  if (dt == FI_DTYPE_F32) return ((const float*)p)[idx];
  return load_f16_or_bf16(p, idx, dt);
}
__device__ __forceinline__ void store_any_float(void* p, int64_t idx, float x, int dt) {
  if (dt == FI_DTYPE_F32)
    ((float*)p)[idx] = x;
  else
    ((uint16_t*)p)[idx] = f32_to_16bit(x, dt);
}

}  // namespace fi
#endif  // __HIPCC__

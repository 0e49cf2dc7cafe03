// Does an MFMA whose operands live in ACCUMULATOR registers (AGPRs) take less of the SIMD's vector issue / register
// bandwidth than one on VGPRs?  Per trip: 4 MX-fp8 32x32x64 MFMAs + 128 independent vector instructions (v_fma, or
// 96 v_fma + 32 v_exp), explicit asm; 1 / 2 / 4 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_agpr.hip -o build/ubench_agpr
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using f32x16 = __attribute__((ext_vector_type(16))) float;
using i32x8 = __attribute__((ext_vector_type(8))) int;
constexpr int ITER = 4000;
struct Stamp { unsigned long long cyc, rt; };

#define FMA8(d0) \
  "v_fma_f32 %" #d0 ", %" #d0 ", %12, %13\n\t"
#define V32_FMA \
  "v_fma_f32 %0, %0, %12, %13\n\tv_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\tv_fma_f32 %3, %3, %12, %13\n\t" \
  "v_fma_f32 %4, %4, %12, %13\n\tv_fma_f32 %5, %5, %12, %13\n\tv_fma_f32 %6, %6, %12, %13\n\tv_fma_f32 %7, %7, %12, %13\n\t" \
  "v_fma_f32 %8, %8, %12, %13\n\tv_fma_f32 %9, %9, %12, %13\n\tv_fma_f32 %10, %10, %12, %13\n\tv_fma_f32 %11, %11, %12, %13\n\t" \
  "v_fma_f32 %0, %0, %12, %13\n\tv_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\tv_fma_f32 %3, %3, %12, %13\n\t" \
  "v_fma_f32 %4, %4, %12, %13\n\tv_fma_f32 %5, %5, %12, %13\n\tv_fma_f32 %6, %6, %12, %13\n\tv_fma_f32 %7, %7, %12, %13\n\t" \
  "v_fma_f32 %8, %8, %12, %13\n\tv_fma_f32 %9, %9, %12, %13\n\tv_fma_f32 %10, %10, %12, %13\n\tv_fma_f32 %11, %11, %12, %13\n\t" \
  "v_fma_f32 %0, %0, %12, %13\n\tv_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\tv_fma_f32 %3, %3, %12, %13\n\t" \
  "v_fma_f32 %4, %4, %12, %13\n\tv_fma_f32 %5, %5, %12, %13\n\tv_fma_f32 %6, %6, %12, %13\n\tv_fma_f32 %7, %7, %12, %13\n\t"
// 24 v_fma + 8 v_exp (the softmax mix)
#define V32_MIX \
  "v_fma_f32 %0, %0, %12, %13\n\tv_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\tv_exp_f32 %3, %3\n\t" \
  "v_fma_f32 %4, %4, %12, %13\n\tv_fma_f32 %5, %5, %12, %13\n\tv_fma_f32 %6, %6, %12, %13\n\tv_exp_f32 %7, %7\n\t" \
  "v_fma_f32 %8, %8, %12, %13\n\tv_fma_f32 %9, %9, %12, %13\n\tv_fma_f32 %10, %10, %12, %13\n\tv_exp_f32 %11, %11\n\t" \
  "v_fma_f32 %0, %0, %12, %13\n\tv_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\tv_exp_f32 %3, %3\n\t" \
  "v_fma_f32 %4, %4, %12, %13\n\tv_fma_f32 %5, %5, %12, %13\n\tv_fma_f32 %6, %6, %12, %13\n\tv_exp_f32 %7, %7\n\t" \
  "v_fma_f32 %8, %8, %12, %13\n\tv_fma_f32 %9, %9, %12, %13\n\tv_fma_f32 %10, %10, %12, %13\n\tv_exp_f32 %11, %11\n\t" \
  "v_fma_f32 %0, %0, %12, %13\n\tv_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\tv_exp_f32 %3, %3\n\t" \
  "v_fma_f32 %4, %4, %12, %13\n\tv_fma_f32 %5, %5, %12, %13\n\tv_fma_f32 %6, %6, %12, %13\n\tv_exp_f32 %7, %7\n\t"
#define MF(acc) "v_mfma_scale_f32_32x32x64_f8f6f4 %" #acc ", %16, %17, %" #acc ", %18, %18 op_sel_hi:[0,0,0]\n\t"
#define MFZ(acc) "v_mfma_scale_f32_32x32x64_f8f6f4 %" #acc ", %16, %17, 0, %18, %18 op_sel_hi:[0,0,0]\n\t"

// CASE: 0 VALU only (fma)   1 MFMA only (acc v)   2 MFMA acc v + fma   3 MFMA acc a + fma   4 MFMA acc a, A/B a + fma
//       5 VALU only (mix)   6 MFMA acc v + mix    7 MFMA acc a + mix   8 MFMA acc a, A/B a + mix   9 MFMA C=0 (acc v) + fma
//       10 MFMA only acc a, A/B a
template <int CASE>
__global__ void __launch_bounds__(256) k(float* out, Stamp* st, float seed) {
  extern __shared__ char pad[];
  float a[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) a[i] = seed * 1e-3f * (i + 1) + threadIdx.x * 1e-6f;
  float m1 = 0.999f + seed * 1e-9f, m2 = seed * 1e-7f;
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  i32x8 fa, fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[i] = 0x38303438 + threadIdx.x; fb[i] = 0x34383038 + i * 0x01000100; }
  int sc = 0x7F7F7F7F;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define OUTS(ACC) "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), \
                  "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(m1), "+v"(m2), "+" ACC(acc0), "+" ACC(acc1)
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
    if constexpr (CASE == 0) asm volatile(V32_FMA V32_FMA V32_FMA V32_FMA : OUTS("v") : "v"(fa), "v"(fb), "v"(sc));
    if constexpr (CASE == 1) asm volatile(MF(14) MF(15) MF(14) MF(15) : OUTS("v") : "v"(fa), "v"(fb), "v"(sc));
    if constexpr (CASE == 2) asm volatile(MF(14) V32_FMA MF(15) V32_FMA MF(14) V32_FMA MF(15) V32_FMA : OUTS("v") : "v"(fa), "v"(fb), "v"(sc));
    if constexpr (CASE == 3) asm volatile(MF(14) V32_FMA MF(15) V32_FMA MF(14) V32_FMA MF(15) V32_FMA : OUTS("a") : "v"(fa), "v"(fb), "v"(sc));
    if constexpr (CASE == 4) asm volatile(MF(14) V32_FMA MF(15) V32_FMA MF(14) V32_FMA MF(15) V32_FMA : OUTS("a") : "a"(fa), "a"(fb), "v"(sc));
    if constexpr (CASE == 5) asm volatile(V32_MIX V32_MIX V32_MIX V32_MIX : OUTS("v") : "v"(fa), "v"(fb), "v"(sc));
    if constexpr (CASE == 6) asm volatile(MF(14) V32_MIX MF(15) V32_MIX MF(14) V32_MIX MF(15) V32_MIX : OUTS("v") : "v"(fa), "v"(fb), "v"(sc));
    if constexpr (CASE == 7) asm volatile(MF(14) V32_MIX MF(15) V32_MIX MF(14) V32_MIX MF(15) V32_MIX : OUTS("a") : "v"(fa), "v"(fb), "v"(sc));
    if constexpr (CASE == 8) asm volatile(MF(14) V32_MIX MF(15) V32_MIX MF(14) V32_MIX MF(15) V32_MIX : OUTS("a") : "a"(fa), "a"(fb), "v"(sc));
    if constexpr (CASE == 9) asm volatile(MFZ(14) V32_FMA MFZ(15) V32_FMA MFZ(14) V32_FMA MFZ(15) V32_FMA : OUTS("v") : "v"(fa), "v"(fb), "v"(sc));
    if constexpr (CASE == 10) asm volatile(MF(14) MF(15) MF(14) MF(15) : OUTS("a") : "a"(fa), "a"(fb), "v"(sc));
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = m1 + m2;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  if (s == 12345.678f) out[threadIdx.x] = s + pad[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) { st->cyc = c1 - c0; st->rt = r1 - r0; }
}
template <int CASE>
static void run(const char* name, float* out, Stamp* st, int cus) {
  printf("%-46s", name);
  for (int w : {1, 2, 4}) {
    auto kern = k<CASE>;
    const int lds = w == 1 ? 100 * 1024 : w == 2 ? 60 * 1024 : 30 * 1024;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int rep = 0; rep < 2; ++rep) kern<<<cus * w, 256, lds>>>(out, st, 1.0f);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    kern<<<cus * w, 256, lds>>>(out, st, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    Stamp h; CHECK(hipMemcpy(&h, st, sizeof(h), hipMemcpyDeviceToHost));
    const double ghz = (double)h.cyc / ((double)h.rt * 10.0);  // s_memrealtime ticks at 100 MHz
    const double ns_trip = ms * 1e6 / ITER;
    // SIMD cycles per wave-trip: w waves share the SIMD for the launch's duration
    printf("  w=%d: %7.1f ns/trip %5.2f GHz %7.1f cyc/wave-trip", w, ns_trip, ghz, ns_trip * ghz / w);
  }
  printf("\n");
}
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float* out; Stamp* st;
  CHECK(hipMalloc(&out, 4096)); CHECK(hipMalloc(&st, sizeof(Stamp)));
  printf("%d CUs; trip = 4 MX-fp8 32x32x64 MFMAs (256 pipe cycles) and / or 128 vector instructions\n", cus);
  run<0>("128 v_fma", out, st, cus);
  run<5>("96 v_fma + 32 v_exp", out, st, cus);
  run<1>("4 MFMA (acc v)", out, st, cus);
  run<10>("4 MFMA (acc a, A/B a)", out, st, cus);
  run<2>("4 MFMA (acc v, A/B v) + 128 v_fma", out, st, cus);
  run<9>("4 MFMA (C = 0, D v)   + 128 v_fma", out, st, cus);
  run<3>("4 MFMA (acc a, A/B v) + 128 v_fma", out, st, cus);
  run<4>("4 MFMA (acc a, A/B a) + 128 v_fma", out, st, cus);
  run<6>("4 MFMA (acc v, A/B v) + 96 fma + 32 exp", out, st, cus);
  run<7>("4 MFMA (acc a, A/B v) + 96 fma + 32 exp", out, st, cus);
  run<8>("4 MFMA (acc a, A/B a) + 96 fma + 32 exp", out, st, cus);
  return 0;
}

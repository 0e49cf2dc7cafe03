#!/usr/bin/env python3
"""Generates tools/ubench_mix.hip: how should transcendental and plain vector instructions be ORDERED?  Every case
issues the same 96 v_fma + 32 v_exp per trip (independent, 12 registers), grouped differently; plus a role split
(even workgroups plain only, odd workgroups transcendental only) and the softmax mix with v_cvt_pk_fp8.
    python tools/gen_ubench_mix.py && hipcc -O3 --offload-arch=gfx950 tools/ubench_mix.hip -o build/ubench_mix"""
import os

def fma(i): return f"v_fma_f32 %{i%12}, %{i%12}, %12, %13"
def exp(i): return f"v_exp_f32 %{i%12}, %{i%12}"
def add(i): return f"v_add_f32 %{i%12}, %{i%12}, %12"
def cvt(i): return f"v_cvt_pk_fp8_f32 %{i%12}, %{(i+1)%12}, %{(i+2)%12}"

def grouped(nf, ne, reps):
    out, k = [], 0
    for _ in range(reps):
        for _ in range(nf): out.append(fma(k)); k += 1
        for _ in range(ne): out.append(exp(k)); k += 1
    return out

cases = [
    ("96 fma | 32 exp  (one block each)", grouped(96, 32, 1)),
    ("4 x (24 fma | 8 exp)", grouped(24, 8, 4)),
    ("8 x (12 fma | 4 exp)", grouped(12, 4, 8)),
    ("16 x (6 fma | 2 exp)", grouped(6, 2, 16)),
    ("32 x (3 fma | 1 exp)", grouped(3, 1, 32)),
    ("128 fma", [fma(i) for i in range(128)]),
    ("128 add (2 sources)", [add(i) for i in range(128)]),
    ("32 exp", [exp(i) for i in range(32)]),
    ("32 cvt_pk_fp8", [cvt(i) for i in range(32)]),
    ("32 exp | 16 cvt", [exp(i) for i in range(32)] + [cvt(i) for i in range(16)]),
    ("softmax grouped: 32 fma | 32 exp | 32 add | 16 cvt", [fma(i) for i in range(32)] + [exp(i) for i in range(32)] + [add(i) for i in range(32)] + [cvt(i) for i in range(16)]),
    ("softmax fine: 16 x (2 fma, 2 exp, 2 add, 1 cvt)", sum([[fma(2*i), fma(2*i+1), exp(2*i), exp(2*i+1), add(2*i), add(2*i+1), cvt(2*i)] for i in range(16)], [])),
    ("softmax mid: 4 x (8 fma | 8 exp | 8 add | 4 cvt)", sum([[fma(8*j+i) for i in range(8)] + [exp(8*j+i) for i in range(8)] + [add(8*j+i) for i in range(8)] + [cvt(4*j+i) for i in range(4)] for j in range(4)], [])),
]

def rate(tmpl, n=48):
    return [tmpl.format(d=f"%{i%12}", a=f"%{(i+1)%12}", b=f"%{(i+2)%12}", D=f"%{14 + i%5}", A=f"%{14 + (i+1)%5}", B=f"%{14 + (i+2)%5}") for i in range(n)]

RATES = [
    ("v_add_f32 d,d,a", "v_add_f32 {d}, {d}, {a}"),
    ("v_add_f32 d,s20,d", "v_add_f32 {d}, s20, {d}"),
    ("v_add_f32 d,1.0,d (inline const)", "v_add_f32 {d}, 1.0, {d}"),
    ("v_sub_f32 d,d,a", "v_sub_f32 {d}, {d}, {a}"),
    ("v_mul_f32 d,d,a", "v_mul_f32 {d}, {d}, {a}"),
    ("v_mul_f32 d,s20,d", "v_mul_f32 {d}, s20, {d}"),
    ("v_max_f32 d,d,a", "v_max_f32 {d}, {d}, {a}"),
    ("v_fma_f32 d,d,a,b (banks distinct)", "v_fma_f32 {d}, {d}, {a}, {b}"),
    ("v_fma_f32 d,d,a,-b (neg modifier)", "v_fma_f32 {d}, {d}, {a}, -{b}"),
    ("v_fma_f32 d,d,%12,%13 (two fixed regs)", "v_fma_f32 {d}, {d}, %12, %13"),
    ("v_fma_f32 d,d,%12,-%13", "v_fma_f32 {d}, {d}, %12, -%13"),
    ("v_fma_f32 d,d,s20,a", "v_fma_f32 {d}, {d}, s20, {a}"),
    ("v_fma_f32 d,d,s20,s20", "v_fma_f32 {d}, {d}, s20, s20"),
    ("v_fma_f32 d,d,2.0,a (inline const)", "v_fma_f32 {d}, {d}, 2.0, {a}"),
    ("v_fmac_f32 d,a,b", "v_fmac_f32 {d}, {a}, {b}"),
    ("v_fmac_f32 d,s20,a", "v_fmac_f32 {d}, s20, {a}"),
    ("v_pk_fma_f32 D,D,A,B", "v_pk_fma_f32 {D}, {D}, {A}, {B}"),
    ("v_pk_fma_f32 D,D,s[20:21],A", "v_pk_fma_f32 {D}, {D}, s[20:21], {A}"),
    ("v_pk_fma_f32 D,D,A,B neg_lo neg_hi", "v_pk_fma_f32 {D}, {D}, {A}, {B} neg_lo:[0,0,1] neg_hi:[0,0,1]"),
    ("v_pk_add_f32 D,D,A", "v_pk_add_f32 {D}, {D}, {A}"),
    ("v_pk_mul_f32 D,D,A", "v_pk_mul_f32 {D}, {D}, {A}"),
    ("v_max3_f32", "v_max3_f32 {d}, {d}, {a}, {b}"),
    ("v_cndmask_b32", "v_cndmask_b32 {d}, {a}, {b}, vcc"),
    ("v_cmp_le_u32", "v_cmp_le_u32 vcc, {a}, {b}"),
    ("v_add_u32", "v_add_u32 {d}, {d}, {a}"),
    ("v_lshl_add_u32", "v_lshl_add_u32 {d}, {d}, 2, {a}"),
    ("v_lshl_add_u64", "v_lshl_add_u64 {D}, {A}, 0, {D}"),
    ("v_and_b32", "v_and_b32 {d}, {d}, {a}"),
    ("v_or_b32", "v_or_b32 {d}, {d}, {a}"),
    ("v_lshlrev_b32", "v_lshlrev_b32 {d}, 3, {d}"),
    ("v_mov_b32", "v_mov_b32 {d}, {a}"),
    ("v_mov_b32 d, s20", "v_mov_b32 {d}, s20"),
    ("v_exp_f32", "v_exp_f32 {d}, {d}"),
    ("v_cvt_pk_fp8_f32", "v_cvt_pk_fp8_f32 {d}, {a}, {b}"),
    ("v_cvt_pk_fp8_f32 op_sel hi", "v_cvt_pk_fp8_f32 {d}, {a}, {b} op_sel:[0,0,1]"),
    ("v_cvt_pk_u8_f32 d,a,1,d", "v_cvt_pk_u8_f32 {d}, {a}, 1, {d}"),
    ("v_cvt_pk_u8_f32 d,a,b,d (vgpr byte select)", "v_cvt_pk_u8_f32 {d}, {a}, {b}, {d}"),
    ("v_cvt_u32_f32", "v_cvt_u32_f32 {d}, {a}"),
    ("v_cvt_i32_f32", "v_cvt_i32_f32 {d}, {a}"),
    ("v_cvt_f32_i32", "v_cvt_f32_i32 {d}, {a}"),
    ("v_cvt_pk_f16_f32 (pkrtz)", "v_cvt_pkrtz_f16_f32 {d}, {a}, {b}"),
    ("v_cvt_pk_bf16_f32", "v_cvt_pk_bf16_f32 {d}, {a}, {b}"),
    ("v_perm_b32", "v_perm_b32 {d}, {a}, {b}, %12"),
    ("v_ldexp_f32", "v_ldexp_f32 {d}, {a}, {b}"),
    ("v_med3_f32", "v_med3_f32 {d}, {d}, {a}, {b}"),
    ("v_max_f32 d,d,d", "v_max_f32 {d}, {d}, {d}"),
    ("v_cvt_scalef32_pk_fp8_f32", "v_cvt_scalef32_pk_fp8_f32 {d}, {a}, {b}, %12"),
    ("v_lshl_or_b32", "v_lshl_or_b32 {d}, {a}, 8, {d}"),
    ("v_add3_u32", "v_add3_u32 {d}, {d}, {a}, {b}"),
    ("v_floor_f32", "v_floor_f32 {d}, {a}"),
    ("v_fract_f32", "v_fract_f32 {d}, {a}"),
    ("v_rndne_f32", "v_rndne_f32 {d}, {a}"),
    ("s_nop 0", "s_nop 0"),
    ("s_mov_b32 (SALU)", "s_mov_b32 s21, s20"),
    ("v_readfirstlane", "v_readfirstlane_b32 s21, {a}"),
]
for nm, t in RATES:
    cases.append(("48 x " + nm, rate(t)))
cases.append(("softmax grouped, cvt via f16: 32 fma | 32 exp | 32 add | 16 cvt_pk_f16 | 16 scalef32_pk_fp8_f16",
              [fma(i) for i in range(32)] + [exp(i) for i in range(32)] + [add(i) for i in range(32)] +
              [f"v_cvt_pk_f16_f32 %{i%12}, %{(i+1)%12}, %{(i+2)%12}" for i in range(16)] +
              [f"v_cvt_scalef32_pk_fp8_f16 %{i%12}, %{(i+1)%12}, %12" for i in range(16)]))
ROLE = len(cases)  # role split case: parity of blockIdx picks 192 fma or 64 exp

bodies = []
for ci, (name, ins) in enumerate(cases):
    txt = "\\n\\t".join(ins)
    bodies.append(f'    if constexpr (CASE == {ci}) asm volatile("{txt}" : OUTS ::"s20", "s21", "a0", "vcc");')
f192 = "\\n\\t".join(fma(i) for i in range(192))
e64 = "\\n\\t".join(exp(i) for i in range(64))
bodies.append(f'    if constexpr (CASE == {ROLE}) {{ if (blockIdx.x & 1) asm volatile("{e64}" : OUTS :: "s20"); else asm volatile("{f192}" : OUTS :: "s20"); }}')
runs = "\n  ".join(f'run<{ci}>("{name}", out, st, cus);' for ci, (name, _) in enumerate(cases))
runs += f'\n  run<{ROLE}>("role split: even WGs 192 fma, odd WGs 64 exp (per pair = 2 x (96 fma + 32 exp))", out, st, cus);'

SRC = r'''// GENERATED by tools/gen_ubench_mix.py -- ordering of transcendental and plain vector instructions on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 4000;
struct Stamp { unsigned long long cyc, rt; };
template <int CASE>
__global__ void __launch_bounds__(256) k(float* out, Stamp* st, float seed) {
  extern __shared__ char pad[];
  float a[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) a[i] = seed * 1e-3f * (i + 1) + threadIdx.x * 1e-6f;
  float m1 = 0.999f + seed * 1e-9f, m2 = seed * 1e-7f;
  using f32x2 = __attribute__((ext_vector_type(2))) float;
  f32x2 pk[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) pk[i] = f32x2{seed * 1e-3f * i, seed * 2e-3f};
  asm volatile("s_mov_b32 s20, 0x3f7fbe77\n\ts_mov_b32 s21, 0x3f7fbe77" ::: "s20", "s21");
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define OUTS "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(m1), "+v"(m2), "+v"(pk[0]), "+v"(pk[1]), "+v"(pk[2]), "+v"(pk[3]), "+v"(pk[4])
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
@BODIES@
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = m1 + m2;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 5; ++i) s += pk[i][0] + pk[i][1];
  if (s == 12345.678f) out[threadIdx.x] = s + pad[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) { st->cyc = c1 - c0; st->rt = r1 - r0; }
}
template <int CASE>
static void run(const char* name, float* out, Stamp* st, int cus) {
  printf("%-72s", name);
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
    const double ghz = (double)h.cyc / ((double)h.rt * 10.0);
    const double ns_trip = ms * 1e6 / ITER;
    printf(" w=%d %7.1f ns %4.2f GHz %6.1f cyc/wt |", w, ns_trip, ghz, ns_trip * ghz / w);
  }
  printf("\n");
}
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float* out; Stamp* st;
  CHECK(hipMalloc(&out, 4096)); CHECK(hipMalloc(&st, sizeof(Stamp)));
  printf("%d CUs; cyc/wt = SIMD cycles per wave-trip (ns x clock / waves per SIMD)\n", cus);
  @RUNS@
  return 0;
}
'''
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench_mix.hip"), "w").write(
    SRC.replace("@BODIES@", "\n".join(bodies)).replace("@RUNS@", runs))

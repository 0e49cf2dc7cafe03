// What does v_cvt_pk_u8_f32 do with fractions, negatives, large values, infinities and NaN?  (the fp8 prefill's
// log-domain P codes rely on: round to nearest, saturation to [0, 255], NaN / -inf -> 0)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n) {
  const int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1, 0xAABBCCDDu);
}
int main() {
  const float vals[] = {0.f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 3.5f, 125.5f, 126.49f, 126.5f, 126.51f, 127.f, 254.6f, 255.4f, 256.f, 1000.f,
                        -0.4f, -0.6f, -3.f, -1e30f, -INFINITY, INFINITY, NAN, 7.9999f, 8.f, 55.75f};
  const int n = sizeof(vals) / sizeof(float);
  float* d_in; unsigned* d_out; unsigned h[64];
  hipMalloc(&d_in, sizeof(vals)); hipMalloc(&d_out, sizeof(unsigned) * n);
  hipMemcpy(d_in, vals, sizeof(vals), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d_in, d_out, n);
  hipMemcpy(h, d_out, sizeof(unsigned) * n, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%12g -> word %08x  byte1 = %u\n", vals[i], h[i], (h[i] >> 8) & 255);
  return 0;
}

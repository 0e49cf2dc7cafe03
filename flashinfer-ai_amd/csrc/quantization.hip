// Bit packing of boolean masks (the producer of the custom-mask operand of prefill).
// ref: PackBitsKernel / SegmentPackBitsKernel include/flashinfer/quantization.cuh:29-126, Python
// flashinfer/quantization.py:57-153.  Semantics: numpy.packbits over a flat 0/1 array, per segment for the
// segment form; a trailing partial byte is zero-padded.  Pure HBM work: each thread reads 8 bytes of x
// (one 64-bit load when aligned) and writes one byte of y.
#include <algorithm>

#include "common.h"

namespace fi {

constexpr int kPackThreads = 256;

__device__ __forceinline__ uint8_t pack8(const uint8_t* x, int64_t begin, int64_t end, bool little) {
  uint32_t byte = 0;
  if (begin + 8 <= end && ((uintptr_t)(x + begin) & 7) == 0) {
    const uint64_t w = *(const uint64_t*)(x + begin);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t bit = ((w >> (8 * i)) & 0xff) != 0;
      byte |= bit << (little ? i : 7 - i);
    }
  } else {
    for (int i = 0; i < 8 && begin + i < end; ++i) {
      const uint32_t bit = x[begin + i] != 0;
      byte |= bit << (little ? i : 7 - i);
    }
  }
  return (uint8_t)byte;
}

__global__ void __launch_bounds__(kPackThreads)
    packbits_kernel(const uint8_t* __restrict__ x, int64_t n, int little, uint8_t* __restrict__ y) {
  const int64_t nb = (n + 7) / 8;
  for (int64_t j = (int64_t)blockIdx.x * kPackThreads + threadIdx.x; j < nb;
       j += (int64_t)gridDim.x * kPackThreads)
    y[j] = pack8(x, 8 * j, n, little != 0);
}

// output byte j belongs to the segment s with out_indptr[s] <= j < out_indptr[s+1] (binary search)
__global__ void __launch_bounds__(kPackThreads)
    segment_packbits_kernel(const uint8_t* __restrict__ x, const int32_t* __restrict__ in_indptr,
                            const int32_t* __restrict__ out_indptr, int batch, int64_t y_bytes, int little,
                            uint8_t* __restrict__ y) {
  for (int64_t j = (int64_t)blockIdx.x * kPackThreads + threadIdx.x; j < y_bytes;
       j += (int64_t)gridDim.x * kPackThreads) {
    int lo = 0, hi = batch;  // invariant: out_indptr[lo] <= j < out_indptr[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)out_indptr[mid] <= j) lo = mid; else hi = mid;
    }
    const int64_t begin = (int64_t)in_indptr[lo] + 8 * (j - out_indptr[lo]);
    y[j] = pack8(x, begin, in_indptr[lo + 1], little != 0);
  }
}

}  // namespace fi

using namespace fi;

extern "C" FI_API int fi_packbits(const uint8_t* x, int64_t n, int32_t bitorder_little, uint8_t* y,
                                  fi_stream_t stream) {
  FI_REQUIRE(n >= 0, "packbits: negative length");
  if (n == 0) return 0;
  FI_REQUIRE(x && y, "packbits: null tensor");
  const int64_t nb = (n + 7) / 8;
  const int grid = (int)std::min<int64_t>((nb + kPackThreads - 1) / kPackThreads, 65535);
  packbits_kernel<<<dim3(grid), dim3(kPackThreads), 0, (hipStream_t)stream>>>(x, n, bitorder_little, y);
  FI_HIP_CALL(hipGetLastError());
  return 0;
}

extern "C" FI_API int fi_segment_packbits(const uint8_t* x, const int32_t* in_indptr,
                                          const int32_t* out_indptr, int32_t batch, int64_t y_bytes,
                                          int32_t bitorder_little, uint8_t* y, fi_stream_t stream) {
  FI_REQUIRE(batch >= 0 && y_bytes >= 0, "segment_packbits: negative size");
  if (batch == 0 || y_bytes == 0) return 0;
  FI_REQUIRE(x && y && in_indptr && out_indptr, "segment_packbits: null tensor");
  const int grid = (int)std::min<int64_t>((y_bytes + kPackThreads - 1) / kPackThreads, 65535);
  segment_packbits_kernel<<<dim3(grid), dim3(kPackThreads), 0, (hipStream_t)stream>>>(
      x, in_indptr, out_indptr, batch, y_bytes, bitorder_little, y);
  FI_HIP_CALL(hipGetLastError());
  return 0;
}

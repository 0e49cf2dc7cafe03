// Page-table kernels: ragged -> (batch index, position) expansion and the K/V append scatter.
// ref: AppendPagedKVCacheKernel include/flashinfer/page.cuh:258-284, launcher :345-411;
//      get_batch_indices_positions flashinfer/page.py:169-221 (Triton kernel triton/page.py:22-39).
// Both are pure HBM copies: 16 bytes per lane, consecutive lanes on consecutive chunks of a row.
#include <algorithm>

#include "common.h"

namespace fi {

constexpr int kPageThreads = 256;

__global__ void __launch_bounds__(kPageThreads)
    batch_indices_positions_kernel(const int32_t* __restrict__ append_indptr,
                                   const int32_t* __restrict__ seq_lens, int32_t* batch_indices,
                                   int32_t* positions, int nnz) {
  const int b = blockIdx.x;
  const int lo = append_indptr[b], hi = append_indptr[b + 1];
  const int seq_len = seq_lens[b];
  for (int i = lo + threadIdx.x; i < hi && i < nnz; i += kPageThreads) {
    batch_indices[i] = b;
    positions[i] = i + seq_len - hi;  // last appended token sits at seq_len - 1
  }
}

struct AppendParams {
  const void* key;
  const void* value;
  void* k_cache;
  void* v_cache;
  const int32_t* batch_indices;
  const int32_t* positions;
  const int32_t* kv_indptr;
  const int32_t* kv_indices;
  int64_t k_stride_n, k_stride_h, v_stride_n, v_stride_h;  // append tensors (elements)
  int64_t stride_page, stride_n, stride_h;                  // cache (elements)
  FastDiv page_div;
  int32_t page_size, num_heads, head_dim, nnz, esize;
};

// one 16-byte chunk per thread: item -> (token, head, chunk)
__global__ void __launch_bounds__(kPageThreads) append_paged_kv_cache_kernel(const AppendParams p) {
  const int cpr = p.head_dim * p.esize / 16;  // chunks per head row
  const int64_t total = (int64_t)p.nnz * p.num_heads * cpr;
  for (int64_t it = (int64_t)blockIdx.x * kPageThreads + threadIdx.x; it < total;
       it += (int64_t)gridDim.x * kPageThreads) {
    const int c = (int)(it % cpr);
    const int64_t r = it / cpr;
    const int h = (int)(r % p.num_heads);
    const int i = (int)(r / p.num_heads);
    const int b = p.batch_indices[i];
    const int pos = p.positions[i];
    // ref: page.cuh:272-275 -- page_iter = indptr[b] + pos / page_size, entry = pos % page_size
    const int pi = (int)fast_div((uint32_t)pos, p.page_div);
    const int entry = pos - pi * p.page_size;
    const int page = p.kv_indices[p.kv_indptr[b] + pi];
    const int64_t dst = ((int64_t)page * p.stride_page + (int64_t)h * p.stride_h +
                         (int64_t)entry * p.stride_n) * p.esize + c * 16;
    const int64_t ks = ((int64_t)i * p.k_stride_n + (int64_t)h * p.k_stride_h) * p.esize + c * 16;
    const int64_t vs = ((int64_t)i * p.v_stride_n + (int64_t)h * p.v_stride_h) * p.esize + c * 16;
    *(u32x4*)((char*)p.k_cache + dst) = *(const u32x4*)((const char*)p.key + ks);
    *(u32x4*)((char*)p.v_cache + dst) = *(const u32x4*)((const char*)p.value + vs);
  }
}

}  // namespace fi

using namespace fi;

extern "C" FI_API int fi_get_batch_indices_positions(const int32_t* append_indptr, const int32_t* seq_lens,
                                                     int32_t batch_size, int32_t nnz,
                                                     int32_t* batch_indices, int32_t* positions,
                                                     fi_stream_t stream) {
  if (batch_size == 0 || nnz == 0) return 0;
  FI_REQUIRE(append_indptr && seq_lens && batch_indices && positions,
             "get_batch_indices_positions: null tensor");
  batch_indices_positions_kernel<<<dim3(batch_size), dim3(kPageThreads), 0, (hipStream_t)stream>>>(
      append_indptr, seq_lens, batch_indices, positions, nnz);
  FI_HIP_CALL(hipGetLastError());
  return 0;
}

extern "C" FI_API int fi_append_paged_kv_cache(const void* append_key, const void* append_value,
                                               int64_t k_stride_n, int64_t k_stride_h,
                                               int64_t v_stride_n, int64_t v_stride_h,
                                               const int32_t* batch_indices, const int32_t* positions,
                                               int32_t nnz, const fi_paged_kv_t* kv, fi_stream_t stream) {
  if (nnz == 0) return 0;
  FI_REQUIRE(append_key && append_value && batch_indices && positions && kv,
             "append_paged_kv_cache: null argument");
  FI_REQUIRE(kv->k_data && kv->v_data && kv->indptr && kv->indices, "append_paged_kv_cache: null cache");
  const int esz = (int)dtype_size(kv->dtype);
  FI_REQUIRE(esz == 1 || esz == 2, "append_paged_kv_cache: unsupported cache dtype %d", kv->dtype);
  FI_REQUIRE((kv->head_dim * esz) % 16 == 0, "append_paged_kv_cache: head_dim rows must be 16-byte multiples");
  FI_REQUIRE((kv->stride_n * esz) % 16 == 0 && (kv->stride_h * esz) % 16 == 0 &&
                 (kv->stride_page * esz) % 16 == 0 && (k_stride_n * esz) % 16 == 0 &&
                 (k_stride_h * esz) % 16 == 0 && (v_stride_n * esz) % 16 == 0 &&
                 (v_stride_h * esz) % 16 == 0 && ((uintptr_t)append_key % 16) == 0 &&
                 ((uintptr_t)append_value % 16) == 0 && ((uintptr_t)kv->k_data % 16) == 0 &&
                 ((uintptr_t)kv->v_data % 16) == 0,
             "append_paged_kv_cache: rows must be 16-byte aligned");
  AppendParams p;
  p.key = append_key;
  p.value = append_value;
  p.k_cache = (void*)kv->k_data;
  p.v_cache = (void*)kv->v_data;
  p.batch_indices = batch_indices;
  p.positions = positions;
  p.kv_indptr = kv->indptr;
  p.kv_indices = kv->indices;
  p.k_stride_n = k_stride_n;
  p.k_stride_h = k_stride_h;
  p.v_stride_n = v_stride_n;
  p.v_stride_h = v_stride_h;
  p.stride_page = kv->stride_page;
  p.stride_n = kv->stride_n;
  p.stride_h = kv->stride_h;
  p.page_div = FastDiv((uint32_t)kv->page_size);
  p.page_size = kv->page_size;
  p.num_heads = kv->num_kv_heads;
  p.head_dim = kv->head_dim;
  p.nnz = nnz;
  p.esize = esz;
  const int64_t total = (int64_t)nnz * kv->num_kv_heads * (kv->head_dim * esz / 16);
  const int grid = (int)std::min<int64_t>((total + kPageThreads - 1) / kPageThreads, 256 * 8);
  append_paged_kv_cache_kernel<<<dim3(grid), dim3(kPageThreads), 0, (hipStream_t)stream>>>(p);
  FI_HIP_CALL(hipGetLastError());
  return 0;
}

// Batch / single decode: host planner, dispatcher and C-ABI entry points.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "decode_kernel.h"
#include "merge_kernel.h"

namespace fi {

// launchers defined by decode_inst.hip, one per (kv dtype, head_dim)
#define FI_DECL(DT, HD)                                                                   \
  hipError_t decode_launch_##DT##_##HD(const DecodeKernelParams& p, int gt, int rope, int grid, \
                                       hipStream_t stream);
FI_DECL(0, 64) FI_DECL(0, 128) FI_DECL(0, 256) FI_DECL(0, 512)
FI_DECL(1, 64) FI_DECL(1, 128) FI_DECL(1, 256) FI_DECL(1, 512)
FI_DECL(2, 64) FI_DECL(2, 128) FI_DECL(2, 256) FI_DECL(2, 512)
FI_DECL(3, 64) FI_DECL(3, 128) FI_DECL(3, 256) FI_DECL(3, 512)
#undef FI_DECL

typedef hipError_t (*decode_launch_fn)(const DecodeKernelParams&, int, int, int, hipStream_t);

static decode_launch_fn find_launcher(int kv_dt, int head_dim) {
#define FI_ROW(DT)                          \
  case DT:                                  \
    switch (head_dim) {                     \
      case 64:                              \
        return decode_launch_##DT##_64;     \
      case 128:                             \
        return decode_launch_##DT##_128;    \
      case 256:                             \
        return decode_launch_##DT##_256;    \
      case 512:                             \
        return decode_launch_##DT##_512;    \
      default:                              \
        return nullptr;                     \
    }
  switch (kv_dt) {
    FI_ROW(0)
    FI_ROW(1)
    FI_ROW(2)
    FI_ROW(3)
    default:
      return nullptr;
  }
#undef FI_ROW
}

// q-head tile: the wave keeps GT query heads of one kv head in registers.
static int pick_head_tile(int group_size, int kv_dt = FI_DTYPE_BF16) {
  static const int max_tile = [] {
    const char* e = getenv("FI_DECODE_MAX_HEAD_TILE");
    int v = e ? atoi(e) : 0;
    return (v == 1 || v == 2 || v == 4) ? v : 4;
  }();
  // groups larger than 4 are processed as several 4-head tiles by neighbouring waves of one workgroup
  // (they stream the same K/V rows, so HBM sees them once): measured 1.7-1.9x faster than an 8-head tile,
  // which is VALU-bound and spills (profiles/r01 notes in DESIGN.md).
  int t = group_size <= 1 ? 1 : group_size == 2 ? 2 : 4;
  (void)kv_dt;
  return t < max_tile ? t : max_tile;
}

// Matrix-core decode (decode_mfma_kernel.h): groups too wide for the VALU kernel, K/V stored in the q
// dtype.  FI_DECODE_MFMA_MIN_GROUP moves the crossover (0 disables the path).
hipError_t decode_mfma_launch(const DecodeKernelParams& p, int q_dtype, int kv_dtype, int head_dim, int rope, int grid,
                              hipStream_t stream);
hipError_t decode_mfma16_launch(const DecodeKernelParams& p, int q_dtype, int kv_dtype, int head_dim, int rope,
                                int grid, hipStream_t stream);
// the 16x16x32 form (decode_mfma16_kernel.h) serves every group of <= 16 heads the matrix-core path accepts:
// measured >= the VALU kernel (G <= 4) and >= the 32x32x16 form (G 5..16) on every shape of
// tools/bench_decode_kernels.py (C2 6.35 -> 6.52 TB/s, bs 8 x 1024 20.3 -> 16.1 us), and the only one with room
// for the fused-RoPE rotation.  FI_DECODE_MFMA16=0 restores the r1 choice.
static bool mfma16_enabled() {
  static const bool on = [] {
    const char* e = getenv("FI_DECODE_MFMA16");
    return e ? atoi(e) != 0 : true;
  }();
  return on;
}
static bool mfma16_decode(int group_size, bool rope) {
  (void)rope;
  return group_size <= 16 && mfma16_enabled();
}
static int tokens_per_load(int kv_dt, int head_dim);
static int ilog2_exact(int x);
static bool mfma_decode_shape(int group_size, int q_dt, int kv_dt, int head_dim, int page_size, bool rope = false) {
  static const int min_group = [] {
    const char* e = getenv("FI_DECODE_MFMA_MIN_GROUP");
    return e ? atoi(e) : 5;
  }();
  static const int min_group_fp8 = [] {
    // an fp8 cache doubles the VALU work per byte: measured 5.0 (VALU) vs 6.35 TB/s (MFMA) at G = 4,
    // 6.6 vs 6.4 TB/s at G = 1
    const char* e = getenv("FI_DECODE_MFMA_MIN_GROUP_FP8");
    return e ? atoi(e) : 3;
  }();
  const bool fp8 = kv_dt == FI_DTYPE_FP8_E4M3 || kv_dt == FI_DTYPE_FP8_E5M2;
  int mg = fp8 ? min_group_fp8 : min_group;
  // pages too small (or not a power of two) for the VALU kernel's scalar-page fast path: its per-lane
  // page-lookup fallback runs at 4.5 TB/s (page_size 1), the matrix-core kernel -- which always gathers per
  // lane -- at 6.3; token-granular page tables (page_size 1) are common
  if (mg > 0 && (ilog2_exact(page_size) < 0 || page_size < tokens_per_load(kv_dt, head_dim))) mg = 1;
  // fused RoPE: the K rotation is vector-ALU work that the VALU kernel has no room for (4.8 TB/s at G = 4);
  // on the matrix-core kernel it rides an idle pipe.  FI_DECODE_MFMA_ROPE=0 keeps the VALU kernel.
  static const bool rope_on_mfma = [] {
    const char* e = getenv("FI_DECODE_MFMA_ROPE");
    return e ? atoi(e) != 0 : true;
  }();
  if (rope && mg > 0) mg = rope_on_mfma ? 1 : 0x7fffffff;
  if (!rope && mg > 0 && group_size <= 16 && mfma16_enabled()) mg = 1;
  return mg > 0 && group_size >= mg && (q_dt == kv_dt || fp8) &&
         (q_dt == FI_DTYPE_F16 || q_dt == FI_DTYPE_BF16) && (head_dim == 64 || head_dim == 128);
}

static int tokens_per_load(int kv_dt, int head_dim) {
  const int vec = 16 / (int)dtype_size(kv_dt);
  return 64 / (head_dim / vec);
}

static int ilog2_exact(int x) {
  if (x <= 0 || (x & (x - 1))) return -1;
  int l = 0;
  while ((1 << l) < x) ++l;
  return l;
}

// work items per CU the planner cuts the batch into (FI_DECODE_WAVES_PER_CU overrides).  r3 sweep over the reference
// benchmark's grid (tools/bench_ref_grids.py, tools/r3/dsweep.py; bf16 32 / 4 and 32 / 8 heads, random and identity
// page order): with a 16-bit cache 4 per CU is level with 8 at C2 (6.72 against 6.65 TB/s) and ahead on everything
// smaller, where 8 cuts chunks of 256-512 tokens whose fixed cost shows (bs 256 x kv 1024: 6.46 against 5.32 -- no
// split at all; bs 64 x kv 4096, 32 / 4: 5.84 against 4.98; bs 32 x kv 4096: 5.05 against 4.37); an fp8 cache moves
// half the bytes per token and keeps 8 (C2 shape 6.08 against 5.59).  Counts that do not divide the chunking evenly
// (3, 5, 6) lose 5-20 %.  A single request (fi_single_decode) always wants many chunks: 8.
static int decode_waves_per_cu(int kv_dtype, bool batch) {
  if (const char* e = getenv("FI_DECODE_WAVES_PER_CU")) {
    int v = atoi(e);
    if (v > 0) return v;
  }
  const bool fp8 = kv_dtype == FI_DTYPE_FP8_E4M3 || kv_dtype == FI_DTYPE_FP8_E5M2;
  return (batch && !fp8) ? 4 : 8;
}

// ref: PartitionPagedKVCacheBinarySearchMinNumPagePerBatch, scheduler.cuh:73-99
static void partition_pages(uint32_t max_grid, uint32_t gdy, const std::vector<int32_t>& num_pages,
                            uint32_t min_pages, uint32_t* pages_per_chunk, uint32_t* new_batch) {
  uint32_t low = min_pages, high = 0;
  for (int32_t e : num_pages) high = std::max<uint32_t>(high, (uint32_t)e);
  while (low < high) {
    uint32_t mid = (low + high) / 2;
    uint64_t nb = 0;
    for (int32_t e : num_pages) nb += ceil_div<uint32_t>((uint32_t)e, mid);
    if (nb * gdy > max_grid)
      low = mid + 1;
    else
      high = mid;
  }
  uint32_t nb = 0;
  for (int32_t e : num_pages) nb += ceil_div<uint32_t>((uint32_t)std::max(e, 1), low);
  *pages_per_chunk = low;
  *new_batch = nb;
}

}  // namespace fi

using namespace fi;

extern "C" FI_API int fi_batch_decode_plan(void* float_ws, size_t float_ws_bytes, void* int_ws,
                                    void* pinned_int_ws, size_t int_ws_bytes,
                                    const int32_t* indptr_h, int32_t batch_size,
                                    int32_t num_qo_heads, int32_t num_kv_heads, int32_t page_size,
                                    int32_t enable_cuda_graph, int32_t head_dim, int32_t q_dtype,
                                    int32_t kv_dtype, int32_t max_grid_hint, int32_t window_left,
                                    int64_t* plan_info_out, fi_stream_t stream) {
  (void)float_ws;
  FI_REQUIRE(pinned_int_ws && indptr_h && plan_info_out, "batch_decode_plan: null argument");
  FI_REQUIRE(batch_size >= 0 && page_size > 0, "batch_decode_plan: bad batch_size/page_size");
  FI_REQUIRE(num_kv_heads > 0 && num_qo_heads % num_kv_heads == 0,
             "batch_decode_plan: num_qo_heads (%d) must be a multiple of num_kv_heads (%d)",
             num_qo_heads, num_kv_heads);
  FI_REQUIRE(q_dtype == FI_DTYPE_F16 || q_dtype == FI_DTYPE_BF16,
             "batch_decode_plan: q dtype must be f16/bf16");
  FI_REQUIRE(find_launcher(kv_dtype, head_dim) != nullptr,
             "batch_decode_plan: unsupported kv dtype %d / head_dim %d", kv_dtype, head_dim);
  for (int i = 0; i < batch_size; ++i)
    FI_REQUIRE(indptr_h[i + 1] >= indptr_h[i], "batch_decode_plan: indptr must be non-decreasing");

  const int group = num_qo_heads / num_kv_heads;
  const int gt = pick_head_tile(group, kv_dtype);
  // the matrix-core kernel covers the whole group with one wave; a run() that cannot use it (rope, alibi,
  // soft cap, window) still works on this plan, with head_tiles x the work items
  const int head_tiles = mfma_decode_shape(group, q_dtype, kv_dtype, head_dim, page_size) ? ceil_div(group, 32) : ceil_div(group, gt);
  const uint32_t gdy = (uint32_t)(num_kv_heads * head_tiles);
  // head tiles of one kv head stream the same rows (the partner wave's loads hit in L2), so a multi-tile
  // launch is sized for twice the waves: measured 4.42 vs 3.73 TB/s at Hq/Hkv = 64/8 (r1)
  const uint32_t max_grid =
      max_grid_hint > 0 ? (uint32_t)max_grid_hint
                        : (uint32_t)(fi_num_compute_units() * decode_waves_per_cu(kv_dtype, true) * (head_tiles > 1 ? 2 : 1));

  // pages a request's chunks are cut from: all of them, or with a sliding window the ones from the page
  // holding the earliest key the last token can see (kv_len >= (pages - 1) * page_size + 1)
  if (window_left < 0) window_left = -1;
  auto eff_pages = [&](int i) -> int32_t {
    const int32_t np = indptr_h[i + 1] - indptr_h[i];
    if (window_left < 0 || np <= 0) return np;
    const int64_t first = std::max<int64_t>((int64_t)(np - 1) * page_size - window_left, 0) / page_size;
    return (int32_t)(np - first);
  };
  // ---- work estimation (ref: scheduler.cuh:183-207) ----
  bool split_kv;
  uint32_t pages_per_chunk, new_batch;
  if ((uint64_t)batch_size * gdy >= max_grid) {
    split_kv = false;
    pages_per_chunk = 1;
    for (int i = 0; i < batch_size; ++i) pages_per_chunk = std::max<uint32_t>(pages_per_chunk, eff_pages(i));
    new_batch = batch_size;
  } else {
    std::vector<int32_t> num_pages(batch_size);
    for (int i = 0; i < batch_size; ++i) num_pages[i] = eff_pages(i);
    // chunks no shorter than one tile pair: >= 128 tokens (ref uses 128/page_size too)
    const uint32_t min_pages = std::max<uint32_t>(128u / (uint32_t)page_size, 1u);
    partition_pages(max_grid, gdy, num_pages, min_pages, &pages_per_chunk, &new_batch);
    split_kv = !(new_batch == (uint32_t)batch_size && !enable_cuda_graph);
  }
  const size_t padded = enable_cuda_graph ? (split_kv ? std::max<size_t>(max_grid / gdy, new_batch)
                                                      : (size_t)batch_size)
                                          : (size_t)new_batch;

  // ---- work list (ref: DecodeSplitKVIndptr, scheduler.cuh:348-364) ----
  OffsetAllocator ia(int_ws_bytes);
  const int64_t req_off = ia.alloc(padded * sizeof(int32_t));
  const int64_t tile_off = ia.alloc(padded * sizeof(int32_t));
  const int64_t oind_off = ia.alloc(((size_t)batch_size + 1) * sizeof(int32_t));
  const int64_t chunk_off = ia.alloc(sizeof(int32_t));
  int64_t mask_off = 0;
  if (split_kv) mask_off = ia.alloc(padded);
  FI_REQUIRE(ia.ok, "batch_decode_plan: int workspace too small (%zu bytes)", int_ws_bytes);

  char* hp = (char*)pinned_int_ws;
  int32_t* req_h = (int32_t*)(hp + req_off);
  int32_t* tile_h = (int32_t*)(hp + tile_off);
  int32_t* oind_h = (int32_t*)(hp + oind_off);
  memset(req_h, 0, padded * sizeof(int32_t));
  memset(tile_h, 0, padded * sizeof(int32_t));
  size_t w = 0;
  oind_h[0] = 0;
  for (int b = 0; b < batch_size; ++b) {
    const uint32_t np = (uint32_t)std::max(eff_pages(b), 1);
    const uint32_t nchunks = split_kv ? ceil_div(np, pages_per_chunk) : 1u;
    for (uint32_t t = 0; t < nchunks; ++t) {
      FI_REQUIRE(w < padded, "batch_decode_plan: work list overflow");
      req_h[w] = b;
      tile_h[w] = (int32_t)t;
      ++w;
    }
    oind_h[b + 1] = oind_h[b] + (int32_t)nchunks;
  }
  *(int32_t*)(hp + chunk_off) = (int32_t)(pages_per_chunk * (uint32_t)page_size);
  if (split_kv) {
    uint8_t* mask_h = (uint8_t*)(hp + mask_off);
    for (size_t i = 0; i < padded; ++i) mask_h[i] = i < w;
  }

  int64_t v_off = 0, s_off = 0;
  if (split_kv) {
    OffsetAllocator fa(float_ws_bytes);
    v_off = fa.alloc((size_t)num_qo_heads * padded * head_dim * sizeof(float));
    s_off = fa.alloc((size_t)num_qo_heads * padded * sizeof(float));
    FI_REQUIRE(fa.ok, "batch_decode_plan: float workspace too small (%zu bytes, need %zu)",
               float_ws_bytes, (size_t)num_qo_heads * padded * (head_dim + 1) * sizeof(float));
  }

  for (int i = 0; i < FI_DECODE_PLAN_INFO_LEN; ++i) plan_info_out[i] = 0;
  plan_info_out[FI_DP_PADDED_BATCH_SIZE] = (int64_t)padded;
  plan_info_out[FI_DP_V_OFFSET] = v_off;
  plan_info_out[FI_DP_S_OFFSET] = s_off;
  plan_info_out[FI_DP_REQUEST_INDICES_OFFSET] = req_off;
  plan_info_out[FI_DP_KV_TILE_INDICES_OFFSET] = tile_off;
  plan_info_out[FI_DP_O_INDPTR_OFFSET] = oind_off;
  plan_info_out[FI_DP_BLOCK_VALID_MASK_OFFSET] = mask_off;
  plan_info_out[FI_DP_KV_CHUNK_SIZE_PTR_OFFSET] = chunk_off;
  plan_info_out[FI_DP_ENABLE_CUDA_GRAPH] = enable_cuda_graph ? 1 : 0;
  plan_info_out[FI_DP_SPLIT_KV] = split_kv ? 1 : 0;
  plan_info_out[FI_DP_KV_CHUNK_SIZE] = (int64_t)pages_per_chunk * page_size;
  plan_info_out[FI_DP_NUM_WORK] = (int64_t)w;
  plan_info_out[FI_DP_BATCH_SIZE] = batch_size;
  plan_info_out[FI_DP_INT_BYTES_USED] = (int64_t)ia.used;
  plan_info_out[FI_DP_WINDOW_LEFT] = window_left;
  plan_info_out[FI_DP_MAGIC] = FI_DECODE_PLAN_MAGIC;

  if (int_ws && ia.used)
    FI_HIP_CALL(hipMemcpyAsync(int_ws, pinned_int_ws, ia.used, hipMemcpyHostToDevice,
                               (hipStream_t)stream));
  return 0;
}

namespace fi {

static int fill_common(DecodeKernelParams& kp, int kv_dt, int head_dim, int num_qo_heads,
                       int num_kv_heads, int page_size) {
  const int group = num_qo_heads / num_kv_heads;
  const int gt = pick_head_tile(group, kv_dt);
  kp.num_qo_heads = num_qo_heads;
  kp.num_kv_heads = num_kv_heads;
  kp.group_size = group;
  kp.head_tiles = ceil_div(group, gt);
  kp.page_size = page_size;
  kp.log2_page_size = ilog2_exact(page_size);
  kp.uniform_page = kp.log2_page_size >= 0 && page_size >= tokens_per_load(kv_dt, head_dim);
  kp.page_div = FastDiv((uint32_t)page_size);
  return gt;
}

}  // namespace fi

extern "C" FI_API int fi_batch_decode_run(void* float_ws, size_t float_ws_bytes, void* int_ws,
                                   size_t int_ws_bytes, const int64_t* plan_info,
                                   int32_t plan_info_len, const fi_batch_decode_params_t* a,
                                   fi_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FI_REQUIRE(plan_info && plan_info_len == FI_DECODE_PLAN_INFO_LEN &&
                 plan_info[FI_DP_MAGIC] == FI_DECODE_PLAN_MAGIC,
             "batch_decode_run: plan_info is not a decode plan (call plan() first)");
  FI_REQUIRE(a && a->q && a->o && a->kv.k_data && a->kv.v_data && a->kv.indptr &&
                 a->kv.last_page_len,
             "batch_decode_run: null tensor");
  FI_REQUIRE(int_ws, "batch_decode_run: null int workspace");
  const fi_paged_kv_t& kv = a->kv;
  FI_REQUIRE(a->q_dtype == FI_DTYPE_F16 || a->q_dtype == FI_DTYPE_BF16,
             "batch_decode_run: q dtype must be f16/bf16");
  FI_REQUIRE(kv.batch_size == plan_info[FI_DP_BATCH_SIZE],
             "batch_decode_run: batch size %d differs from the planned %lld", kv.batch_size,
             (long long)plan_info[FI_DP_BATCH_SIZE]);
  FI_REQUIRE(kv.num_kv_heads > 0 && a->num_qo_heads % kv.num_kv_heads == 0,
             "batch_decode_run: num_qo_heads must be a multiple of num_kv_heads");
  decode_launch_fn fn = find_launcher(kv.dtype, kv.head_dim);
  FI_REQUIRE(fn, "batch_decode_run: unsupported kv dtype %d / head_dim %d", kv.dtype, kv.head_dim);
  FI_REQUIRE(a->pos_encoding_mode >= 0 && a->pos_encoding_mode <= 2,
             "batch_decode_run: bad pos_encoding_mode %d", a->pos_encoding_mode);
  FI_REQUIRE(a->pos_encoding_mode != FI_POS_ALIBI || a->alibi_slopes,
             "batch_decode_run: ALIBI needs alibi_slopes");
  const size_t esz = dtype_size(kv.dtype);
  FI_REQUIRE(((uintptr_t)a->q % 16) == 0 && (a->q_stride_n * 2) % 16 == 0 && (a->q_stride_h * 2) % 16 == 0,
             "batch_decode_run: q rows must be 16-byte aligned");
  FI_REQUIRE((kv.stride_n * esz) % 16 == 0 && (kv.stride_h * esz) % 16 == 0 &&
                 (kv.stride_page * esz) % 16 == 0 && ((uintptr_t)kv.k_data % 16) == 0 &&
                 ((uintptr_t)kv.v_data % 16) == 0,
             "batch_decode_run: kv cache rows must be 16-byte aligned");
  if (kv.batch_size == 0) return 0;

  const bool split = plan_info[FI_DP_SPLIT_KV] != 0;
  const int64_t padded = plan_info[FI_DP_PADDED_BATCH_SIZE];
  DecodeKernelParams kp;
  memset(&kp, 0, sizeof(kp));
  const int gt = fill_common(kp, kv.dtype, kv.head_dim, a->num_qo_heads, kv.num_kv_heads,
                             kv.page_size);
  kp.q = a->q;
  kp.o = a->o;
  kp.lse = a->lse;
  kp.k = kv.k_data;
  kp.v = kv.v_data;
  kp.indptr = kv.indptr;
  kp.indices = kv.indices;
  kp.last_page_len = kv.last_page_len;
  const char* ib = (const char*)int_ws;
  kp.request_indices = (const int32_t*)(ib + plan_info[FI_DP_REQUEST_INDICES_OFFSET]);
  kp.kv_tile_indices = (const int32_t*)(ib + plan_info[FI_DP_KV_TILE_INDICES_OFFSET]);
  kp.o_indptr = (const int32_t*)(ib + plan_info[FI_DP_O_INDPTR_OFFSET]);
  kp.block_valid_mask =
      (split && plan_info[FI_DP_ENABLE_CUDA_GRAPH])
          ? (const uint8_t*)(ib + plan_info[FI_DP_BLOCK_VALID_MASK_OFFSET])
          : nullptr;
  (void)int_ws_bytes;
  kp.q_rope_offset = a->q_rope_offset;
  kp.kv_rope_pos_offset = kv.rope_pos_offset;
  kp.alibi_slopes = a->alibi_slopes;
  kp.q_stride_n = a->q_stride_n;
  kp.q_stride_h = a->q_stride_h;
  kp.kv_stride_page = kv.stride_page;
  kp.kv_stride_n = kv.stride_n;
  kp.kv_stride_h = kv.stride_h;
  const bool rope = a->pos_encoding_mode == FI_POS_ROPE_LLAMA;
  // ALiBi and the logits soft cap exist in the 16x16x32 kernel only
  const bool plain_logits = a->pos_encoding_mode != FI_POS_ALIBI && !(a->logits_soft_cap > 0.f);
  const bool use_mfma = mfma_decode_shape(kp.group_size, a->q_dtype, kv.dtype, kv.head_dim, kv.page_size, rope) &&
                        (plain_logits || mfma16_decode(kp.group_size, rope)) &&
                        kv.stride_page < (1ll << 31) && kv.stride_n < (1ll << 31);
  if (use_mfma) kp.head_tiles = ceil_div(kp.group_size, 32);
  kp.num_items = (int32_t)(padded * kv.num_kv_heads * kp.head_tiles);
  kp.kv_chunk_size = (int32_t)plan_info[FI_DP_KV_CHUNK_SIZE];
  // the kernels read the chunk size from the slot plan() refreshes (graph replay after a re-plan)
  kp.kv_chunk_size_ptr = (const int32_t*)(ib + plan_info[FI_DP_KV_CHUNK_SIZE_PTR_OFFSET]);
  kp.split_kv = split;
  kp.window_left = a->window_left;
  // chunks were cut from the window's pages at plan(): the kernel offsets them by the same first page
  kp.plan_window_left = (int32_t)plan_info[FI_DP_WINDOW_LEFT];
  FI_REQUIRE(kp.plan_window_left < 0 || kp.plan_window_left == a->window_left,
             "batch_decode_run: window_left %d differs from the planned %d", a->window_left, kp.plan_window_left);
  kp.q_dtype = a->q_dtype;
  kp.use_alibi = a->pos_encoding_mode == FI_POS_ALIBI;
  kp.logits_soft_cap = a->logits_soft_cap > 0.f ? a->logits_soft_cap : 0.f;
  kp.sm_scale = a->sm_scale;
  kp.rope_rcp_scale = a->rope_rcp_scale;
  kp.rope_rcp_theta = a->rope_rcp_theta;
  if (split) {
    FI_REQUIRE(float_ws, "batch_decode_run: split-kv plan needs the float workspace");
    const size_t need = (size_t)plan_info[FI_DP_S_OFFSET] +
                        (size_t)a->num_qo_heads * padded * sizeof(float);
    FI_REQUIRE(need <= float_ws_bytes, "batch_decode_run: float workspace too small");
    kp.tmp_o = (float*)((char*)float_ws + plan_info[FI_DP_V_OFFSET]);
    kp.tmp_lse = (float*)((char*)float_ws + plan_info[FI_DP_S_OFFSET]);
  }
  // fast path: scalar page ids, no logits transform (see decode_kernel.h)
  kp.fast_path = kp.uniform_page && kp.indices && !kp.use_alibi && kp.logits_soft_cap == 0.f &&
                 !getenv("FI_DECODE_FORCE_GENERIC");
  if (kp.num_items > 0) {
    const int grid = ceil_div(kp.num_items, kDecodeWaves);
    if (use_mfma && mfma16_decode(kp.group_size, rope))
      FI_HIP_CALL(decode_mfma16_launch(kp, a->q_dtype, kv.dtype, kv.head_dim, rope, grid, stream));
    else if (use_mfma)
      FI_HIP_CALL(decode_mfma_launch(kp, a->q_dtype, kv.dtype, kv.head_dim, rope, grid, stream));
    else
      FI_HIP_CALL(fn(kp, gt, a->pos_encoding_mode == FI_POS_ROPE_LLAMA, grid, stream));
  }
  if (split) {
    // ref: VariableLengthMergeStates after the partition-kv kernel, decode.cuh:798-821
    MergeNParams mp{kp.tmp_o, kp.tmp_lse, kp.o_indptr, a->o, a->lse, 0, kv.batch_size,
                    a->num_qo_heads, kv.head_dim, FI_DTYPE_F32, a->q_dtype};
    FI_HIP_CALL(launch_merge_n(mp, stream));
  }
  return 0;
}

extern "C" FI_API int fi_single_decode_run(const fi_single_decode_params_t* a, void* tmp, size_t tmp_bytes,
                                    fi_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FI_REQUIRE(a && a->q && a->k && a->v && a->o, "single_decode_run: null tensor");
  FI_REQUIRE(a->q_dtype == FI_DTYPE_F16 || a->q_dtype == FI_DTYPE_BF16,
             "single_decode_run: q dtype must be f16/bf16");
  FI_REQUIRE(a->num_kv_heads > 0 && a->num_qo_heads % a->num_kv_heads == 0,
             "single_decode_run: num_qo_heads must be a multiple of num_kv_heads");
  FI_REQUIRE(a->kv_len >= 0, "single_decode_run: negative kv_len");
  decode_launch_fn fn = find_launcher(a->kv_dtype, a->head_dim);
  FI_REQUIRE(fn, "single_decode_run: unsupported kv dtype %d / head_dim %d", a->kv_dtype,
             a->head_dim);
  FI_REQUIRE(a->pos_encoding_mode != FI_POS_ALIBI || a->alibi_slopes,
             "single_decode_run: ALIBI needs alibi_slopes");
  const size_t esz = dtype_size(a->kv_dtype);
  FI_REQUIRE(((uintptr_t)a->q % 16) == 0 && (a->q_stride_h * 2) % 16 == 0,
             "single_decode_run: q rows must be 16-byte aligned");
  FI_REQUIRE((a->kv_stride_n * esz) % 16 == 0 && (a->kv_stride_h * esz) % 16 == 0 &&
                 ((uintptr_t)a->k % 16) == 0 && ((uintptr_t)a->v % 16) == 0,
             "single_decode_run: k/v rows must be 16-byte aligned");

  // The dense tensor is addressed as an identity page table of 16-token pages.
  const int vpage = 16;
  DecodeKernelParams kp;
  memset(&kp, 0, sizeof(kp));
  const int gt = fill_common(kp, a->kv_dtype, a->head_dim, a->num_qo_heads, a->num_kv_heads, vpage);
  kp.q = a->q;
  kp.o = a->o;
  kp.lse = a->lse;
  kp.k = a->k;
  kp.v = a->v;
  kp.alibi_slopes = a->alibi_slopes;
  kp.q_stride_n = 0;
  kp.q_stride_h = a->q_stride_h;
  kp.kv_stride_page = (int64_t)vpage * a->kv_stride_n;
  kp.kv_stride_n = a->kv_stride_n;
  kp.kv_stride_h = a->kv_stride_h;
  kp.single_kv_len = a->kv_len;
  kp.window_left = a->window_left;
  kp.plan_window_left = -1;
  kp.q_dtype = a->q_dtype;
  kp.use_alibi = a->pos_encoding_mode == FI_POS_ALIBI;
  kp.logits_soft_cap = a->logits_soft_cap > 0.f ? a->logits_soft_cap : 0.f;
  kp.sm_scale = a->sm_scale;
  kp.rope_rcp_scale = a->rope_rcp_scale;
  kp.rope_rcp_theta = a->rope_rcp_theta;

  const bool rope = a->pos_encoding_mode == FI_POS_ROPE_LLAMA;
  const bool plain_logits = a->pos_encoding_mode != FI_POS_ALIBI && !(a->logits_soft_cap > 0.f);
  const bool use_mfma = mfma_decode_shape(kp.group_size, a->q_dtype, a->kv_dtype, a->head_dim, vpage, rope) &&
                        (plain_logits || mfma16_decode(kp.group_size, rope)) &&
                        kp.kv_stride_page < (1ll << 31);
  if (use_mfma) kp.head_tiles = ceil_div(kp.group_size, 32);
  // split-KV so that the chip is filled (ref: decode.cuh:689-733, kv_len > 256 -> chunks >= 256)
  const int gdy = a->num_kv_heads * kp.head_tiles;
  const int max_grid = fi_num_compute_units() * decode_waves_per_cu(a->kv_dtype, false);
  int chunk = a->kv_len, nchunks = 1;
  if (a->kv_len > 256 && tmp) {
    const int want = std::max(1, max_grid / gdy);
    chunk = std::max(ceil_div(a->kv_len, want), 256);
    chunk = ceil_div(chunk, vpage) * vpage;
    nchunks = ceil_div(a->kv_len, chunk);
    const size_t need = (size_t)nchunks * a->num_qo_heads * (a->head_dim + 1) * sizeof(float) + 64;
    if (need > tmp_bytes || nchunks <= 1) {
      chunk = a->kv_len;
      nchunks = 1;
    }
  }
  kp.split_kv = nchunks > 1;
  kp.kv_chunk_size = chunk;
  kp.num_items = nchunks * gdy;
  if (kp.split_kv) {
    kp.tmp_o = (float*)tmp;
    size_t vbytes = (size_t)nchunks * a->num_qo_heads * a->head_dim * sizeof(float);
    vbytes = (vbytes + 15) / 16 * 16;
    kp.tmp_lse = (float*)((char*)tmp + vbytes);
  }
  const int grid = ceil_div(kp.num_items, kDecodeWaves);
  if (use_mfma && mfma16_decode(kp.group_size, rope))
    FI_HIP_CALL(decode_mfma16_launch(kp, a->q_dtype, a->kv_dtype, a->head_dim, rope, grid, stream));
  else if (use_mfma)
    FI_HIP_CALL(decode_mfma_launch(kp, a->q_dtype, a->kv_dtype, a->head_dim, rope, grid, stream));
  else
    FI_HIP_CALL(fn(kp, gt, a->pos_encoding_mode == FI_POS_ROPE_LLAMA, grid, stream));
  if (kp.split_kv) {
    // partial states are [nchunks, Hq, D] == dense [row=1, n=nchunks, Hq, D]
    MergeNParams mp{kp.tmp_o, kp.tmp_lse, nullptr, a->o, a->lse, nchunks, 1,
                    a->num_qo_heads, a->head_dim, FI_DTYPE_F32, a->q_dtype};
    FI_HIP_CALL(launch_merge_n(mp, stream));
  }
  return 0;
}

// Batch / single prefill: host planner, dispatcher and C-ABI entry points.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "merge_kernel.h"
#include "prefill_kernel.h"

namespace fi {

typedef hipError_t (*prefill_launch_fn)(const PrefillKernelParams&, int, hipStream_t);

#define FI_PF_DECL(T, K, Q, D) \
  hipError_t prefill_launch_##T##_##K##_##Q##_##D(const PrefillKernelParams&, int, hipStream_t);
#define FI_PF_DECL_D(T, K, Q) FI_PF_DECL(T, K, Q, 64) FI_PF_DECL(T, K, Q, 128) FI_PF_DECL(T, K, Q, 256)
// 16-bit q: kv of the same type, or fp8 kv upcast on the fly (ref: prefill.cuh:637-647, 993-1004)
FI_PF_DECL_D(0, 0, 0) FI_PF_DECL_D(0, 2, 0) FI_PF_DECL_D(0, 3, 0)
FI_PF_DECL_D(1, 1, 1) FI_PF_DECL_D(1, 2, 1) FI_PF_DECL_D(1, 3, 1)
// fp8 q + fp8 kv (ref FA3 fp8 path), output type picks the compute type
FI_PF_DECL_D(0, 2, 2) FI_PF_DECL_D(1, 2, 2) FI_PF_DECL_D(0, 3, 3) FI_PF_DECL_D(1, 3, 3)
#undef FI_PF_DECL
#undef FI_PF_DECL_D

hipError_t prefill_fp8_launch(const PrefillKernelParams& p, int out_dtype, int e5m2, int head_dim, hipStream_t stream);

// fp8-native kernel (MX-scaled MFMA for both contractions): e4m3 or e5m2 q/k/v, head_dim 128, plain logits, no fused
// RoPE, no sliding window.  FI_PREFILL_FP8_NATIVE=0 forces the
// upcast-to-16-bit kernel (same arithmetic, kept as the cross-check).
static bool use_fp8_native(const PrefillKernelParams& kp, int q_dt, int kv_dt, int head_dim, int rope) {
  static const bool enabled = [] {
    const char* e = getenv("FI_PREFILL_FP8_NATIVE");
    return e ? atoi(e) != 0 : true;
  }();
  static const bool d256 = [] {  // FI_PREFILL_FP8_NATIVE_D256=0: head_dim 256 through the upcast kernel (cross-check)
    const char* e = getenv("FI_PREFILL_FP8_NATIVE_D256");
    return e ? atoi(e) != 0 : true;
  }();
  static const bool d64 = [] {
    const char* e = getenv("FI_PREFILL_FP8_NATIVE_D64");
    return e ? atoi(e) != 0 : true;
  }();
  return enabled && (q_dt == FI_DTYPE_FP8_E4M3 || q_dt == FI_DTYPE_FP8_E5M2) && kv_dt == q_dt &&
         (head_dim == 128 || (head_dim == 64 && d64 && kp.tile_q == kTileQ) ||
          (head_dim == 256 && d256 && kp.tile_q == kTileQ)) &&
         !rope && !kp.use_alibi && kp.logits_soft_cap == 0.f &&
         kp.window_left < 0 && !kp.custom_mask;
}

static prefill_launch_fn find_prefill(int t16, int kvs, int qs, int d) {
#define FI_TRY(T, K, Q)                                       \
  if (t16 == T && kvs == K && qs == Q) {                      \
    if (d == 64) return prefill_launch_##T##_##K##_##Q##_64;  \
    if (d == 128) return prefill_launch_##T##_##K##_##Q##_128; \
    if (d == 256) return prefill_launch_##T##_##K##_##Q##_256; \
    return nullptr;                                           \
  }
  FI_TRY(0, 0, 0) FI_TRY(0, 2, 0) FI_TRY(0, 3, 0)
  FI_TRY(1, 1, 1) FI_TRY(1, 2, 1) FI_TRY(1, 3, 1)
  FI_TRY(0, 2, 2) FI_TRY(1, 2, 2) FI_TRY(0, 3, 3) FI_TRY(1, 3, 3)
#undef FI_TRY
  return nullptr;
}

// compute type for (q dtype, o dtype)
static int compute_type(int q_dt, int o_dt) {
  if (q_dt == FI_DTYPE_F16 || q_dt == FI_DTYPE_BF16) return q_dt;
  return o_dt;  // fp8 q: f16 or bf16 output decides
}

}  // namespace fi

using namespace fi;

// Chunk choice by price (see fi_batch_prefill_plan): the reference chunk, its doublings and "whole" (returned as the
// kv span rounded up to a kv tile); rounds of workgroups per CU x tokens per item x 20 ns per 256 of head_dim_qk +
// head_dim_vo, for a split + 6 us + partial-state bytes at 3 TB/s; ties go to the coarser.
static int64_t price_kv_chunk(int64_t ref_chunk, int64_t max_kv_len, int n, const int64_t* q_tiles, const int64_t* kv_len,
                              const int64_t* qo_rows, int num_kv_heads, int num_qo_heads, int head_dim_qk,
                              int head_dim_vo) {
  const int64_t cus = fi_num_compute_units();
  const int64_t tok_ns = std::max<int64_t>(20 * (head_dim_qk + head_dim_vo) / 256, 1);
  auto cost_ns = [&](int64_t chunk, bool split) {
    int64_t items = 0, entries = 0;
    for (int b = 0; b < n; ++b) {
      const int64_t nc = split ? ceil_div<int64_t>(kv_len[b], chunk) : 1;
      items += q_tiles[b] * nc;
      entries += qo_rows[b] * nc;
    }
    int64_t t = ceil_div<int64_t>(items * num_kv_heads, cus) * std::min(chunk, max_kv_len) * tok_ns;
    if (split) t += 6000 + entries * num_qo_heads * head_dim_vo * 8 / 3000;
    return t;
  };
  const int64_t whole = ceil_div<int64_t>(max_kv_len, kTileKV) * kTileKV;
  int64_t best = whole, best_cost = cost_ns(whole, false);
  for (int64_t c = ref_chunk; c < max_kv_len; c *= 2) {
    const int64_t t = cost_ns(c, true);
    if (t < best_cost || (t == best_cost && c > best && best != whole)) {
      best = c;
      best_cost = t;
    }
  }
  return best;
}

extern "C" FI_API int fi_batch_prefill_plan_tile(
    void* float_ws, size_t float_ws_bytes, void* int_ws, void* pinned_int_ws, size_t int_ws_bytes,
    const int32_t* qo_indptr_h, const int32_t* kv_indptr_h, const int32_t* kv_len_arr_h,
    int32_t total_num_rows, int32_t batch_size, int32_t num_qo_heads, int32_t num_kv_heads,
    int32_t page_size, int32_t enable_cuda_graph, int32_t head_dim_qk, int32_t head_dim_vo,
    int32_t causal, int32_t window_left, int32_t fixed_split_size, int32_t disable_split_kv,
    int32_t cta_tile_q, int64_t* plan_info_out, fi_stream_t stream) {
  (void)float_ws; (void)kv_indptr_h;
  FI_REQUIRE(cta_tile_q == kTileQ || cta_tile_q == 2 * kTileQ,
             "batch_prefill_plan: cta_tile_q must be %d or %d (the fp8-native kernel's 8-wave form)", kTileQ,
             2 * kTileQ);
  const int64_t tile_q = cta_tile_q;
  FI_REQUIRE(pinned_int_ws && qo_indptr_h && kv_len_arr_h && plan_info_out,
             "batch_prefill_plan: null argument");
  FI_REQUIRE(batch_size >= 0 && page_size > 0, "batch_prefill_plan: bad batch size / page size");
  FI_REQUIRE(num_kv_heads > 0 && num_qo_heads % num_kv_heads == 0,
             "batch_prefill_plan: num_qo_heads (%d) must be a multiple of num_kv_heads (%d)",
             num_qo_heads, num_kv_heads);
  FI_REQUIRE(head_dim_qk == head_dim_vo, "batch_prefill_plan: head_dim_qk != head_dim_vo unsupported");
  FI_REQUIRE(head_dim_qk == 64 || head_dim_qk == 128 || head_dim_qk == 256,
             "batch_prefill_plan: unsupported head_dim %d (64/128/256)", head_dim_qk);
  FI_REQUIRE(qo_indptr_h[0] == 0, "batch_prefill_plan: qo_indptr[0] must be 0");
  const int group = num_qo_heads / num_kv_heads;

  // ---- q tiles and kv chunks (ref: PrefillSplitQOKVIndptr, scheduler.cuh:495-614, with
  // packed_qo_len = qo_len * G and a fixed 128-row q tile) ----
  std::vector<int64_t> q_tiles(batch_size), kv_len(batch_size);
  int64_t total_q_tiles = 0, max_kv_len = 1;
  for (int b = 0; b < batch_size; ++b) {
    const int64_t qo_len = qo_indptr_h[b + 1] - qo_indptr_h[b];
    FI_REQUIRE(qo_len >= 0, "batch_prefill_plan: qo_indptr must be non-decreasing");
    FI_REQUIRE(kv_len_arr_h[b] >= 0, "batch_prefill_plan: negative kv length");
    q_tiles[b] = ceil_div<int64_t>(qo_len * group, tile_q);
    kv_len[b] = std::max<int64_t>(kv_len_arr_h[b], 1);
    // sliding window: a q tile only walks the keys from its first row's window start on (the kernel
    // skips the rest), so chunks are cut from that span (ref: effective_kv_len_arr, scheduler.cuh:561-567)
    if (window_left >= 0)
      kv_len[b] = std::min<int64_t>(kv_len[b], (int64_t)window_left + (causal ? tile_q : qo_len) + kTileKV);
    total_q_tiles += q_tiles[b];
    max_kv_len = std::max(max_kv_len, kv_len[b]);
  }
  // resident workgroups (2 per CU) over the kv heads each item is launched for
  // (ref: max_batch_size_if_split = max_grid_size / num_kv_heads, scheduler.cuh:718)
  const int64_t max_items = std::max<int64_t>((int64_t)fi_num_compute_units() * (tile_q == kTileQ ? 2 : 1) / num_kv_heads, 1);
  const int64_t graph_bound =
      ceil_div<int64_t>((int64_t)total_num_rows * group, tile_q) + std::max(batch_size, 1) - 1;
  // chunk sizes are multiples of one 64-row kv tile and at least 128 tokens (ref: min_kv_chunk_size)
  auto items_at = [&](int64_t chunk) {
    int64_t n = 0;
    for (int b = 0; b < batch_size; ++b) n += q_tiles[b] * ceil_div<int64_t>(kv_len[b], chunk);
    return n;
  };
  const int64_t chunk_unit = kTileKV;
  int64_t kv_chunk = ceil_div<int64_t>(max_kv_len, chunk_unit) * chunk_unit;  // one chunk = no split
  bool split_kv = false;
  if (!disable_split_kv && batch_size > 0) {
    if (fixed_split_size > 0) {
      kv_chunk = ceil_div<int64_t>(fixed_split_size, chunk_unit) * chunk_unit;
    } else {
      // ref: PrefillBinarySearchKVChunkSize, scheduler.cuh:101-130 (in units of 64 tokens)
      int64_t low = 128 / chunk_unit, high = ceil_div<int64_t>(max_kv_len, chunk_unit);
      while (low < high) {
        const int64_t mid = (low + high) / 2;
        if (items_at(mid * chunk_unit) > max_items) low = mid + 1; else high = mid;
      }
      kv_chunk = std::max<int64_t>(low, 128 / chunk_unit) * chunk_unit;
      // The reference rule cuts as fine as max_items allows.  On this part that over-splits whenever the batch has
      // rows to merge: every chunk writes, and the merge reads back, an f32 partial row per query row and head (bs 1,
      // qo = kv = 1024: 82 us split in two against 31 us whole; bs 2, 512 x 4096: 132 against 95), while few-row
      // requests gain a lot (bs 1, 16 x 8192: 28 against 159 us).  So the candidates chunk, 2 x chunk, 4 x chunk ...
      // and "whole" are priced with a two-term model -- workgroup rounds per CU x tokens per item x 20 ns (per 256 of
      // head_dim_qk + head_dim_vo), plus for a split 6 us + partial-state bytes at 3 TB/s -- and the cheapest wins
      // (ties: the coarser).  Graph plans keep the reference rule (they always split).
      if (!enable_cuda_graph) {
        std::vector<int64_t> rows(batch_size);
        for (int b = 0; b < batch_size; ++b) rows[b] = qo_indptr_h[b + 1] - qo_indptr_h[b];
        kv_chunk = price_kv_chunk(kv_chunk, max_kv_len, batch_size, q_tiles.data(), kv_len.data(), rows.data(),
                                  num_kv_heads, num_qo_heads, head_dim_qk, head_dim_vo);
      }
    }
    // Load balance (not in the reference, whose rule above only ever splits a batch of fewer than max_items items): a
    // mixed batch -- many short requests and a few long ones with few query rows -- otherwise ends in a tail of
    // single workgroups walking the long requests (reference benchmark bench_batch_attention.py, 122 x (600, 1) + 8 x
    // (10000, 17): 0.23 ms for 0.3 GB).  With W = sum of q tiles x kv length, the ideal makespan is W / max_items;
    // chunks of at most half of that (and >= 256 tokens) let the longest-first work list even out.  A batch whose long
    // requests also have many query rows has a large W and keeps its single chunk; graph plans keep the reference
    // rule (their item count must stay under the captured bound).
    // Only where the reference rule left every request whole AND the batch is uneven (longest request >= twice the
    // mean item): an even batch gains nothing from more items (decode-only 128 x 8192 through this wrapper: -6 %), and
    // a batch the reference rule already cut is compute-bound prefill (4 x (4096, 128): -12 % when cut finer).
    if (fixed_split_size <= 0 && !enable_cuda_graph && kv_chunk >= max_kv_len && total_q_tiles > 0) {
      int64_t work = 0;
      for (int b = 0; b < batch_size; ++b) work += q_tiles[b] * kv_len[b];
      const int64_t bal = ceil_div<int64_t>(std::max<int64_t>(work / (2 * max_items), 256), chunk_unit) * chunk_unit;
      if (max_kv_len * total_q_tiles >= 2 * work && 2 * bal <= max_kv_len && items_at(bal) <= 8 * max_items)
        kv_chunk = bal;
    }
    // the partial states must fit the caller's float workspace: grow the chunks until they do (a plan
    // that cannot split at all is still correct, only less parallel)
    if (fixed_split_size <= 0) {
      auto ws_need = [&](int64_t chunk) {
        int64_t entries = 0;
        for (int b = 0; b < batch_size; ++b)
          entries += (int64_t)(qo_indptr_h[b + 1] - qo_indptr_h[b]) * ceil_div<int64_t>(kv_len[b], chunk);
        int64_t lse_entries = entries;
        if (enable_cuda_graph)  // the fixed lse region of a graph plan (below)
          lse_entries = std::max(entries, std::max(items_at(chunk), std::max(max_items, graph_bound)) *
                                              (ceil_div<int64_t>(tile_q, group) + 1));
        return (entries * num_qo_heads * head_dim_vo + lse_entries * num_qo_heads + 64) * (int64_t)sizeof(float);
      };
      while (kv_chunk < max_kv_len && ws_need(kv_chunk) > (int64_t)float_ws_bytes) kv_chunk *= 2;
    }
    split_kv = kv_chunk < max_kv_len;
    // a fixed-shape (graph) launch always takes the split path so that the kernel sequence does not
    // depend on the page table (ref: scheduler.cuh:129)
    if (enable_cuda_graph) split_kv = true;
  }
  FI_REQUIRE(kv_chunk < (1ll << 31), "batch_prefill_plan: kv chunk too large");

  // work list, costliest first (requests by rows per q tile x kv_len descending; for causal masks the later = heavier q
  // tiles first) so the tail of the launch is made of the cheapest items (ref LPT idea: scheduler.cuh:900-946)
  std::vector<int> order(batch_size);
  std::iota(order.begin(), order.end(), 0);
  // (cost of a request's items ~ rows of a q tile x kv length: in a mixed batch the compute-bound full tiles of a
  // prefill request go out before the memory-bound one-row items of equally long decode requests and run beside
  // them, instead of forming the tail -- bench_batch_attention.py's 254 x (8192, 1) + (8192, 4096))
  auto item_cost = [&](int b) {
    const int64_t rows = std::min<int64_t>((int64_t)(qo_indptr_h[b + 1] - qo_indptr_h[b]) * group, tile_q);
    return (int64_t)kv_len_arr_h[b] * std::max<int64_t>(rows, 1);
  };
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return item_cost(a) > item_cost(b); });
  std::vector<int32_t> req, tile, kvt;
  for (int b : order) {
    const int64_t ntiles = q_tiles[b];
    const int64_t nchunks = split_kv ? ceil_div<int64_t>(kv_len[b], kv_chunk) : 1;
    for (int64_t t = 0; t < ntiles; ++t)
      for (int64_t c = 0; c < nchunks; ++c) {
        req.push_back(b);
        tile.push_back((int32_t)(causal ? ntiles - 1 - t : t));
        kvt.push_back((int32_t)c);
      }
  }
  // Mixed batches: a q tile with many rows is compute-bound, one with a few rows (decode-like) streams its keys at
  // HBM rate.  Listed one kind after the other they run as two phases; interleaved in proportion (each kind keeps its
  // costliest-first order) the two kinds share the CUs and overlap.  Pure batches (one kind only) are unchanged.
  {
    std::vector<size_t> wide, narrow;
    for (size_t i = 0; i < req.size(); ++i) {
      const int b = req[i];
      const int64_t rows_left = (int64_t)(qo_indptr_h[b + 1] - qo_indptr_h[b]) * group - (int64_t)tile[i] * tile_q;
      (std::min<int64_t>(rows_left, tile_q) * 2 >= tile_q ? wide : narrow).push_back(i);
    }
    if (!wide.empty() && !narrow.empty()) {
      std::vector<int32_t> r2, t2, k2;
      const size_t n = req.size();
      size_t iw = 0, in = 0;
      for (size_t k = 0; k < n; ++k) {
        // wide items are due when their share of the first k + 1 slots falls behind
        const bool take_wide = in >= narrow.size() || (iw < wide.size() && iw * n <= k * wide.size());
        const size_t src = take_wide ? wide[iw++] : narrow[in++];
        r2.push_back(req[src]);
        t2.push_back(tile[src]);
        k2.push_back(kvt[src]);
      }
      req.swap(r2);
      tile.swap(t2);
      kvt.swap(k2);
    }
  }
  size_t padded = req.size();
  if (enable_cuda_graph) padded = std::max<size_t>(padded, (size_t)std::max(max_items, graph_bound));
  // partial-state ranges per qo row (ref merge_indptr, scheduler.cuh:597-600)
  const int64_t nrows_tab = split_kv ? (int64_t)std::max(total_num_rows, qo_indptr_h[batch_size]) : 0;
  OffsetAllocator ia(int_ws_bytes);
  const int64_t req_off = ia.alloc(std::max<size_t>(padded, 1) * sizeof(int32_t));
  const int64_t tile_off = ia.alloc(std::max<size_t>(padded, 1) * sizeof(int32_t));
  const int64_t kvt_off = ia.alloc(std::max<size_t>(padded, 1) * sizeof(int32_t));
  const int64_t mrg_off = ia.alloc((size_t)(nrows_tab + 1) * sizeof(int32_t));
  const int64_t chunk_off = ia.alloc(sizeof(int32_t));
  FI_REQUIRE(ia.ok, "batch_prefill_plan: int workspace too small (%zu bytes)", int_ws_bytes);
  int32_t* req_h = (int32_t*)((char*)pinned_int_ws + req_off);
  int32_t* tile_h = (int32_t*)((char*)pinned_int_ws + tile_off);
  int32_t* kvt_h = (int32_t*)((char*)pinned_int_ws + kvt_off);
  int32_t* mrg_h = (int32_t*)((char*)pinned_int_ws + mrg_off);
  for (size_t i = 0; i < padded; ++i) {
    req_h[i] = i < req.size() ? req[i] : -1;  // -1: padding item, the workgroup exits
    tile_h[i] = i < tile.size() ? tile[i] : 0;
    kvt_h[i] = i < kvt.size() ? kvt[i] : 0;
  }
  *(int32_t*)((char*)pinned_int_ws + chunk_off) = (int32_t)kv_chunk;
  int64_t entries = 0;
  mrg_h[0] = 0;
  if (split_kv) {
    int64_t row = 0;
    for (int b = 0; b < batch_size; ++b) {
      const int64_t nchunks = ceil_div<int64_t>(kv_len[b], kv_chunk);
      for (int64_t r = qo_indptr_h[b]; r < qo_indptr_h[b + 1]; ++r) {
        entries += nchunks;
        mrg_h[++row] = (int32_t)entries;
      }
    }
    FI_REQUIRE(entries < (1ll << 31), "batch_prefill_plan: too many partial states");
    for (; row < nrows_tab; ) mrg_h[++row] = (int32_t)entries;  // padded rows of a graph launch: empty
  }
  int64_t v_off = 0, s_off = 0;
  if (split_kv) {
    // lse region first, sized for the most partial states a launch of `padded` items can write (each
    // (row, chunk) pair belongs to one item of <= tile_q / G + 1 rows): with a fixed-shape (graph) plan
    // both offsets are then the same for every plan, so a captured run() stays valid after a re-plan.
    // The outputs follow and may use the rest of the workspace.
    OffsetAllocator fa(float_ws_bytes);
    const int64_t rows_per_item = ceil_div<int64_t>(tile_q, group) + 1;
    const int64_t lse_entries =
        enable_cuda_graph ? std::max<int64_t>(entries, (int64_t)padded * rows_per_item) : std::max<int64_t>(entries, 1);
    s_off = fa.alloc((size_t)lse_entries * num_qo_heads * sizeof(float));
    v_off = fa.alloc((size_t)std::max<int64_t>(entries, 1) * num_qo_heads * head_dim_vo * sizeof(float));
    FI_REQUIRE(fa.ok, "batch_prefill_plan: float workspace too small (%zu bytes, need %zu for %lld partial "
               "states)", float_ws_bytes,
               (size_t)entries * num_qo_heads * (head_dim_vo + 1) * sizeof(float), (long long)entries);
  }
  for (int i = 0; i < FI_PREFILL_PLAN_INFO_LEN; ++i) plan_info_out[i] = 0;
  plan_info_out[FI_PP_PADDED_BATCH_SIZE] = (int64_t)padded;
  plan_info_out[FI_PP_TOTAL_NUM_ROWS] = split_kv ? nrows_tab : total_num_rows;
  plan_info_out[FI_PP_KV_CHUNK_SIZE_PTR_OFFSET] = chunk_off;
  plan_info_out[FI_PP_CTA_TILE_Q] = tile_q;
  plan_info_out[FI_PP_REQUEST_INDICES_OFFSET] = req_off;
  plan_info_out[FI_PP_QO_TILE_INDICES_OFFSET] = tile_off;
  plan_info_out[FI_PP_KV_TILE_INDICES_OFFSET] = kvt_off;
  plan_info_out[FI_PP_MERGE_INDPTR_OFFSET] = mrg_off;
  plan_info_out[FI_PP_BATCH_SIZE] = batch_size;
  plan_info_out[FI_PP_KV_CHUNK_SIZE] = kv_chunk;
  plan_info_out[FI_PP_V_OFFSET] = v_off;
  plan_info_out[FI_PP_S_OFFSET] = s_off;
  plan_info_out[FI_PP_NUM_WORK] = (int64_t)req.size();
  plan_info_out[FI_PP_ENABLE_CUDA_GRAPH] = enable_cuda_graph ? 1 : 0;
  plan_info_out[FI_PP_SPLIT_KV] = split_kv ? 1 : 0;
  plan_info_out[FI_PP_MAGIC] = FI_PREFILL_PLAN_MAGIC;
  if (int_ws && ia.used)
    FI_HIP_CALL(hipMemcpyAsync(int_ws, pinned_int_ws, ia.used, hipMemcpyHostToDevice,
                               (hipStream_t)stream));
  return 0;
}


extern "C" FI_API int fi_batch_prefill_plan(
    void* float_ws, size_t float_ws_bytes, void* int_ws, void* pinned_int_ws, size_t int_ws_bytes,
    const int32_t* qo_indptr_h, const int32_t* kv_indptr_h, const int32_t* kv_len_arr_h,
    int32_t total_num_rows, int32_t batch_size, int32_t num_qo_heads, int32_t num_kv_heads,
    int32_t page_size, int32_t enable_cuda_graph, int32_t head_dim_qk, int32_t head_dim_vo,
    int32_t causal, int32_t window_left, int32_t fixed_split_size, int32_t disable_split_kv,
    int64_t* plan_info_out, fi_stream_t stream) {
  return fi_batch_prefill_plan_tile(float_ws, float_ws_bytes, int_ws, pinned_int_ws, int_ws_bytes, qo_indptr_h,
                                    kv_indptr_h, kv_len_arr_h, total_num_rows, batch_size, num_qo_heads,
                                    num_kv_heads, page_size, enable_cuda_graph, head_dim_qk, head_dim_vo, causal,
                                    window_left, fixed_split_size, disable_split_kv, kTileQ, plan_info_out, stream);
}

namespace fi {
static int check_prefill_dtypes(const char* who, int q_dt, int kv_dt, int o_dt) {
  FI_REQUIRE(o_dt == FI_DTYPE_F16 || o_dt == FI_DTYPE_BF16, "%s: output dtype must be f16/bf16", who);
  if (q_dt == FI_DTYPE_F16 || q_dt == FI_DTYPE_BF16) {
    FI_REQUIRE(o_dt == q_dt, "%s: output dtype must equal the 16-bit q dtype", who);
    FI_REQUIRE(kv_dt == q_dt || kv_dt == FI_DTYPE_FP8_E4M3 || kv_dt == FI_DTYPE_FP8_E5M2,
               "%s: kv dtype must equal q dtype or be fp8", who);
  } else {
    FI_REQUIRE((q_dt == FI_DTYPE_FP8_E4M3 || q_dt == FI_DTYPE_FP8_E5M2) && kv_dt == q_dt,
               "%s: fp8 attention needs q, k and v of one fp8 type (e4m3 or e5m2)", who);
  }
  return 0;
}
}  // namespace fi

extern "C" FI_API int fi_batch_prefill_paged_run(void* float_ws, size_t float_ws_bytes, void* int_ws,
                                                 size_t int_ws_bytes, const int64_t* plan_info,
                                                 int32_t plan_info_len,
                                                 const fi_batch_prefill_params_t* a,
                                                 fi_stream_t stream_) {
  (void)int_ws_bytes;
  hipStream_t stream = (hipStream_t)stream_;
  FI_REQUIRE(plan_info && plan_info_len == FI_PREFILL_PLAN_INFO_LEN &&
                 plan_info[FI_PP_MAGIC] == FI_PREFILL_PLAN_MAGIC,
             "batch_prefill_paged_run: plan_info is not a prefill plan (call plan() first)");
  FI_REQUIRE(a && int_ws, "batch_prefill_paged_run: null argument");
  const fi_paged_kv_t& kv = a->kv;
  const int64_t num_work = plan_info[FI_PP_PADDED_BATCH_SIZE];
  if (num_work == 0 || kv.batch_size == 0) return 0;
  // kv.indices == NULL && kv.last_page_len == NULL: ragged KV (identity page table, every page full)
  FI_REQUIRE(a->q && a->o && a->qo_indptr && kv.k_data && kv.v_data && kv.indptr,
             "batch_prefill_paged_run: null tensor");
  FI_REQUIRE(kv.last_page_len || !kv.indices, "batch_prefill_paged_run: a page table needs last_page_len");
  FI_REQUIRE(kv.batch_size == plan_info[FI_PP_BATCH_SIZE], "batch_prefill_paged_run: batch size differs from the plan");
  FI_REQUIRE(kv.num_kv_heads > 0 && a->num_qo_heads % kv.num_kv_heads == 0,
             "batch_prefill_paged_run: num_qo_heads must be a multiple of num_kv_heads");
  FI_REQUIRE(a->mask_mode >= FI_MASK_NON_CAUSAL && a->mask_mode <= FI_MASK_MULTIITEMSCORING,
             "batch_prefill_paged_run: bad mask_mode %d", a->mask_mode);
  FI_REQUIRE(a->mask_mode != FI_MASK_CUSTOM || (a->custom_mask && a->mask_indptr),
             "batch_prefill_paged_run: mask_mode CUSTOM needs custom_mask and mask_indptr");
  FI_REQUIRE(a->mask_mode != FI_MASK_MULTIITEMSCORING ||
                 (a->prefix_len_ptr && a->token_pos_in_items_ptr && a->token_pos_in_items_len > 0),
             "batch_prefill_paged_run: mask_mode MULTIITEMSCORING needs prefix_len_ptr, token_pos_in_items_ptr and "
             "token_pos_in_items_len");
  FI_REQUIRE(a->mask_mode != FI_MASK_MULTIITEMSCORING ||
                 (a->q_dtype == FI_DTYPE_F16 || a->q_dtype == FI_DTYPE_BF16),
             "batch_prefill_paged_run: multi-item scoring needs 16-bit queries");
  if (check_prefill_dtypes("batch_prefill_paged_run", a->q_dtype, kv.dtype, a->o_dtype)) return 1;
  const int t16 = compute_type(a->q_dtype, a->o_dtype);
  prefill_launch_fn fn = find_prefill(t16, kv.dtype, a->q_dtype, kv.head_dim);
  FI_REQUIRE(fn, "batch_prefill_paged_run: unsupported q/kv dtype %d/%d or head_dim %d", a->q_dtype,
             kv.dtype, kv.head_dim);
  FI_REQUIRE(a->pos_encoding_mode != FI_POS_ALIBI || a->alibi_slopes,
             "batch_prefill_paged_run: ALIBI needs alibi_slopes");
  const size_t qsz = dtype_size(a->q_dtype), ksz = dtype_size(kv.dtype);
  FI_REQUIRE(((uintptr_t)a->q % 16) == 0 && (a->q_stride_n * qsz) % 8 == 0 &&
                 (a->q_stride_h * qsz) % 8 == 0 && (qsz == 1 || ((a->q_stride_n * qsz) % 16 == 0 && (a->q_stride_h * qsz) % 16 == 0)),
             "batch_prefill_paged_run: q rows must be 16-byte (8-byte for fp8) aligned");
  FI_REQUIRE((kv.stride_n * ksz) % (8 * ksz) == 0 && (kv.stride_h * ksz) % (8 * ksz) == 0 &&
                 (kv.stride_page * ksz) % (8 * ksz) == 0 && ((uintptr_t)kv.k_data % 16) == 0 &&
                 ((uintptr_t)kv.v_data % 16) == 0,
             "batch_prefill_paged_run: kv cache rows must be aligned to 8 elements");

  FI_REQUIRE(kv.stride_page < (1ll << 31) && kv.stride_n < (1ll << 31) && kv.stride_page >= 0 && kv.stride_n >= 0,
             "batch_prefill_paged_run: kv page / token strides must be below 2^31 elements");
  PrefillKernelParams kp;
  memset(&kp, 0, sizeof(kp));
  kp.q = a->q;
  kp.o = a->o;
  kp.lse = a->lse;
  kp.k = kv.k_data;
  kp.v = kv.v_data;
  kp.qo_indptr = a->qo_indptr;
  kp.kv_indptr = kv.indptr;
  kp.kv_indices = kv.indices;
  kp.kv_last_page_len = kv.last_page_len;
  kp.request_indices = (const int32_t*)((const char*)int_ws + plan_info[FI_PP_REQUEST_INDICES_OFFSET]);
  kp.qo_tile_indices = (const int32_t*)((const char*)int_ws + plan_info[FI_PP_QO_TILE_INDICES_OFFSET]);
  const bool split = plan_info[FI_PP_SPLIT_KV] != 0;
  if (split) {
    FI_REQUIRE(float_ws, "batch_prefill_paged_run: a split-kv plan needs the float workspace");
    FI_REQUIRE((size_t)plan_info[FI_PP_V_OFFSET] <= float_ws_bytes,
               "batch_prefill_paged_run: float workspace smaller than at plan()");
    kp.kv_tile_indices = (const int32_t*)((const char*)int_ws + plan_info[FI_PP_KV_TILE_INDICES_OFFSET]);
    kp.merge_indptr = (const int32_t*)((const char*)int_ws + plan_info[FI_PP_MERGE_INDPTR_OFFSET]);
    kp.tmp_o = (float*)((char*)float_ws + plan_info[FI_PP_V_OFFSET]);
    kp.tmp_lse = (float*)((char*)float_ws + plan_info[FI_PP_S_OFFSET]);
    kp.kv_chunk_size = (int32_t)plan_info[FI_PP_KV_CHUNK_SIZE];
    kp.kv_chunk_size_ptr = (const int32_t*)((const char*)int_ws + plan_info[FI_PP_KV_CHUNK_SIZE_PTR_OFFSET]);
  }
  kp.alibi_slopes = a->alibi_slopes;
  kp.scale_q = a->scale_q;
  kp.scale_k = a->scale_k;
  kp.scale_v = a->scale_v;
  kp.q_stride_n = a->q_stride_n;
  kp.q_stride_h = a->q_stride_h;
  kp.kv_stride_page = kv.stride_page;
  kp.kv_stride_n = kv.stride_n;
  kp.kv_stride_h = kv.stride_h;
  kp.num_work = (int32_t)num_work;
  kp.num_qo_heads = a->num_qo_heads;
  kp.num_kv_heads = kv.num_kv_heads;
  kp.group_size = a->num_qo_heads / kv.num_kv_heads;
  kp.page_size = kv.page_size;
  kp.page_div = FastDiv((uint32_t)kv.page_size);
  kp.group_div = FastDiv((uint32_t)kp.group_size);
  // multi-item scoring is causal plus the per-item predicate (ref: prefill.cuh:845-856); plan() must have
  // been called with causal = true so that the kv range of a q tile ends at its last row
  kp.causal = a->mask_mode == FI_MASK_CAUSAL || a->mask_mode == FI_MASK_MULTIITEMSCORING;
  if (a->mask_mode == FI_MASK_CUSTOM) {
    kp.custom_mask = a->custom_mask;
    kp.mask_indptr = a->mask_indptr;
  }
  if (a->mask_mode == FI_MASK_MULTIITEMSCORING) {
    kp.prefix_len_ptr = a->prefix_len_ptr;
    kp.token_pos_in_items_ptr = a->token_pos_in_items_ptr;
    kp.token_pos_in_items_len = a->token_pos_in_items_len;
  }
  kp.window_left = a->window_left;
  kp.use_alibi = a->pos_encoding_mode == FI_POS_ALIBI;
  kp.o_dtype = a->o_dtype;
  kp.fp8_p_quant = a->q_dtype == FI_DTYPE_FP8_E4M3 || a->q_dtype == FI_DTYPE_FP8_E5M2;
  kp.logits_soft_cap = a->logits_soft_cap > 0.f ? a->logits_soft_cap : 0.f;
  kp.sm_scale = a->sm_scale;
  kp.rope_rcp_scale = a->rope_rcp_scale;
  kp.rope_rcp_theta = a->rope_rcp_theta;
  kp.bf16_pv_mode = a->bf16_pv_mode;
  kp.tile_q = (int32_t)plan_info[FI_PP_CTA_TILE_Q];
  const bool fp8_native = use_fp8_native(kp, a->q_dtype, kv.dtype, kv.head_dim, a->pos_encoding_mode == FI_POS_ROPE_LLAMA);
  FI_REQUIRE(kp.tile_q == kTileQ || fp8_native,
             "batch_prefill_paged_run: the plan was cut for %d-row q tiles (fi_batch_prefill_plan_tile), which only "
             "the fp8-native kernel runs: fp8 q/k/v of one type, head_dim 128, no RoPE / ALiBi / soft cap / window / mask",
             kp.tile_q);
  if (fp8_native) {
    FI_HIP_CALL(prefill_fp8_launch(kp, a->o_dtype, a->q_dtype == FI_DTYPE_FP8_E5M2, kv.head_dim, stream));
  } else {
    FI_HIP_CALL(fn(kp, a->pos_encoding_mode == FI_POS_ROPE_LLAMA, stream));
  }
  if (split) {
    // ref: VariableLengthMergeStates after the partition-kv kernel, prefill.cuh:2590-2671
    MergeNParams mp{kp.tmp_o, kp.tmp_lse, kp.merge_indptr, a->o, a->lse, 0,
                    (int32_t)plan_info[FI_PP_TOTAL_NUM_ROWS], a->num_qo_heads, kv.head_dim, FI_DTYPE_F32,
                    a->o_dtype, /*skip_empty=*/1};
    FI_HIP_CALL(launch_merge_n(mp, stream));
  }
  return 0;
}

extern "C" FI_API int fi_single_prefill_run(const fi_single_prefill_params_t* a, void* tmp,
                                            size_t tmp_bytes, fi_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FI_REQUIRE(a, "single_prefill_run: null params");
  if (a->qo_len == 0) return 0;
  FI_REQUIRE(a->q && a->k && a->v && a->o, "single_prefill_run: null tensor");
  FI_REQUIRE(a->num_kv_heads > 0 && a->num_qo_heads % a->num_kv_heads == 0,
             "single_prefill_run: num_qo_heads must be a multiple of num_kv_heads");
  FI_REQUIRE(a->mask_mode >= FI_MASK_NON_CAUSAL && a->mask_mode <= FI_MASK_CUSTOM,
             "single_prefill_run: bad mask_mode %d", a->mask_mode);
  FI_REQUIRE(a->mask_mode != FI_MASK_CUSTOM || a->custom_mask,
             "single_prefill_run: mask_mode CUSTOM needs custom_mask");
  if (check_prefill_dtypes("single_prefill_run", a->q_dtype, a->kv_dtype, a->o_dtype)) return 1;
  const int t16 = compute_type(a->q_dtype, a->o_dtype);
  prefill_launch_fn fn = find_prefill(t16, a->kv_dtype, a->q_dtype, a->head_dim);
  FI_REQUIRE(fn, "single_prefill_run: unsupported q/kv dtype %d/%d or head_dim %d", a->q_dtype,
             a->kv_dtype, a->head_dim);
  FI_REQUIRE(a->pos_encoding_mode != FI_POS_ALIBI || a->alibi_slopes,
             "single_prefill_run: ALIBI needs alibi_slopes");
  const size_t qsz = dtype_size(a->q_dtype), ksz = dtype_size(a->kv_dtype);
  FI_REQUIRE(((uintptr_t)a->q % 16) == 0 && (a->q_stride_n * qsz) % (8 * qsz) == 0 &&
                 (a->q_stride_h * qsz) % (8 * qsz) == 0,
             "single_prefill_run: q rows must be aligned to 8 elements");
  FI_REQUIRE((a->kv_stride_n * ksz) % (8 * ksz) == 0 && (a->kv_stride_h * ksz) % (8 * ksz) == 0 &&
                 ((uintptr_t)a->k % 16) == 0 && ((uintptr_t)a->v % 16) == 0,
             "single_prefill_run: k/v rows must be aligned to 8 elements");
  const int vpage = 16;  // dense tensor == identity page table of 16-token pages
  FI_REQUIRE(a->kv_stride_n >= 0 && (int64_t)vpage * a->kv_stride_n < (1ll << 31),
             "single_prefill_run: kv token stride must be below 2^27 elements");
  PrefillKernelParams kp;
  memset(&kp, 0, sizeof(kp));
  kp.q = a->q;
  kp.o = a->o;
  kp.lse = a->lse;
  kp.k = a->k;
  kp.v = a->v;
  kp.alibi_slopes = a->alibi_slopes;
  kp.scale_q = a->scale_q;
  kp.scale_k = a->scale_k;
  kp.scale_v = a->scale_v;
  kp.q_stride_n = a->q_stride_n;
  kp.q_stride_h = a->q_stride_h;
  kp.kv_stride_page = (int64_t)vpage * a->kv_stride_n;
  kp.kv_stride_n = a->kv_stride_n;
  kp.kv_stride_h = a->kv_stride_h;
  kp.num_qo_heads = a->num_qo_heads;
  kp.num_kv_heads = a->num_kv_heads;
  kp.group_size = a->num_qo_heads / a->num_kv_heads;
  kp.tile_q = kTileQ;
  kp.num_work = (int32_t)ceil_div<int64_t>((int64_t)a->qo_len * kp.group_size, kTileQ);
  kp.page_size = vpage;
  kp.page_div = FastDiv((uint32_t)vpage);
  kp.group_div = FastDiv((uint32_t)kp.group_size);
  kp.single_qo_len = a->qo_len;
  kp.single_kv_len = a->kv_len;
  kp.causal = a->mask_mode == FI_MASK_CAUSAL;
  if (a->mask_mode == FI_MASK_CUSTOM) kp.custom_mask = a->custom_mask;
  kp.window_left = a->window_left;
  kp.use_alibi = a->pos_encoding_mode == FI_POS_ALIBI;
  kp.o_dtype = a->o_dtype;
  kp.fp8_p_quant = a->q_dtype == FI_DTYPE_FP8_E4M3 || a->q_dtype == FI_DTYPE_FP8_E5M2;
  kp.logits_soft_cap = a->logits_soft_cap > 0.f ? a->logits_soft_cap : 0.f;
  kp.sm_scale = a->sm_scale;
  kp.rope_rcp_scale = a->rope_rcp_scale;
  kp.rope_rcp_theta = a->rope_rcp_theta;
  kp.bf16_pv_mode = a->bf16_pv_mode;
  // split the kv axis when the q tiles alone cannot fill the chip and the caller lent a scratch buffer
  // (same search as the batch planner; ref: PrefillBinarySearchKVChunkSize, scheduler.cuh:101-130)
  if (tmp && tmp_bytes > 0 && a->mask_mode != FI_MASK_CUSTOM) {
    const int64_t q_tiles = kp.num_work;
    const int64_t max_items = std::max<int64_t>((int64_t)fi_num_compute_units() * 2 / a->num_kv_heads, 1);
    int64_t span = std::max<int64_t>(a->kv_len, 1);
    if (a->window_left >= 0)
      span = std::min<int64_t>(span, (int64_t)a->window_left + (kp.causal ? kTileQ : a->qo_len) + kTileKV);
    int64_t low = 128 / kTileKV, high = ceil_div<int64_t>(span, kTileKV);
    while (low < high) {
      const int64_t mid = (low + high) / 2;
      if (q_tiles * ceil_div<int64_t>(span, mid * kTileKV) > max_items) low = mid + 1; else high = mid;
    }
    int64_t chunk = std::max<int64_t>(low, 128 / kTileKV) * kTileKV;
    {
      const int64_t rows = a->qo_len;
      chunk = price_kv_chunk(chunk, span, 1, &q_tiles, &span, &rows, a->num_kv_heads, a->num_qo_heads, a->head_dim,
                             a->head_dim);
    }
    auto need = [&](int64_t c) {
      return ((int64_t)a->qo_len * ceil_div<int64_t>(span, c) * a->num_qo_heads * (a->head_dim + 1) + 64) *
             (int64_t)sizeof(float);
    };
    while (chunk < span && need(chunk) > (int64_t)tmp_bytes) chunk *= 2;
    const int64_t nchunks = ceil_div<int64_t>(span, chunk);
    if (nchunks > 1 && q_tiles * nchunks < (1ll << 30)) {
      kp.num_kv_chunks = (int32_t)nchunks;
      kp.kv_chunk_size = (int32_t)chunk;
      kp.num_work = (int32_t)(q_tiles * nchunks);
      kp.tmp_o = (float*)tmp;
      size_t vbytes = (size_t)a->qo_len * nchunks * a->num_qo_heads * a->head_dim * sizeof(float);
      vbytes = (vbytes + 15) / 16 * 16;
      kp.tmp_lse = (float*)((char*)tmp + vbytes);
    }
  }
  if (use_fp8_native(kp, a->q_dtype, a->kv_dtype, a->head_dim, a->pos_encoding_mode == FI_POS_ROPE_LLAMA)) {
    FI_HIP_CALL(prefill_fp8_launch(kp, a->o_dtype, a->q_dtype == FI_DTYPE_FP8_E5M2, a->head_dim, stream));
  } else {
    FI_HIP_CALL(fn(kp, a->pos_encoding_mode == FI_POS_ROPE_LLAMA, stream));
  }
  if (kp.num_kv_chunks > 1) {
    // partial states are [qo_len, chunks, Hq, D]: the dense n-way merge
    MergeNParams mp{kp.tmp_o, kp.tmp_lse, nullptr, a->o, a->lse, kp.num_kv_chunks, a->qo_len, a->num_qo_heads,
                    a->head_dim, FI_DTYPE_F32, a->o_dtype, 0};
    FI_HIP_CALL(launch_merge_n(mp, stream));
  }
  return 0;
}

/*
 * fi_mi355.h -- C ABI of libfi_mi355.so, the MI355X (gfx950) implementation of FlashInfer's
 * batch paged-KV attention hot path (decode / prefill / cascade merge / page append) and the fp8
 * groupwise (grouped) GEMM.
 *
 * Every entry point replaces one TVM-FFI export of the reference (FlashInfer v0.3.1); the export it
 * stands in for is cited as `ref: file:line` (paths relative to the reference checkout).  The
 * reference passes DLPack tensor views; this ABI passes the same information as plain device pointers,
 * element strides and sizes, so it can be bound from ctypes / cgo / JNI without torch types.
 *
 * Conventions
 *   - return value: 0 on success, non-zero on failure; fi_last_error() returns a thread-local message
 *     (the reference throws flashinfer::Error / TVM_FFI_ICHECK, ref: include/flashinfer/exception.h:23).
 *   - all tensors are borrowed; outputs are pre-allocated by the caller
 *     (ref: flashinfer/decode.py:1262-1275).
 *   - `stream` is a hipStream_t passed as void* (the reference takes the current torch stream,
 *     ref: csrc/tvm_ffi_utils.h:256-264).  No call synchronises the device.
 *   - strides are in ELEMENTS, not bytes (as paged_kv_t, ref: include/flashinfer/page.cuh:127-145).
 *   - log-sum-exp values are base 2 (ref: include/flashinfer/attention/state.cuh:45), and an empty
 *     KV range yields o = 0, lse = FI_NEG_INF (-5e4, ref: include/flashinfer/math.cuh:32).
 */
#ifndef FI_MI355_H_
#define FI_MI355_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FI_ABI_VERSION 2
#define FI_NEG_INF (-5.0e4f)

typedef void* fi_stream_t; /* hipStream_t */

#if defined(FI_BUILDING_LIB)
#define FI_API __attribute__((visibility("default")))
#else
#define FI_API
#endif

/* element types (ref: dtype dispatch in csrc/tvm_ffi_utils.h:70-200) */
enum fi_dtype {
  FI_DTYPE_F16 = 0,
  FI_DTYPE_BF16 = 1,
  FI_DTYPE_FP8_E4M3 = 2, /* OCP e4m3fn == torch.float8_e4m3fn */
  FI_DTYPE_FP8_E5M2 = 3,
  FI_DTYPE_F32 = 4
};

/* ref: flashinfer/utils.py:30-46 */
enum fi_pos_encoding_mode { FI_POS_NONE = 0, FI_POS_ROPE_LLAMA = 1, FI_POS_ALIBI = 2 };
enum fi_mask_mode { FI_MASK_NON_CAUSAL = 0, FI_MASK_CAUSAL = 1, FI_MASK_CUSTOM = 2, FI_MASK_MULTIITEMSCORING = 3 };

/* ------------------------------------------------------------------------------------------------
 * library / device information
 * ---------------------------------------------------------------------------------------------- */
FI_API const char* fi_last_error(void);
FI_API int fi_abi_version(void);
/* number of compute units the planner balances for (hipDeviceProp.multiProcessorCount of the current
 * device, or FI_NUM_CUS from the environment, or 256 when no device is visible). */
FI_API int fi_num_compute_units(void);

/* ------------------------------------------------------------------------------------------------
 * Paged KV cache view.  ref: paged_kv_t, include/flashinfer/page.cuh:37-210.
 *   element offset of (page, head, entry, feat) = page*stride_page + head*stride_h + entry*stride_n + feat
 *   kv_len(b) = (indptr[b+1]-indptr[b]-1)*page_size + last_page_len[b]   (0 when the request has no page)
 * `indices == NULL` means the identity page table (page i of request 0 is physical page i): this is how
 * a dense [kv_len, H, D] tensor is addressed by fi_single_decode_run.
 * ---------------------------------------------------------------------------------------------- */
typedef struct fi_paged_kv {
  const void* k_data;
  const void* v_data;
  const int32_t* indptr;        /* [batch_size+1] device */
  const int32_t* indices;       /* [indptr[batch_size]] device */
  const int32_t* last_page_len; /* [batch_size] device */
  const int32_t* rope_pos_offset; /* optional [batch_size] device, NULL = 0 */
  int64_t stride_page, stride_n, stride_h;
  int32_t page_size;
  int32_t num_kv_heads;
  int32_t head_dim;
  int32_t batch_size;
  int32_t dtype; /* fi_dtype */
} fi_paged_kv_t;

/* ------------------------------------------------------------------------------------------------
 * Batch decode.  ref: BatchDecodeWithPagedKVCachePlan / Run, csrc/batch_decode.cu:39-79, 81-191;
 * planner DecodePlan, include/flashinfer/attention/scheduler.cuh:424-493.
 * ---------------------------------------------------------------------------------------------- */
#define FI_DECODE_PLAN_INFO_LEN 16
/* plan_info (int64[FI_DECODE_PLAN_INFO_LEN]); the reference's DecodePlanInfo has 10 entries
 * (scheduler.cuh:391-402); entries 0..9 keep their meaning, 10.. are ours. */
enum fi_decode_plan_slot {
  FI_DP_PADDED_BATCH_SIZE = 0, /* number of (request, kv-chunk) work items launched */
  FI_DP_V_OFFSET = 1,          /* byte offset of tmp_v in the float workspace */
  FI_DP_S_OFFSET = 2,          /* byte offset of tmp_s in the float workspace */
  FI_DP_REQUEST_INDICES_OFFSET = 3,
  FI_DP_KV_TILE_INDICES_OFFSET = 4,
  FI_DP_O_INDPTR_OFFSET = 5,
  FI_DP_BLOCK_VALID_MASK_OFFSET = 6,
  FI_DP_KV_CHUNK_SIZE_PTR_OFFSET = 7,
  FI_DP_ENABLE_CUDA_GRAPH = 8,
  FI_DP_SPLIT_KV = 9,
  FI_DP_KV_CHUNK_SIZE = 10, /* tokens per chunk */
  FI_DP_NUM_WORK = 11,      /* valid work items (<= padded) */
  FI_DP_BATCH_SIZE = 12,
  FI_DP_INT_BYTES_USED = 13,
  FI_DP_WINDOW_LEFT = 14,   /* sliding window the chunks were cut for (-1: none); run() must pass the same */
  FI_DP_MAGIC = 15
};
#define FI_DECODE_PLAN_MAGIC 0x4649444543ll /* "FIDEC" */

/* Host-side planning.  Writes the work list into `pinned_int_ws` and, when `int_ws` is non-NULL,
 * enqueues ONE host-to-device copy of it on `stream` (ref: scheduler.cuh:488-491).  `int_ws == NULL`
 * plans on the host only (used by CPU tests).  `indptr_h` is the HOST copy of the page indptr.
 * `max_grid_hint` <= 0 lets the library size the grid from the device (CUs x resident waves).
 * `window_left` >= 0 (the reference passes it to plan too, csrc/batch_decode.cu:39-79): only the pages that
 * can intersect the window of the last token are partitioned into chunks; the pages before
 * max(0, (num_pages - 1) * page_size - window_left) / page_size are never read. */
FI_API int fi_batch_decode_plan(void* float_ws, size_t float_ws_bytes, void* int_ws, void* pinned_int_ws,
                         size_t int_ws_bytes, const int32_t* indptr_h, int32_t batch_size,
                         int32_t num_qo_heads, int32_t num_kv_heads, int32_t page_size,
                         int32_t enable_cuda_graph, int32_t head_dim, int32_t q_dtype,
                         int32_t kv_dtype, int32_t max_grid_hint, int32_t window_left,
                         int64_t* plan_info_out, fi_stream_t stream);

typedef struct fi_batch_decode_params {
  const void* q; /* [batch, num_qo_heads, head_dim], strides below */
  int64_t q_stride_n, q_stride_h;
  fi_paged_kv_t kv;
  void* o;    /* [batch, num_qo_heads, head_dim] contiguous, dtype = q dtype */
  float* lse; /* optional [batch, num_qo_heads] */
  const float* alibi_slopes;    /* [num_qo_heads], used when pos_encoding_mode == FI_POS_ALIBI */
  const int32_t* q_rope_offset; /* optional [batch]; NULL = kv_len-1 (ref: decode.cuh:445-450) */
  int32_t num_qo_heads;
  int32_t q_dtype;           /* FI_DTYPE_F16 / FI_DTYPE_BF16 */
  int32_t pos_encoding_mode; /* fi_pos_encoding_mode */
  int32_t window_left;       /* -1 = full */
  float logits_soft_cap;     /* 0 = off */
  float sm_scale;
  float rope_rcp_scale; /* 1/rope_scale */
  float rope_rcp_theta; /* 1/rope_theta */
} fi_batch_decode_params_t;

FI_API int fi_batch_decode_run(void* float_ws, size_t float_ws_bytes, void* int_ws, size_t int_ws_bytes,
                        const int64_t* plan_info, int32_t plan_info_len,
                        const fi_batch_decode_params_t* params, fi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Single-request decode over a dense KV tensor.
 * ref: single_decode_with_kv_cache, csrc/single_decode.cu:33-104; decode.cuh:658-737.
 * k, v: [kv_len, num_kv_heads, head_dim] (NHD) or [num_kv_heads, kv_len, head_dim] (HND) described by
 * strides.  `tmp` holds split-KV partial states (the reference uses a 32 MB cache buffer,
 * flashinfer/decode.py:486).
 * ---------------------------------------------------------------------------------------------- */
typedef struct fi_single_decode_params {
  const void* q; /* [num_qo_heads, head_dim] */
  int64_t q_stride_h;
  const void* k;
  const void* v;
  int64_t kv_stride_n, kv_stride_h;
  void* o;    /* [num_qo_heads, head_dim] */
  float* lse; /* optional [num_qo_heads] */
  const float* alibi_slopes;
  int32_t kv_len, num_qo_heads, num_kv_heads, head_dim;
  int32_t q_dtype, kv_dtype;
  int32_t pos_encoding_mode;
  int32_t window_left;
  float logits_soft_cap, sm_scale, rope_rcp_scale, rope_rcp_theta;
} fi_single_decode_params_t;

FI_API int fi_single_decode_run(const fi_single_decode_params_t* params, void* tmp, size_t tmp_bytes,
                         fi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Attention-state merge (cascade).  ref: csrc/cascade.cu:23-57 (merge_state), 59-100
 * (merge_state_in_place), 102-... (merge_states); kernels include/flashinfer/attention/cascade.cuh.
 *   (v, s) (+) (v', s'):  m = max(s, s'); w = 2^(s-m); w' = 2^(s'-m);
 *   v_out = (w v + w' v') / (w + w');  s_out = m + log2(w + w')
 * v: [seq_len, num_heads, head_dim] f16/bf16/f32 contiguous; s: [seq_len, num_heads] f32.
 * ---------------------------------------------------------------------------------------------- */
FI_API int fi_merge_state(const void* v_a, const float* s_a, const void* v_b, const float* s_b,
                   void* v_merged, float* s_merged, int32_t seq_len, int32_t num_heads,
                   int32_t head_dim, int32_t dtype, fi_stream_t stream);
/* mask: optional uint8[seq_len]; rows with mask==0 keep (v, s) unchanged (cascade.cuh:86-116). */
FI_API int fi_merge_state_in_place(void* v, float* s, const void* v_other, const float* s_other,
                            const uint8_t* mask, int32_t seq_len, int32_t num_heads,
                            int32_t head_dim, int32_t dtype, fi_stream_t stream);
/* v: [seq_len, num_index_sets, num_heads, head_dim], s: [seq_len, num_index_sets, num_heads]. */
FI_API int fi_merge_states(const void* v, const float* s, void* v_merged, float* s_merged,
                    int32_t num_index_sets, int32_t seq_len, int32_t num_heads, int32_t head_dim,
                    int32_t dtype, fi_stream_t stream);
/* Ragged form used after split-KV (ref: VariableLengthMergeStates, cascade.cuh:686-736):
 * v: [nnz, num_heads, head_dim] of in_dtype, s: [nnz, num_heads]; row r merges entries
 * indptr[r]..indptr[r+1]; 0 entries -> zeros / FI_NEG_INF. */
FI_API int fi_variable_length_merge_states(const void* v, const float* s, const int32_t* indptr,
                                    void* v_merged, float* s_merged, int32_t seq_len,
                                    int32_t num_heads, int32_t head_dim, int32_t in_dtype,
                                    int32_t out_dtype, fi_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * Batch prefill / append attention over a paged KV cache.
 * ref: BatchPrefillWithKVCachePlan csrc/batch_prefill.cu:47-74, BatchPrefillWithPagedKVCacheRun
 * csrc/batch_prefill.cu:199-327; fp8: csrc/batch_prefill_fp8_sm90.cu:39-67, 81-185; planner
 * PrefillPlan include/flashinfer/attention/scheduler.cuh:694-795.
 * ---------------------------------------------------------------------------------------------- */
#define FI_PREFILL_PLAN_INFO_LEN 16
enum fi_prefill_plan_slot {
  FI_PP_PADDED_BATCH_SIZE = 0, /* work items launched (q tiles over all requests) */
  FI_PP_TOTAL_NUM_ROWS = 1,
  FI_PP_KV_CHUNK_SIZE_PTR_OFFSET = 2, /* int workspace: int32 chunk size the kernels read (rewritten by plan) */
  FI_PP_CTA_TILE_Q = 3,
  FI_PP_REQUEST_INDICES_OFFSET = 4,
  FI_PP_QO_TILE_INDICES_OFFSET = 5,
  FI_PP_KV_TILE_INDICES_OFFSET = 6, /* int workspace, valid when SPLIT_KV */
  FI_PP_MERGE_INDPTR_OFFSET = 7,    /* int workspace: [total_num_rows + 1] partial-state ranges per qo row */
  FI_PP_BATCH_SIZE = 8,
  FI_PP_KV_CHUNK_SIZE = 9,          /* tokens */
  FI_PP_V_OFFSET = 10,              /* float workspace: f32 partial outputs */
  FI_PP_S_OFFSET = 11,              /* float workspace: f32 partial lse */
  FI_PP_NUM_WORK = 12,              /* real work items (<= PADDED_BATCH_SIZE) */
  FI_PP_ENABLE_CUDA_GRAPH = 13,
  FI_PP_SPLIT_KV = 14,
  FI_PP_MAGIC = 15
};
#define FI_PREFILL_PLAN_MAGIC 0x4649505245ll /* "FIPRE" */

/* qo_indptr_h / kv_indptr_h: HOST [batch+1]; kv_len_arr_h: HOST [batch]. int_ws == NULL plans on the
 * host only.  The kv axis is split into chunks when the (request, q tile) items alone cannot fill the chip
 * (binary search of the chunk size as PrefillBinarySearchKVChunkSize, scheduler.cuh:101-130); partial states
 * then go to the float workspace and are merged by run().  fixed_split_size (tokens, > 0) fixes the chunk
 * size, disable_split_kv forbids splitting (batch-invariant results). */
FI_API int fi_batch_prefill_plan(void* float_ws, size_t float_ws_bytes, void* int_ws, void* pinned_int_ws,
                          size_t int_ws_bytes, const int32_t* qo_indptr_h, const int32_t* kv_indptr_h,
                          const int32_t* kv_len_arr_h, int32_t total_num_rows, int32_t batch_size,
                          int32_t num_qo_heads, int32_t num_kv_heads, int32_t page_size,
                          int32_t enable_cuda_graph, int32_t head_dim_qk, int32_t head_dim_vo,
                          int32_t causal, int32_t window_left, int32_t fixed_split_size,
                          int32_t disable_split_kv, int64_t* plan_info_out, fi_stream_t stream);
/* The same planner for a caller-chosen q tile: cta_tile_q = 128 (every kernel; what fi_batch_prefill_plan uses) or
 * 256 -- the 8-wave form of the fp8-native kernel (e4m3 q/k/v, head_dim 128, plain logits), which halves the K/V
 * staging work per query row; run() rejects such a plan for anything that kernel does not cover. */
FI_API int fi_batch_prefill_plan_tile(void* float_ws, size_t float_ws_bytes, void* int_ws, void* pinned_int_ws,
                          size_t int_ws_bytes, const int32_t* qo_indptr_h, const int32_t* kv_indptr_h,
                          const int32_t* kv_len_arr_h, int32_t total_num_rows, int32_t batch_size,
                          int32_t num_qo_heads, int32_t num_kv_heads, int32_t page_size,
                          int32_t enable_cuda_graph, int32_t head_dim_qk, int32_t head_dim_vo,
                          int32_t causal, int32_t window_left, int32_t fixed_split_size,
                          int32_t disable_split_kv, int32_t cta_tile_q, int64_t* plan_info_out, fi_stream_t stream);

typedef struct fi_batch_prefill_params {
  const void* q; /* [nnz_qo, num_qo_heads, head_dim] */
  int64_t q_stride_n, q_stride_h;
  const int32_t* qo_indptr; /* [batch+1] device */
  fi_paged_kv_t kv;
  void* o;    /* [nnz_qo, num_qo_heads, head_dim] contiguous */
  float* lse; /* optional [nnz_qo, num_qo_heads] */
  const float* alibi_slopes;
  const float* scale_q; /* fp8 path: [num_qo_heads] (NULL = 1) */
  const float* scale_k; /* [num_kv_heads] */
  const float* scale_v; /* [num_kv_heads] */
  /* mask_mode CUSTOM (ref variants.cuh:57-86): bit (qo_idx * kv_len + kv_idx) of request r, little-endian
   * within bytes, starting at byte mask_indptr[r] of custom_mask (flashinfer.quantization.segment_packbits
   * layout, ref prefill.py:1203-1223, 1693-1706) */
  const uint8_t* custom_mask;
  const int32_t* mask_indptr; /* [batch+1] device, byte offsets */
  /* mask_mode MULTIITEMSCORING (ref: maybe_prefix_len_ptr / maybe_token_pos_in_items_ptr / maybe_max_item_len_ptr /
   * token_pos_in_items_len of paged_run, csrc/batch_prefill.cu:199-205, include/flashinfer/attention/prefill.cuh
   * :795-858): causal, and a query at position p >= prefix_len[b] sees only the prefix and the keys
   * kv_idx > p - token_pos_in_items[b][p - prefix_len[b]] (its own item).  max_item_len is accepted for
   * signature parity (the reference uses it to skip fully masked tiles) and may be NULL. */
  const uint32_t* prefix_len_ptr;         /* [batch] device */
  const uint16_t* token_pos_in_items_ptr; /* [batch, token_pos_in_items_len] device */
  const uint16_t* max_item_len_ptr;       /* [batch] device, optional */
  int32_t token_pos_in_items_len;
  int32_t num_qo_heads;
  int32_t q_dtype; /* f16 / bf16 / fp8_e4m3 (then kv must be fp8_e4m3 too) */
  int32_t o_dtype; /* f16 / bf16 */
  int32_t mask_mode; /* fi_mask_mode: NON_CAUSAL / CAUSAL / CUSTOM / MULTIITEMSCORING */
  int32_t pos_encoding_mode;
  int32_t window_left;
  float logits_soft_cap, sm_scale, rope_rcp_scale, rope_rcp_theta;
  /* bf16 queries only -- how the probabilities enter P.V (csrc/prefill_kernel.h, PMODE): 0 default = P.V on the f16
   * MFMA (P rounded to f16, V converted bf16 -> f16 while staged: exact for 2^-14 <= |v| < 65504, the result meets the
   * reference's 1e-3 bar); 1 = P as hi + lo bf16 halves on the bf16 MFMA (no range limit on V; ~25 % slower) -- for
   * caches that may hold |v| >= 65504 or many |v| < 6e-5; 2 = the default, explicitly; 3 = the reference's single bf16
   * rounding of P (prefill.cuh:962-985).  Ignored for other q dtypes. */
  int32_t bf16_pv_mode;
} fi_batch_prefill_params_t;

/* Ragged (non-paged) KV, ref BatchPrefillWithRaggedKVCacheRun csrc/batch_prefill.cu:76-197: pass the
 * [nnz_kv, H, D] tensors as a cache with page_size = 1, stride_page = stride_n, kv.indptr = the ragged
 * kv_indptr, kv.indices = NULL (identity) and kv.last_page_len = NULL (all pages full). */
FI_API int fi_batch_prefill_paged_run(void* float_ws, size_t float_ws_bytes, void* int_ws, size_t int_ws_bytes,
                               const int64_t* plan_info, int32_t plan_info_len,
                               const fi_batch_prefill_params_t* params, fi_stream_t stream);

/* Single-request prefill over dense K/V.  ref: csrc/single_prefill.cu, flashinfer/prefill.py:960-1194. */
typedef struct fi_single_prefill_params {
  const void* q; /* [qo_len, num_qo_heads, head_dim] */
  int64_t q_stride_n, q_stride_h;
  const void* k; /* [kv_len, H, D] (NHD) or [H, kv_len, D] (HND) by strides */
  const void* v;
  int64_t kv_stride_n, kv_stride_h;
  void* o;
  float* lse;
  const float* alibi_slopes;
  const float* scale_q;
  const float* scale_k;
  const float* scale_v;
  const uint8_t* custom_mask; /* mask_mode CUSTOM: packed [qo_len * kv_len] bits, little-endian */
  int32_t qo_len, kv_len, num_qo_heads, num_kv_heads, head_dim;
  int32_t q_dtype, kv_dtype, o_dtype;
  int32_t mask_mode, pos_encoding_mode, window_left;
  float logits_soft_cap, sm_scale, rope_rcp_scale, rope_rcp_theta;
  int32_t bf16_pv_mode; /* as fi_batch_prefill_params_t.bf16_pv_mode */
} fi_single_prefill_params_t;

FI_API int fi_single_prefill_run(const fi_single_prefill_params_t* params, void* tmp, size_t tmp_bytes,
                          fi_stream_t stream);

/* Bit packing of boolean masks (numpy.packbits semantics).  ref: csrc/quantization.cu,
 * include/flashinfer/quantization.cuh:29-126, Python flashinfer/quantization.py:57-153.
 *   x: n bytes, each 0 / non-zero;  y: ceil(n / 8) bytes;  bitorder_little: 0 = "big", 1 = "little".
 * Segment form: segment i is x[in_indptr[i] : in_indptr[i+1]], packed into
 * y[out_indptr[i] : out_indptr[i+1]] with out_indptr[i+1] - out_indptr[i] = ceil(len_i / 8) (both int32
 * device arrays of batch + 1 entries; y_bytes = out_indptr[batch], passed by the host for the launch). */
FI_API int fi_packbits(const uint8_t* x, int64_t n, int32_t bitorder_little, uint8_t* y, fi_stream_t stream);
FI_API int fi_segment_packbits(const uint8_t* x, const int32_t* in_indptr, const int32_t* out_indptr,
                               int32_t batch, int64_t y_bytes, int32_t bitorder_little, uint8_t* y,
                               fi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * fp8 groupwise-scaled GEMM / grouped GEMM ("nt": D = A . B^T, B given as (n, k)).
 * ref: gemm_fp8_nt_groupwise csrc/gemm_sm100_binding.cu:23, csrc/gemm_groupwise_sm100.cu:89-120;
 * group_gemm_fp8_nt_groupwise csrc/group_gemm_sm100_binding.cu:34,
 * csrc/group_gemm_fp8_groupwise_sm100.cu:89-124; Python flashinfer/gemm.py:2321-2484, 2657-2811.
 *   a: (m | cum_m, k) fp8;  b: (n, k) | (G, n, k) fp8;  d: (m | cum_m, n) f16/bf16
 *   scale_k_major = 0 ("MN"): a_scale (k/128, m/gran_m), b_scale ([G,] k/128, n/128)
 *   scale_k_major = 1 ("K") : a_scale (m/gran_m, k/128), b_scale ([G,] n/128, k/128)
 *   m_indptr: int32 [G+1] on the DEVICE (ref: gemm.py:2689-2691; entries multiples of 4).
 * The reference's workspace and mma_sm arguments have no counterpart (no workspace is needed).
 * ---------------------------------------------------------------------------------------------- */
FI_API int fi_gemm_fp8_nt_groupwise(const void* a, const void* b, const void* a_scale, const void* b_scale,
                             void* d, int32_t m, int32_t n, int32_t k, int32_t gran_m, int32_t gran_n,
                             int32_t gran_k, int32_t scale_k_major, int32_t a_dtype, int32_t b_dtype,
                             int32_t d_dtype, fi_stream_t stream);
FI_API int fi_group_gemm_fp8_nt_groupwise(const void* a, const void* b, const void* a_scale,
                                   const void* b_scale, void* d, const int32_t* m_indptr,
                                   int32_t num_groups, int32_t cum_m, int32_t n, int32_t k,
                                   int32_t gran_m, int32_t gran_n, int32_t gran_k, int32_t scale_k_major,
                                   int32_t a_dtype, int32_t b_dtype, int32_t d_dtype, fi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Page-table operations.  ref: append_paged_kv_cache csrc/flashinfer_page_binding.cu:36, csrc/page.cu,
 * include/flashinfer/page.cuh:258-284, 345-411; get_batch_indices_positions flashinfer/page.py:169-221.
 *   positions[i] = i + seq_lens[b] - append_indptr[b+1]   for append_indptr[b] <= i < append_indptr[b+1]
 *   append: K/V row i (shape [nnz, num_kv_heads, head_dim], element strides given) is written at
 *           page kv.indices[kv.indptr[b] + pos / page_size], entry pos % page_size.
 * `kv->k_data` / `kv->v_data` are WRITTEN by fi_append_paged_kv_cache; kv->last_page_len is unused.
 * ---------------------------------------------------------------------------------------------- */
FI_API int fi_get_batch_indices_positions(const int32_t* append_indptr, const int32_t* seq_lens,
                                   int32_t batch_size, int32_t nnz, int32_t* batch_indices,
                                   int32_t* positions, fi_stream_t stream);
FI_API int fi_append_paged_kv_cache(const void* append_key, const void* append_value, int64_t k_stride_n,
                             int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
                             const int32_t* batch_indices, const int32_t* positions, int32_t nnz,
                             const fi_paged_kv_t* kv, fi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Standalone rotary embedding (SURVEY.md 8f "next" row).  ref: apply_rope / apply_rope_pos_ids /
 * apply_llama31_rope* / apply_rope_pos_ids_cos_sin_cache in csrc/rope.cu, include/flashinfer/pos_enc.cuh:465-1070,
 * flashinfer/rope.py:321-1150.  One entry point: positions are explicit (fi_rope_positions_from_indptr
 * expands the (indptr, offsets) form), llama-3.1 scaling enters through smooth_a / smooth_b
 * (pos_enc.cuh:976-977; 0, 0 = plain RoPE), cos_sin_cache != NULL selects the table form.
 * q/k: [nnz, heads, head_dim] f16/bf16 with element strides; q_out/k_out may alias q/k (in place).
 * ---------------------------------------------------------------------------------------------- */
typedef struct fi_rope_params {
  const void* q;
  const void* k;
  void* q_out;
  void* k_out;
  const int32_t* pos_ids;     /* [nnz] device */
  const float* cos_sin_cache; /* optional [max_pos, rotary_dim] f32: cos half | sin half */
  int64_t q_stride_n, q_stride_h, k_stride_n, k_stride_h;
  int64_t qo_stride_n, qo_stride_h, ko_stride_n, ko_stride_h;
  int32_t nnz, num_q_heads, num_k_heads, head_dim, rotary_dim;
  int32_t interleave; /* 1: pairs (2i, 2i+1); 0: pairs (i, i + rotary_dim/2) */
  int32_t dtype;
  float rope_rcp_scale, rope_rcp_theta, smooth_a, smooth_b;
} fi_rope_params_t;

FI_API int fi_apply_rope_pos_ids(const fi_rope_params_t* params, fi_stream_t stream);
/* RoPE fused with the cache append ("fused RoPE-then-append"; the reference runs
 * apply_rope_pos_ids (csrc/rope.cu) and then append_paged_kv_cache (csrc/page.cu:25-120) -- two passes over k):
 * q is rotated into q_out as above; each rotated k row is written ONCE, into the paged cache at
 * (batch_indices[i], positions[i]) (page.cuh:272-275), and the v row is copied beside it.  params->k_out and
 * its strides are ignored; params->pos_ids are the rotation positions (normally == positions); the cache must
 * have k's dtype.  Bit-identical to the two calls it replaces. */
FI_API int fi_apply_rope_append_paged_kv_cache(const fi_rope_params_t* params, const void* append_value,
                                        int64_t v_stride_n, int64_t v_stride_h, const int32_t* batch_indices,
                                        const int32_t* positions, const fi_paged_kv_t* kv, fi_stream_t stream);
/* pos_ids[i] = offsets[b] + i - indptr[b] for indptr[b] <= i < indptr[b+1] (ref: pos_enc.cuh:540-575) */
FI_API int fi_rope_positions_from_indptr(const int32_t* indptr, const int32_t* offsets, int32_t batch_size,
                                  int32_t nnz, int32_t* pos_ids, fi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FI_MI355_H_ */

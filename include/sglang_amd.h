/*
 * sglang_amd.h -- C ABI of the MI355X (gfx950) RadixAttention hot-path library
 * (libsglang_amd.so).
 *
 * This is the drop-in boundary under SGLang's operator surface: every entry
 * point below is what the reference's Python op wrappers bind for this path
 * (torch.ops.sgl_kernel.* schemas in
 * /root/reference/python/sglang/kernels/aot/csrc/common_extension_rocm.cc:25-246
 * and the Triton launchers they replace).  The reference-side ctypes binding a
 * maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers
 *     unless a parameter says "host";
 *   - bf16 tensors are passed as void* / uint16 storage; strides are in ELEMENTS;
 *   - the caller owns every buffer, including workspaces -- the library never
 *     allocates device memory and keeps no mutable device state.  ONE exception:
 *     the tensor-parallel communicator workspace comes from sgl_amd_xgmi_alloc()
 *     (an uncached, hipIpc-exportable allocation no framework allocator hands
 *     out) and goes back through sgl_amd_xgmi_free();
 *   - every function enqueues on `stream` (a hipStream_t passed as void*; NULL =
 *     the null stream), never synchronises, and is safe inside hipGraph capture;
 *   - return 0 on success, negative on error (-1 bad argument, -2 launch
 *     failure); sgl_amd_last_error() returns a thread-local message.  Nothing
 *     throws across the ABI.
 */
#ifndef SGLANG_AMD_H_
#define SGLANG_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGL_AMD_ABI_VERSION 1

/* flags for the attention entry points */
#define SGL_AMD_ATTN_FLAG_NO_DPP 1 /* use ds_bpermute shuffles instead of DPP row ops (debug A/B) */

const char* sgl_amd_last_error(void);
int sgl_amd_abi_version(void);
/* "gfx950" -- the only code object in the library. */
const char* sgl_amd_target_arch(void);

/* ---- RMSNorm (reference: srt/layers/layernorm.py:423,777-826;
 *      sgl_kernel.rmsnorm / fused_add_rmsnorm, common_extension_rocm.cc) ------ */
/* out[r,:] = bf16((x[r,:] * rsqrt(mean(x^2)+eps)) * weight), fp32 math. */
int sgl_amd_rmsnorm(const void* x, const void* weight, void* out, int64_t num_rows, int hidden,
                    int64_t x_row_stride, int64_t out_row_stride, float eps, void* stream);
/* hidden_out[r,:] = table[ids[r],:] (the embedding lookup of a decode step: models/llama.py:433-437 `embed_tokens(input_ids)`, the
 * output becomes the residual stream) and out[r,:] = RMSNorm(hidden_out[r,:]) * weight (the first layer's input_layernorm,
 * :349-353) in one launch.  bf16; ids int64 in [0, vocab). */
int sgl_amd_embedding_rmsnorm(const int64_t* ids, const void* table, const void* weight, void* hidden_out, void* out,
                              int64_t num_rows, int hidden, int64_t vocab, int64_t table_row_stride, int64_t hidden_row_stride,
                              int64_t out_row_stride, float eps, void* stream);
/* in place: residual <- bf16(x + residual); x <- rmsnorm(fp32(x + residual)). */
int sgl_amd_fused_add_rmsnorm(void* x, void* residual, const void* weight, int64_t num_rows,
                              int hidden, int64_t x_row_stride, int64_t res_row_stride, float eps,
                              void* stream);

/* ---- SiLU-and-mul (reference: srt/layers/activation.py:130-150;
 *      sgl_kernel.silu_and_mul, kernels/aot/csrc/elementwise/activation.cu) --- */
/* in [rows, 2d] -> out [rows, d].  round_intermediate=1 reproduces the torch-native
 * bf16 rounding of silu(x) before the multiply; 0 keeps the product in fp32. */
int sgl_amd_silu_and_mul(const void* in, void* out, int64_t num_rows, int d, int64_t in_row_stride,
                         int64_t out_row_stride, int round_intermediate, void* stream);

/* ---- Rotary embedding (reference: srt/layers/rotary_embedding/base.py:236-276,
 *      utils.py:36-63; sgl_kernel.rotary_embedding, common_extension_rocm.cc:236-240) */
/* In place on q [T,Hq,D] and k [T,Hk,D].  cos_sin_cache [max_pos, rot_dim] = cos||sin,
 * bf16 (cache_is_f32=0) or fp32 (1).  If k_cache != NULL the rotated K row and the V row
 * of every token are also scattered to k_cache/v_cache[cache_loc[t]] (fused rope + KV
 * store, base.py:385-417). */
int sgl_amd_rotary_embedding(const int64_t* positions, void* q, void* k, const void* cos_sin_cache,
                             int cache_is_f32, int64_t num_tokens, int num_q_heads, int num_k_heads,
                             int head_dim, int rot_dim, int64_t q_token_stride,
                             int64_t k_token_stride, int is_neox, const void* v,
                             int64_t v_token_stride, void* k_cache, void* v_cache,
                             const int64_t* cache_loc, int64_t cache_row_stride, void* stream);

/* ---- KV store (reference: srt/mem_cache/memory_pool.py:141-193 store_cache) ---- */
/* k_cache[loc[t], :] = k[t, :]; v_cache[loc[t], :] = v[t, :]. */
int sgl_amd_store_kv_cache(const void* k, const void* v, void* k_cache, void* v_cache,
                           const int64_t* loc, int64_t num_tokens, int k_row_elems,
                           int v_row_elems, int64_t k_token_stride, int64_t v_token_stride,
                           int64_t k_cache_row_stride, int64_t v_cache_row_stride, void* stream);

/* ---- Integer metadata (bit-exact) ------------------------------------------- */
/* kv_indices[kv_indptr[b]+i] = req_to_token[req_pool_indices[b], kv_start_idx[b]+i]
 * (reference: kernels/ops/attention/utils.py create_flashinfer_kv_indices_triton,
 *  call site srt/layers/attention/triton_backend.py:447-455). */
int sgl_amd_create_kv_indices(const int32_t* req_to_token, int64_t req_to_token_stride,
                              const void* req_pool_indices, int req_pool_indices_is_i64,
                              const int32_t* kernel_lens, const int32_t* kv_indptr,
                              const int32_t* kv_start_idx /* may be NULL */, void* kv_indices,
                              int kv_indices_is_i64, int64_t batch, void* stream);
/* reference: srt/mem_cache/allocation.py:54-103 write_cache_indices.  prefix_ptrs is a
 * device array of `batch` device pointers to int64 prefix slot tensors (may be NULL). */
int sgl_amd_write_req_to_token(int32_t* req_to_token, int64_t req_to_token_stride,
                               const int64_t* req_pool_indices, const void* prefix_ptrs,
                               const int64_t* prefix_lens, const int64_t* seq_lens,
                               const int64_t* extend_lens, const int64_t* out_cache_loc,
                               int64_t batch, void* stream);
/* One decode step's per-request bookkeeping at page_size 1 (reference: schedule_batch.py prepare_for_decode,
 * allocation.py:512-560 alloc_for_decode, :73-82 write_req_to_token_pool): req_to_token[req_pool_indices[b],
 * seq_lens[b]] = new_slots[b]; out_cache_loc[b] = new_slots[b]; seq_lens[b] += 1 (int32, in place). */
int sgl_amd_decode_advance(int32_t* req_to_token, int64_t req_to_token_stride, const int64_t* req_pool_indices,
                           int32_t* seq_lens, const int64_t* new_slots, int64_t* out_cache_loc, int64_t batch,
                           void* stream);
/* reference: allocation.py:139-148 get_last_loc_torch. */
int sgl_amd_get_last_loc(const int32_t* req_to_token, int64_t req_to_token_stride,
                         const int64_t* req_pool_indices, const int64_t* prefix_lens,
                         int64_t* last_loc, int64_t batch, void* stream);
/* reference: srt/model_executor/forward_batch_info.py:1790-1804 compute_position_torch. */
int sgl_amd_compute_position(const void* extend_prefix_lens, const void* extend_seq_lens,
                             int lens_are_i64, int64_t* positions, void* extend_start_loc,
                             int64_t batch, void* stream);
/* reference: forward_batch_info.py:1807 _clamp_position_native. */
int sgl_amd_clamp_position(const void* seq_lens, int lens_are_i64, int64_t* positions,
                           int64_t batch, void* stream);
/* reference: srt/mem_cache/allocator/paged.py:45-102,172-260 (alloc_extend / alloc_decode). */
int sgl_amd_alloc_extend(const int64_t* prefix_lens, const int64_t* seq_lens,
                         const int64_t* last_loc, const int64_t* free_pages, int64_t* out_indices,
                         int64_t batch, int64_t page_size, void* stream);
int sgl_amd_alloc_decode(const int64_t* seq_lens, const int64_t* last_loc,
                         const int64_t* free_pages, int64_t* out_indices, int64_t batch,
                         int64_t page_size, void* stream);

/* ---- Attention (reference: AttentionBackend.forward_extend / forward_decode,
 *      srt/layers/attention/base_attn_backend.py:260-284; oracle
 *      torch_native_backend.py:61-398; replaced kernels
 *      kernels/ops/attention/{extend,decode}_attention.py) ---------------------- */
/* Decode: one query token per request.  q/out [B,Hq,D] bf16, KV pool [slots,Hkv,D] bf16.
 * Token t of request b lives in slot req_to_token[req_pool_indices[b], t]  (kv_indptr==NULL)
 * or req_to_token[kv_indptr[b] + t] (flat kv_indices form).  num_splits>1 needs
 * ws_acc fp32 [B,Hq,num_splits,D] and ws_ml fp32 [B,Hq,num_splits,2].  batch_order (optional, [B]
 * permutation, e.g. the cascade plan's) lays requests that share KV rows out on one XCD so the
 * shared rows are re-read from that XCD's L2. */
int sgl_amd_decode_attention(const void* q, const void* k_cache, const void* v_cache, void* out,
                             const int32_t* req_to_token, int64_t req_to_token_stride,
                             const int64_t* req_pool_indices, const int32_t* seq_lens,
                             const int32_t* kv_indptr, int64_t batch, int num_q_heads,
                             int num_kv_heads, int head_dim, int64_t q_token_stride,
                             int64_t out_token_stride, int64_t k_cache_row_stride,
                             int64_t v_cache_row_stride, float sm_scale, int num_splits,
                             void* ws_acc, void* ws_ml, const int32_t* batch_order, int flags,
                             void* stream);
int sgl_amd_decode_attention_min_chunk(void);
/* Extend (prefill): request b owns query tokens qo_indptr[b]..qo_indptr[b+1] of q/out
 * [T,Hq,D]; query i attends kv positions [0, prefix_lens[b]+i] (causal) or
 * [0, seq_lens[b]) (causal=0) read from the pool through req_to_token.  The new K/V
 * rows must already be in the pool (sgl_amd_store_kv_cache / fused rope store). */
int sgl_amd_extend_attention(const void* q, void* out, const void* k_cache, const void* v_cache,
                             const int32_t* req_to_token, int64_t req_to_token_stride,
                             const int64_t* req_pool_indices, const int32_t* seq_lens,
                             const int32_t* prefix_lens, const int32_t* qo_indptr, int64_t batch,
                             int max_extend_len, int num_q_heads, int num_kv_heads, int head_dim,
                             int64_t q_token_stride, int64_t out_token_stride,
                             int64_t k_cache_row_stride, int64_t v_cache_row_stride, float sm_scale,
                             int causal, void* stream);

/* ---- Shared-prefix (cascade) decode attention ---------------------------------------------------
 * RadixAttention batches share KV rows: requests whose req_to_token rows start with the same slots
 * read the same pool rows (reference: the Triton decode path of triton_backend.py:136-1012 re-reads
 * them once per request).  sgl_amd_cascade_plan (two launches per decode STEP, device-only, graph-safe)
 * groups such requests; sgl_amd_cascade_decode_attention (per layer, two launches) cuts every group's
 * shared prefix and every request's private suffix into sgl_amd_cascade_chunk_tokens()-token items,
 * reads each item's K/V rows once for all its member requests (MFMA), and merges the per-item
 * partials of a (request, head) in slot order (deterministic).
 * plan: int32[sgl_amd_cascade_plan_ints(batch, max_items)]; ws_acc fp32 [B,Hq,slots_total,D], ws_ml fp32
 * [B,Hq,slots_total,2] with slots_total >= ceil(max_context_len / chunk) + 1; head_dim 64 or 128. */
int sgl_amd_cascade_chunk_tokens(void);
int sgl_amd_cascade_members_per_item(int num_q_heads, int num_kv_heads);
int64_t sgl_amd_cascade_plan_ints(int64_t batch, int64_t max_items);
int sgl_amd_cascade_plan(const int32_t* req_to_token, int64_t req_to_token_stride,
                         const int64_t* req_pool_indices, const int32_t* seq_lens, int64_t batch,
                         int num_q_heads, int num_kv_heads, int min_shared_len, int64_t max_context_len,
                         int32_t* plan, int64_t max_items, void* stream);
int sgl_amd_cascade_decode_attention(const void* q, const void* k_cache, const void* v_cache, void* out,
                                     const int32_t* req_to_token, int64_t req_to_token_stride,
                                     const int64_t* req_pool_indices, const int32_t* seq_lens,
                                     const int32_t* plan, int64_t batch, int64_t max_items,
                                     int num_q_heads, int num_kv_heads, int head_dim,
                                     int64_t q_token_stride, int64_t out_token_stride,
                                     int64_t k_cache_row_stride, int64_t v_cache_row_stride,
                                     float sm_scale, int64_t max_context_len, int slots_total,
                                     void* ws_acc, void* ws_ml, void* stream);
/* The same over any pool format of srt/mem_cache/memory_pool.py (as sgl_amd_decode_attention_ex): kv_fp8 rows are OCP
 * e4m3 bytes of K / k_scale, V / v_scale (:2364-2374; k_cache_row_stride then counts bytes = elements of a row);
 * kv_layout_hnd pools are [pages, H_kv, page_size, D] (:2061-2117), page_size a power of two. */
int sgl_amd_cascade_decode_attention_ex(const void* q, const void* k_cache, const void* v_cache, void* out,
                                        const int32_t* req_to_token, int64_t req_to_token_stride,
                                        const int64_t* req_pool_indices, const int32_t* seq_lens,
                                        const int32_t* plan, int64_t batch, int64_t max_items,
                                        int num_q_heads, int num_kv_heads, int head_dim,
                                        int64_t q_token_stride, int64_t out_token_stride,
                                        int64_t k_cache_row_stride, int64_t v_cache_row_stride,
                                        float sm_scale, int64_t max_context_len, int slots_total,
                                        void* ws_acc, void* ws_ml, int kv_fp8, float k_scale, float v_scale,
                                        int page_size, int kv_layout_hnd, void* stream);

/* ---- Sampling (reference: srt/layers/sampler.py:98-260,567-750;
 *      kernels/ops/sampling/murmur_hash.py:51-121) ------------------------------- */
/* ids[b] = argmax(logits[b,:]) (first maximum), logits fp32 (is_bf16=0) or bf16. */
int sgl_amd_argmax(const void* logits, int logits_is_bf16, int64_t* ids, int64_t batch,
                   int64_t vocab, int64_t row_stride, void* stream);
/* The same ids with every row cut into num_splits column ranges (one workgroup each; a decode batch on one workgroup
 * per row leaves most CUs idle): every range's winner is a 64-bit key (order-preserving value bits, then the smaller
 * index) in the workspace, a second launch takes the row maxima (round 6: two plain launches instead of atomics + fences
 * behind the lm_head stream).  workspace: sgl_amd_argmax_split_workspace_bytes(batch) bytes, caller-owned, no initial
 * state, private to one stream at a time; rows 16-byte aligned; vocab < 2^32. */
int64_t sgl_amd_argmax_split_workspace_bytes(int64_t batch);
int sgl_amd_argmax_split(const void* logits, int logits_is_bf16, int64_t* ids, int64_t batch, int64_t vocab,
                         int64_t row_stride, int num_splits, void* workspace, void* stream);
/* in place: logits[b,:] = softmax(logits[b,:] / temperatures[b]) (fp32). */
int sgl_amd_softmax_temperature(float* logits, const float* temperatures, int64_t batch,
                                int64_t vocab, int64_t row_stride, void* stream);
/* The same for decode-sized batches of wide rows (rows cut into num_splits <= 64 column ranges over the whole chip,
 * partial (max, sum) pairs merged in range order: deterministic).  workspace:
 * sgl_amd_softmax_temperature_split_workspace_bytes(batch, num_splits) bytes, caller-owned, no initial state; rows
 * 16-byte aligned. */
int64_t sgl_amd_softmax_temperature_split_workspace_bytes(int64_t batch, int num_splits);
int sgl_amd_softmax_temperature_split(float* logits, const float* temperatures, int64_t batch, int64_t vocab,
                                      int64_t row_stride, int num_splits, void* workspace, void* stream);
/* The same from bf16 logits (the model dtype: sampler.py widens them first, `logits.float()`): probs fp32 [B, V] =
 * softmax(float(logits) / T) -- the widening is exact, the arithmetic the fp32 kernel's, without the separate pass. */
int sgl_amd_softmax_temperature_split_bf16(const void* logits_bf16, float* probs, const float* temperatures, int64_t batch, int64_t vocab,
                                           int64_t logits_row_stride, int64_t probs_row_stride, int num_splits, void* workspace,
                                           void* stream);

/* Top-k / top-p / min-p sampling with the reference's deterministic gumbel mode
 * (sampler.py:567-612 top_k_top_p_min_p_sampling_from_probs_torch + :688-729
 * multinomial_with_seed).  probs fp32 [B,V] (after softmax).  filtered=1: keep the sorted
 * prefix {rank < top_k, exclusive cumsum <= top_p, [p >= max*min_p]} and return the token whose
 * SORTED RANK j maximises log(p_j) + gumbel(murmur(seed, position, j)) in fp64.  filtered=0:
 * sampling_from_probs_torch (:732-750), no filter, column = token id.  seeds are required (for
 * unseeded sampling the caller draws fresh ones).  Nuclei larger than sgl_amd_sampling_lds_keep()
 * are ranked in the caller-owned workspaces ws_keys/ws_toks (each B*2*V 4-byte words);
 * without them such a row returns id -1.  filtered=0 uses ws_keys only (256 bytes per row suffice; ws_toks any
 * non-NULL pointer): given it, decode-sized batches are cut into column ranges over the whole chip -- same ids.
 * top_ks/top_ps/min_ps/positions/out_n_keep may be NULL. */
int sgl_amd_top_k_top_p_min_p_sample(const float* probs, int64_t row_stride, int64_t batch,
                                     int64_t vocab, const int32_t* top_ks, const float* top_ps,
                                     const float* min_ps, const int64_t* seeds,
                                     const int64_t* positions, int32_t* out_ids, void* ws_keys,
                                     void* ws_toks, int32_t* out_n_keep, int filtered, void* stream);
int sgl_amd_sampling_lds_keep(void);
/* The filtered case (filtered = 1 above) for decode-sized batches of wide rows: the two full-row passes of the single-workgroup
 * kernel -- the first radix level's histogram, the collection of the cut's candidates -- run as `num_ranges` column ranges per
 * row over the whole chip, a third launch finishes every row from its candidate list (rows the shortcut does not cover run the
 * whole routine there).  Same ids and kept counts as sgl_amd_top_k_top_p_min_p_sample.  ws_ranges: caller-owned,
 * sgl_amd_sample_ranges_workspace_bytes(batch, num_ranges) bytes, 16-byte aligned, no initial state. */
int64_t sgl_amd_sample_ranges_workspace_bytes(int64_t batch, int num_ranges);
int sgl_amd_top_k_top_p_min_p_sample_ranges(const float* probs, int64_t row_stride, int64_t batch, int64_t vocab,
                                            const int32_t* top_ks, const float* top_ps, const float* min_ps,
                                            const int64_t* seeds, const int64_t* positions, int32_t* out_ids,
                                            void* ws_keys, void* ws_toks, int32_t* out_n_keep, int num_ranges,
                                            void* ws_ranges, void* stream);
/* sampler.py:211-260 in one call for the logits of a decode-sized batch -- bf16 (the model's dtype) or fp32 (what the reference's
 * LogitsProcessor hands its Sampler: logits_processor.py `.float()`) --: softmax(logits / T) followed by the filtered
 * sampler above, without ever writing the [batch, vocab] probabilities (the fp32 logits are NOT overwritten with them).  p(x) is monotone in the logit, so every column range
 * selects its largest bf16 logits exactly (two-level radix select on 16-bit order keys) and only those candidates are turned into
 * probabilities -- with the range partials, the merge and the formula of sgl_amd_softmax_temperature_split{,_bf16} (num_splits
 * ranges: the same bits) -- ranked, filtered by the three rules and sampled.  A row whose candidates cannot decide the rules
 * (they reach the largest value a range left out: flat rows, top-p over a wide nucleus without top-k, > 127 ties, top_k <= 0, a
 * non-finite softmax) is redone inside the same call from its full probability row, written into probs_scratch for that row only.
 * Same ids and kept counts as the two calls it replaces.  num_splits: 2..16 or a multiple of 16 up to 64; ws_fast: caller-owned,
 * sgl_amd_sample_from_logits_workspace_bytes(batch, num_splits) bytes, 16-byte aligned, no initial state (its last `batch`
 * int32 words hold, after the call, which rows went the long way: tests / telemetry). */
int64_t sgl_amd_sample_from_logits_workspace_bytes(int64_t batch, int num_splits);
int sgl_amd_top_k_top_p_min_p_sample_from_logits(const void* logits, int logits_is_bf16, int64_t logits_row_stride, const float* temperatures,
                                                 float* probs_scratch, int64_t probs_row_stride, int64_t batch, int64_t vocab,
                                                 const int32_t* top_ks, const float* top_ps, const float* min_ps,
                                                 const int64_t* seeds, const int64_t* positions, int32_t* out_ids,
                                                 void* ws_keys, void* ws_toks, int32_t* out_n_keep, int num_splits,
                                                 void* ws_fast, void* stream);
/* sgl_kernel.top_k_renorm_prob / top_p_renorm_prob (kernels/aot/python/sgl_kernel/sampling.py:28,79)
 * and sampler.py:753-762 top_p_normalize_probs_torch: zero everything outside the kept sorted
 * prefix and renormalise.  Per-row arrays override the scalar values when non-NULL;
 * top_k_val < 0 disables top-k, top_p_val >= 1 disables top-p. */
int sgl_amd_top_k_top_p_renorm_probs(const float* probs, float* out, int64_t in_row_stride,
                                     int64_t out_row_stride, int64_t batch, int64_t vocab,
                                     const int32_t* top_ks, int top_k_val, const float* top_ps,
                                     float top_p_val, void* stream);

/* ---- Skinny / grouped GEMM (reference: kernels/ops/moe/fused_moe_triton_kernels.py:324,771
 *      fused_moe_kernel / invoke_fused_moe_kernel; srt/layers/linear.py:1596-1660 for the dense
 *      decode projections; srt/layers/activation.py:130 when fuse_silu=1) ------------------- */
/* y[M,N] = x[M,K] . w[N,K]^T (+ bias[N]), bf16 in/out, fp32 accumulate, M <= sgl_amd_skinny_gemm_max_rows().
 * fuse_silu=1: w is [2N,K] (gate rows then up rows) and y = silu(x.gate^T) * (x.up^T) with torch's
 * bf16 rounding points.  tiles_per_wave (1|2) = 16-column tiles each wave owns.
 * num_k_splits > 1 splits K over workgroups: partial tiles go to ws_slabs
 * (sgl_amd_skinny_gemm_slab_floats(1, N, splits, fuse_silu, tiles_per_wave) floats) and a second
 * launch sums them in split order (deterministic) and runs the epilogue. */
int sgl_amd_skinny_gemm(const void* x, const void* w, const void* bias, void* y, int64_t M, int64_t N,
                        int64_t K, int64_t x_row_stride, int64_t w_row_stride, int64_t y_row_stride,
                        int fuse_silu, int tiles_per_wave, int num_k_splits, void* ws_slabs,
                        void* stream);
int sgl_amd_skinny_gemm_max_rows(void);
int sgl_amd_skinny_gemm_chunk(void);
int64_t sgl_amd_skinny_gemm_slab_floats(int64_t row_blocks, int64_t N, int splits, int fuse_silu,
                                        int tiles_per_wave);
/* Weight-streaming GEMM for decode batches (M <= sgl_amd_wstream_gemm_max_rows()): y = x . w^T, bf16,
 * fp32 accumulate; needs N % 16 == 0 and K % 128 == 0 (other shapes: sgl_amd_skinny_gemm).
 * Replaces the library matmul of srt/layers/linear.py:1596-1660 (UnquantizedLinearMethod.apply) at decode.
 * waves_per_group (4..8; 4..5 beyond 64 rows) x tiles_per_wave x num_k_splits is the host's choice of decomposition:
 * one workgroup is resident per CU (its LDS ring holds the in-flight chunks), so ceil(N/16/tiles/waves) x splits
 * should be a whole number of 256-workgroup rounds.  tiles_per_wave = 2 (N % 32 == 0, waves_per_group 2..4): a wave
 * owns output tiles t and t + N/32, which halves the activation traffic per weight byte.  num_k_splits > 1 writes fp32 partials
 * [splits, M, N] to ws_partials (sgl_amd_wstream_gemm_workspace_floats) and a combine kernel sums
 * them in split order (deterministic) and applies `epilogue`:
 *   0: y[M,N]   = bf16(acc + bias)                         (bias may be NULL; also valid with 1 split)
 *   1: y[M,N/2] = silu_and_mul of the [gate | up] columns  (srt/layers/activation.py:141-143 rounding);
 *      with num_k_splits == 1 it runs in the GEMM's own epilogue (waves_per_group 2..4, each wave owns a
 *      gate tile and its up tile; no workspace, no second launch)
 *   2: h = bf16(acc + bias); residual += h (bf16, in place); y = RMSNorm(residual) * norm_weight
 *      (srt/layers/layernorm.py:786-820 forward_native with residual).
 * Activation layouts: x_chunk_stride / y_chunk_stride = 0 is the reference's row-major [M, K] / [M, N].  A non-zero
 * value is the "chunk-major" form a chain of these GEMMs may keep BETWEEN its own launches: element (m, n) at
 * (n / 128) * chunk_stride + m * row_stride + n % 128, e.g. [K/128][M][128] with row_stride 128 and chunk_stride
 * 128 M, so that the 128-wide K chunk of all rows a workgroup stages per step is one contiguous burst.  The
 * reduction walks K from a per-workgroup staggered starting chunk (wrapping around): the fp32 summation order is a
 * fixed function of (shape, waves_per_group, num_k_splits), not of the run. */
int sgl_amd_wstream_gemm(const void* x, const void* w, const void* bias, void* y, int64_t M, int64_t N,
                         int64_t K, int64_t x_row_stride, int64_t x_chunk_stride, int64_t w_row_stride,
                         int64_t y_row_stride, int64_t y_chunk_stride, int epilogue, void* residual,
                         int64_t residual_row_stride, const void* norm_weight, float eps, int waves_per_group,
                         int tiles_per_wave, int num_k_splits, void* ws_partials, void* stream);
/* qkv_proj + neox rotary embedding + KV-pool store for a decode batch, as one GEMM + combine pair:
 * q_out[M, Hq*D] = rope(x . w_q^T + b), k_cache[cache_loc[m]] = rope(x . w_k^T + b), v_cache[...] = x . w_v^T + b,
 * with the rounding points of QKVParallelLinear -> RotaryEmbedding.forward_native -> set_kv_buffer
 * (srt/layers/linear.py:1596, rotary_embedding/utils.py:49-57, base.py:385-417).  w_qkv is [(Hq+2Hkv)*D, K]
 * (q rows, k rows, v rows); cos_sin_cache [max_pos, D] = cos | sin halves, bf16 or fp32; ws_partials holds
 * sgl_amd_wstream_gemm_workspace_floats(M, (Hq+2Hkv)*D, num_k_splits) floats (always needed).  The pool may be any
 * format of sgl_amd_store_kv_cache_ex (kv_fp8: e4m3 of the bf16 row / scale; kv_layout_hnd: [pages, Hkv, page, D]);
 * cache_row_stride = elements of one token's [Hkv, D] row. */
int sgl_amd_wstream_qkv_rope(const void* x, const void* w_qkv, const void* bias, void* q_out, int64_t M,
                             int64_t K, int num_q_heads, int num_kv_heads, int head_dim,
                             int64_t x_row_stride, int64_t x_chunk_stride, int64_t w_row_stride, int64_t q_row_stride,
                             const int64_t* positions, const void* cos_sin_cache, int cache_is_f32,
                             int64_t rotary_dim, void* k_cache, void* v_cache, const int64_t* cache_loc,
                             int64_t cache_row_stride, int kv_fp8, float k_scale, float v_scale,
                             int page_size, int kv_layout_hnd, int waves_per_group, int tiles_per_wave,
                             int num_k_splits, void* ws_partials, void* stream);
/* Grouped (mixture-of-experts) form of the weight-streaming GEMM, same contract as sgl_amd_moe_grouped_gemm
 * (fused_moe_triton_kernels.py:324,771) for shapes with N % 16 == 0 and K % 128 == 0, without split-K: one
 * workgroup per (row block of moe_align_block_size, group of weight tiles of that block's expert).  fuse_silu=1:
 * w is [E, 2N, K] and c[id, :N] = silu_and_mul; otherwise c[id, :] = a[id / top_k_div] . w[e]^T, optionally
 * rounded to bf16 and scaled by topk_weights[id], written bf16 or fp32 (out_f32). */
int sgl_amd_wstream_moe_gemm(const void* a, const void* w, void* c, const int32_t* sorted_token_ids,
                             const int32_t* expert_ids, const int32_t* num_tokens_post_padded,
                             const float* topk_weights, int mul_routed_weight, int round_before_scale,
                             int top_k_div, int64_t num_valid_ids, int64_t N, int64_t K,
                             int64_t a_row_stride, int64_t w_row_stride, int64_t w_expert_stride,
                             int64_t c_row_stride, int block_m, int64_t max_m_blocks, int fuse_silu,
                             int out_f32, int waves_per_group, void* stream);
int sgl_amd_wstream_gemm_max_rows(void);
int64_t sgl_amd_wstream_gemm_workspace_floats(int64_t M, int64_t N, int num_k_splits);
/* Grouped GEMM over moe_align_block_size output: for every row block b < num_tokens_post_padded/block_m
 * with expert e = expert_ids[b]:  c[id, :] = a[id / top_k_div, :] . w[e]^T  for id in
 * sorted_token_ids[b*block_m : (b+1)*block_m] with id < num_valid_ids, optionally scaled by
 * topk_weights[id] (round_before_scale=1 rounds the accumulator to bf16 first, which is the
 * arithmetic of fused_moe_native.py:157-163).  out_f32=1 writes c as fp32.  Split-K workspaces as
 * above with row_blocks = max_m_blocks. */
int sgl_amd_moe_grouped_gemm(const void* a, const void* w, void* c, const int32_t* sorted_token_ids,
                             const int32_t* expert_ids, const int32_t* num_tokens_post_padded,
                             const float* topk_weights, int mul_routed_weight, int round_before_scale,
                             int top_k_div, int64_t num_valid_ids, int64_t N, int64_t K,
                             int64_t num_experts, int64_t a_row_stride, int64_t w_row_stride,
                             int64_t w_expert_stride, int64_t c_row_stride, int block_m,
                             int64_t max_m_blocks, int fuse_silu, int out_f32, int tiles_per_wave,
                             int num_k_splits, void* ws_slabs, void* stream);

/* ---- MoE routing / bookkeeping (reference: sgl_kernel.topk_softmax / moe_align_block_size /
 *      moe_sum_reduce, common_extension_rocm.cc:135-149; torch spec srt/layers/moe/topk.py:690-736) */
/* softmax over the gate logits (fp32 math), top-k (ties: lowest expert id), optional renormalise
 * w / (sum w + 1e-20).  gating [M,E] fp32 or bf16; topk_weights fp32 [M,k]; topk_ids int32 [M,k]. */
int sgl_amd_topk_softmax(const void* gating_output, int gating_is_bf16, float* topk_weights,
                         int32_t* topk_ids, int64_t num_tokens, int num_experts, int topk,
                         int64_t gating_row_stride, int renormalize, void* stream);
/* Stable counting sort of the numel = M*topk flat pair ids by expert, each expert padded to
 * block_size with the value numel.  ids may be -1 (filtered, sorted first, expert_ids = -1).
 * sorted_token_ids capacity >= numel + (num_experts+1)*(block_size-1); expert_ids gets one entry per
 * row block up to expert_capacity (-1 beyond num_tokens_post_pad). */
int sgl_amd_moe_align_block_size(const void* topk_ids, int ids_are_i64, int64_t numel, int num_experts,
                                 int block_size, int32_t* sorted_token_ids, int32_t* expert_ids,
                                 int32_t* num_tokens_post_pad, int64_t sorted_capacity,
                                 int64_t expert_capacity, void* stream);
/* out[m,:] = bf16(routed_scaling_factor * sum_k input[m,k,:]), fp32 accumulate; input bf16 or fp32. */
int sgl_amd_moe_sum_reduce(const void* input, int input_is_f32, void* output, int64_t num_tokens, int topk,
                           int hidden, int64_t in_token_stride, int64_t in_k_stride,
                           int64_t out_row_stride, float routed_scaling_factor, void* stream);

/* ---- test-only probes (used by tests/ to pin the MFMA lane maps) --------------- */
int sgl_amd_probe_mfma_16x16x32(const void* a_16x32_bf16, const void* b_32x16_bf16,
                                void* c_16x16_f32, void* stream);

/* ---- one-shot all-reduce over xGMI peer mappings (reference:
 *      kernels/aot/csrc/allreduce/custom_all_reduce_hip.cuh:150-236,347-420,448-590 and
 *      srt/distributed/device_communicators/custom_all_reduce.py:182-307, under
 *      GroupCoordinator.all_reduce, srt/distributed/parallel_state.py:648-758) -------------------------------
 * The ONE place where the library owns device memory: a communicator workspace must be its own (uncached)
 * allocation to be exportable with hipIpcGetMemHandle, so it is allocated / freed here, once per process, by the
 * caller's communicator object (sglang_amd/distributed/xgmi_all_reduce.py).  Layout: a 32 KiB signal block
 * (per-workgroup start / middle / end flags of every rank, the rank's own flag counters) + the data area.
 * Every rank opens every peer's handle; `peer_workspaces_host` is a HOST array of `world` device pointers
 * (entry `rank` = the rank's own workspace).  All ranks must issue the same sequence of calls (sizes, num_blocks).
 * The launch itself has no host state (flag counters live in the workspace): hipGraph-capturable. */
int64_t sgl_amd_xgmi_workspace_bytes(int64_t max_message_bytes);
int sgl_amd_xgmi_max_world(void);
int sgl_amd_xgmi_alloc(int64_t bytes, void** out_ptr);
int sgl_amd_xgmi_free(void* ptr);
int sgl_amd_xgmi_ipc_handle_bytes(void);
int sgl_amd_xgmi_ipc_get_handle(void* ptr, void* out_handle /* host, ipc_handle_bytes */);
int sgl_amd_xgmi_ipc_open_handle(const void* handle /* host */, void** out_ptr);
int sgl_amd_xgmi_ipc_close_handle(void* ptr);
/* 1 when a flag wait of an earlier launch gave up (a peer never arrived); synchronises. */
int sgl_amd_xgmi_timed_out(const void* workspace);
/* out[rows, hidden] = sum over ranks of inp (bf16, fp32 accumulation in rank order, one rounding); world 2/4/8.
 * epilogue 1: residual <- bf16(bf16(sum) + residual); out = RMSNorm(that, norm_weight, eps) -- the operator that
 * follows a row-parallel projection (layernorm.py:777-826).  num_blocks <= 0: chosen from the message size. */
int sgl_amd_xgmi_one_shot_all_reduce(const void* inp, void* out, int64_t rows, int hidden, int rank, int world,
                                     const void* const* peer_workspaces_host, int64_t workspace_bytes, int epilogue,
                                     void* residual, const void* norm_weight, float eps, int num_blocks, void* stream);
/* Arm the workspace after the start-up self-test: from then on a flag wait that gives up TRAPS (the stream fails and
 * every later call on the rank raises) instead of letting an unreduced sum pass for a result. */
int sgl_amd_xgmi_arm(void* workspace, int trap_on_timeout);
/* Test hook (several ranks sharing ONE GPU): cap on the automatically chosen workgroup counts of the collectives. */
int sgl_amd_xgmi_debug_auto_blocks_cap(int cap);
/* Two-stage all-reduce for prefill-sized messages (custom_all_reduce_hip.cuh:595-652; custom_all_reduce.py:260-307
 * picks it above the one-shot sizes): reduce-scatter + all-gather by pulling over the direct links, 2/world of the
 * message per link direction instead of the whole message.  Every 8 KiB chunk is summed once, by its owner rank, in
 * rank order: identical bits on all ranks.  The workspace's data area is cut in two (copies, published sums):
 * needs workspace_bytes >= 32 KiB + 4 * rows * hidden (+ padding).  epilogue 1 as in the one-shot kernel: the rows
 * are the chunks, and the workgroup that gathers a row finishes residual add + RMSNorm (decode batches of a few
 * hundred rows per rank: weak-scaled TP). */
int sgl_amd_xgmi_two_stage_all_reduce(const void* inp, void* out, int64_t rows, int hidden, int rank, int world,
                                      const void* const* peer_workspaces_host, int64_t workspace_bytes, int epilogue,
                                      void* residual, const void* norm_weight, float eps, int num_blocks, void* stream);
/* out[rows, world * cols_per_rank] = the ranks' inp[rows, cols_per_rank] side by side (the vocab-parallel logits of
 * logits_processor.py:676) -- one launch with the same flag protocol, so the decode graph holds no RCCL node. */
int sgl_amd_xgmi_all_gather(const void* inp, void* out, int64_t rows, int cols_per_rank, int rank, int world,
                            const void* const* peer_workspaces_host, int64_t workspace_bytes, int num_blocks, void* stream);
/* Protocol switch of the flag barriers, PER COMMUNICATOR: a word of `workspace`'s own signal block (this rank's workspace, as
 * returned by sgl_amd_xgmi_alloc), read by the kernels when a launch RUNS -- a captured graph follows the current setting, and
 * no process-wide state is kept in the library.  0 (default): everything a peer reads is written with system-scope write-through
 * stores and published by their completion (s_waitcnt vmcnt(0)) + a relaxed system-scope flag; 1: a full system-scope
 * RELEASE fence precedes every flag as well -- the reference's protocol (custom_all_reduce_hip.cuh:150-236
 * __atomic_store_n(..., __ATOMIC_RELEASE) after __threadfence_system), slower by the write-back of the device's
 * dirty L2, kept as the fallback should the light protocol ever misbehave across physical xGMI links.  Synchronous (a
 * 4-byte host-to-device copy); every rank of a group sets its own. */
int sgl_amd_xgmi_set_release_fence(void* workspace, int on);
/* Byte offset of the data area inside a workspace (the signal block precedes it). */
int64_t sgl_amd_xgmi_data_offset(void);
/* Host-side mirror of the two-stage kernel's work split, for tests and sizing: units_per_rank[r] = number of units
 * (8 KiB chunks, or rows with epilogue 1) rank r sums and publishes for a [rows, hidden] message; returns the
 * workgroup count used (num_blocks <= 0: the automatic one), < 0 on a bad argument.  A balanced split is what makes
 * the per-link traffic 2/world of the message. */
int sgl_amd_xgmi_two_stage_owner_units(int64_t rows, int hidden, int world, int epilogue, int num_blocks,
                                       int64_t* units_per_rank /* host, [world] */);

/* ---- pool layouts / element formats / masks beyond the bf16 NHD default ----------
 * kv_fp8 = 1: the pools hold OCP e4m3 bytes of K / k_scale and V / v_scale (memory_pool.py:2364-2374,
 * `--kv-cache-dtype fp8_e4m3`); the attention folds k_scale into the logit scale and multiplies v_scale into the
 * output (triton_backend.py:1418-1420 k_descale / v_descale).  row strides are in ELEMENTS (= bytes for fp8).
 * kv_layout_hnd = 1: pools are [pages, H_kv, page_size, D] (memory_pool.py:2061-2117), slot = page * page_size + off,
 * page_size a power of two; 0: [slots, H_kv, D].
 * sliding_window >= 0: a query at position p sees kv positions [p - window, p] (torch_native_backend.py:36-48,
 * extend_attention.py:480-485); logit_cap > 0: s <- cap * tanh(s / cap) (extend_attention.py:546-547).
 * skip_prefix_custom_mask = 1: the mask is consulted on the extend part only, the prefix stays fully visible (the
 * reference's call form for TARGET_VERIFY: extend_attention.py:774 default, :437 `not SKIP_PREFIX_CUSTOM_MASK`).
 * custom_mask (extend only; speculative-decoding verify, triton_backend.py:860-919): request b's
 * [extend_len, kv_len] row-major uint8 mask at custom_mask + mask_indptr[b] replaces the causal rule. */
int sgl_amd_store_kv_cache_ex(const void* k, const void* v, void* k_cache, void* v_cache, const int64_t* loc,
                              int64_t num_tokens, int num_kv_heads, int head_dim, int64_t k_token_stride,
                              int64_t v_token_stride, int64_t cache_row_stride, int kv_fp8, float k_scale, float v_scale,
                              int page_size, int kv_layout_hnd, void* stream);
int sgl_amd_decode_attention_ex(const void* q, const void* k_cache, const void* v_cache, void* out,
                                const int32_t* req_to_token, int64_t req_to_token_stride,
                                const int64_t* req_pool_indices, const int32_t* seq_lens,
                                const int32_t* kv_indptr, int64_t batch, int num_q_heads,
                                int num_kv_heads, int head_dim, int64_t q_token_stride,
                                int64_t out_token_stride, int64_t k_cache_row_stride,
                                int64_t v_cache_row_stride, float sm_scale, int num_splits,
                                void* ws_acc, void* ws_ml, const int32_t* batch_order, int flags,
                                int kv_fp8, float k_scale, float v_scale, int page_size, int kv_layout_hnd,
                                int sliding_window, float logit_cap, void* stream);
int sgl_amd_extend_attention_ex(const void* q, void* out, const void* k_cache, const void* v_cache,
                                const int32_t* req_to_token, int64_t req_to_token_stride,
                                const int64_t* req_pool_indices, const int32_t* seq_lens,
                                const int32_t* prefix_lens, const int32_t* qo_indptr, int64_t batch,
                                int max_extend_len, int num_q_heads, int num_kv_heads, int head_dim,
                                int64_t q_token_stride, int64_t out_token_stride,
                                int64_t k_cache_row_stride, int64_t v_cache_row_stride, float sm_scale,
                                int causal, int kv_fp8, float k_scale, float v_scale, int page_size, int kv_layout_hnd,
                                int sliding_window, float logit_cap, const void* custom_mask, const int64_t* mask_indptr, int skip_prefix_custom_mask,
                                void* stream);
/* Test / tuning override of the extend kernel's workgroup shape (process-wide; the library never reads the
 * environment): shape 0 = automatic, 41 / 42 / 82 = waves x 16-row tiles per wave (rows per workgroup 64 / 128 / 256);
 * flags bit 0 keeps bf16 8-wave launches on the general single-image kernel, bit 1 sends them to the ping-pong
 * 16x16x32 kernel the 32x32 two-score-set kernel replaced (A/B runs).  Not part of the reference surface. */
int sgl_amd_debug_extend_attention_shape(int shape, int flags);
/* Test / tuning override of the shared-prefix chunk kernel's launch form (process-wide; the library never reads the
 * environment).  A launch whose worst-case workgroup count (from the request table's width) is <= single_shot_units runs one
 * workgroup per (item, kv head) unit; above it a grid of loop_grid resident workgroups walks the device-built item list.
 * 0 restores a default (10240 / 1280).  Results are identical in both forms.  Not part of the reference surface. */
int sgl_amd_debug_cascade_launch_form(int64_t single_shot_units, int64_t loop_grid);
/* Test / tuning switch of the weight-streaming GEMM (process-wide; the library never reads the environment): bit 1 = the fp32
 * split-K partials are stored with plain write-back stores instead of write-through ones (the default since round 6:
 * profiles/r06_exp2_gemm_ab.json).  Results are identical.  flags < 0 only reads.  Returns the previous value.  Not part of the
 * reference surface. */
int sgl_amd_debug_wstream_flags(int flags);

/* ---- row-tiled grouped GEMM for prefill-sized MoE batches (reference: fused_moe_triton_kernels.py:324,771 with
 *      BLOCK_SIZE_M >= 64; fused_experts, triton_utils/fused_moe.py:242-455) -------------------------------------
 * Same contract as sgl_amd_wstream_moe_gemm, with the moe_align block size fixed at sgl_amd_moe_tiled_gemm_block_m()
 * (128): a 128-row x 128-column MFMA tile per workgroup, both operands staged in LDS, so an expert's weights are
 * read once per 128 of its rows.  fuse_silu: w is [E, 2N, K] (gate rows, then up rows), c [num_valid_ids, N] bf16 =
 * silu(gate) * up.  Needs K %% 64 == 0, N %% 4 == 0 (N %% 32 == 0 with fuse_silu). */
int sgl_amd_moe_tiled_gemm_block_m(void);
int sgl_amd_moe_tiled_gemm(const void* a, const void* w, void* c, const int32_t* sorted_token_ids, const int32_t* expert_ids,
                           const int32_t* num_tokens_post_padded, const float* topk_weights, int mul_routed_weight,
                           int round_before_scale, int top_k_div, int64_t num_valid_ids, int64_t N, int64_t K,
                           int64_t a_row_stride, int64_t w_row_stride, int64_t w_expert_stride, int64_t c_row_stride,
                           int64_t max_m_blocks, int fuse_silu, int out_f32, void* stream);
/* The same with the row geometry spelled out: `align_block_m` = the moe_align_block_size block the metadata was built with
 * (128 or 256), `tile_rows` = rows of one expert per workgroup tile (128: the form above, also over a 256-row alignment;
 * 256: the 256 x 256 x 64 form -- 8 waves, 128 x 64 outputs per wave, two 64 KiB LDS-DMA stages, one barrier per K step,
 * XCD-patched workgroup order -- for experts that own a thousand rows or more; needs align_block_m == 256). */
int sgl_amd_moe_tiled_gemm_ex(const void* a, const void* w, void* c, const int32_t* sorted_token_ids, const int32_t* expert_ids,
                              const int32_t* num_tokens_post_padded, const float* topk_weights, int mul_routed_weight,
                              int round_before_scale, int top_k_div, int64_t num_valid_ids, int64_t N, int64_t K,
                              int64_t a_row_stride, int64_t w_row_stride, int64_t w_expert_stride, int64_t c_row_stride,
                              int64_t max_m_blocks, int fuse_silu, int out_f32, int align_block_m, int tile_rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SGLANG_AMD_H_ */

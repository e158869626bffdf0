// Integer host-metadata kernels of the RadixAttention path (bit-exact work).
//
// Replaces (reference, /root/reference/python/sglang):
//   kernels/ops/attention/utils.py create_flashinfer_kv_indices_triton   (K1)
//   kernels/ops/memory/allocator.py alloc_extend_kernel / alloc_decode_kernel (K14)
//   srt/mem_cache/allocation.py:73-82 write_req_to_token_pool_triton,
//     :106-148 get_last_loc                                                (K15)
//   srt/model_executor/forward_batch_info.py:1790-1808 compute_position /
//     clamp_position                                                       (K15)
// Specs followed: srt/mem_cache/allocator/paged.py:45-102 (alloc_extend_naive),
// allocation.py:85-103 (CPU write loop), :139-148 (get_last_loc_torch).
//
// All of these are tiny (one workgroup per request, O(tokens) int traffic);
// they exist so a step never needs a device->host sync, and so the decode
// step can be captured in a hipGraph with static buffers.
#include "common.hpp"
#include "sglang_amd.h"

using namespace sgl_amd;

namespace {

// Exclusive prefix over [0, pid) of f(i), computed redundantly by every block
// (batch sizes are small; this mirrors the reference kernels' masked loads).
template <typename F>
__device__ __forceinline__ int64_t block_prefix(int pid, F f, int64_t* scratch) {
  int64_t s = 0;
  for (int i = threadIdx.x; i < pid; i += blockDim.x) s += f(i);
  // block reduce (int64)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) scratch[wid] = s;
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  int64_t t = 0;
  for (int w = 0; w < nw; ++w) t += scratch[w];
  __syncthreads();
  return t;
}

template <typename OutT>
__global__ void kv_indices_kernel(const int32_t* __restrict__ req_to_token,
                                  const int64_t* __restrict__ req_pool_indices64,
                                  const int32_t* __restrict__ req_pool_indices32,
                                  const int32_t* __restrict__ kernel_lens,
                                  const int32_t* __restrict__ kv_indptr,
                                  const int32_t* __restrict__ kv_start_idx,
                                  OutT* __restrict__ kv_indices, int64_t r2t_stride) {
  const int b = blockIdx.x;
  const int64_t req = req_pool_indices64 ? req_pool_indices64[b] : req_pool_indices32[b];
  const int start = kv_start_idx ? kv_start_idx[b] : 0;
  const int len = kernel_lens[b];
  const int64_t out0 = kv_indptr[b];
  const int32_t* src = req_to_token + req * r2t_stride + start;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < len; i += gridDim.y * blockDim.x)
    kv_indices[out0 + i] = static_cast<OutT>(src[i]);
}

// req_to_token[req, 0:prefix] = prefix_slots ; req_to_token[req, prefix:seq] = out_cache_loc[...]
__global__ void write_req_to_token_kernel(int32_t* __restrict__ req_to_token,
                                          const int64_t* __restrict__ req_pool_indices,
                                          const int64_t* const* __restrict__ prefix_ptrs,
                                          const int64_t* __restrict__ prefix_lens,
                                          const int64_t* __restrict__ seq_lens,
                                          const int64_t* __restrict__ extend_lens,
                                          const int64_t* __restrict__ out_cache_loc,
                                          int64_t r2t_stride) {
  __shared__ int64_t scratch[16];
  const int pid = blockIdx.x;
  const int64_t start = block_prefix(pid, [&](int i) { return extend_lens[i]; }, scratch);
  const int64_t pre = prefix_lens[pid];
  const int64_t seq = seq_lens[pid];
  int32_t* row = req_to_token + req_pool_indices[pid] * r2t_stride;
  const int64_t* pp = prefix_ptrs ? prefix_ptrs[pid] : nullptr;
  if (pp)
    for (int64_t i = threadIdx.x; i < pre; i += blockDim.x) row[i] = static_cast<int32_t>(pp[i]);
  for (int64_t i = threadIdx.x; i < seq - pre; i += blockDim.x)
    row[pre + i] = static_cast<int32_t>(out_cache_loc[start + i]);
}

// One decode step's bookkeeping for page_size 1 (schedule_batch.py prepare_for_decode + allocation.py:512-560
// alloc_for_decode): request b takes slot new_slots[b], which is written to its req_to_token row at column seq_lens[b]
// and to out_cache_loc[b]; seq_lens[b] += 1.  Five eager tensor ops (two casts, an index_put, an add, a copy) in one
// launch: the decode loop spends ~4.5 us per eager launch between two graph replays.
__global__ void decode_advance_kernel(int32_t* __restrict__ req_to_token, int64_t r2t_stride,
                                      const int64_t* __restrict__ req_pool_indices, int32_t* __restrict__ seq_lens,
                                      const int64_t* __restrict__ new_slots, int64_t* __restrict__ out_cache_loc, int64_t n) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const int32_t len = seq_lens[i];
  const int64_t slot = new_slots[i];
  req_to_token[req_pool_indices[i] * r2t_stride + len] = static_cast<int32_t>(slot);
  out_cache_loc[i] = slot;
  seq_lens[i] = len + 1;
}

__global__ void get_last_loc_kernel(const int32_t* __restrict__ req_to_token,
                                    const int64_t* __restrict__ req_pool_indices,
                                    const int64_t* __restrict__ prefix_lens,
                                    int64_t* __restrict__ out, int64_t n, int64_t r2t_stride) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const int64_t pre = prefix_lens[i];
  out[i] = pre > 0 ? static_cast<int64_t>(req_to_token[req_pool_indices[i] * r2t_stride + pre - 1])
                   : -1;
}

// positions[start_b + j] = prefix_b + j ; extend_start_loc[b] = sum_{i<b} extend_len_i
template <typename LenT>
__global__ void compute_position_kernel(const LenT* __restrict__ prefix_lens,
                                        const LenT* __restrict__ extend_lens,
                                        int64_t* __restrict__ positions,
                                        LenT* __restrict__ extend_start_loc) {
  __shared__ int64_t scratch[16];
  const int pid = blockIdx.x;
  const int64_t start = block_prefix(pid, [&](int i) { return static_cast<int64_t>(extend_lens[i]); }, scratch);
  const int64_t pre = prefix_lens[pid];
  const int64_t ext = extend_lens[pid];
  for (int64_t j = threadIdx.x; j < ext; j += blockDim.x) positions[start + j] = pre + j;
  if (threadIdx.x == 0) extend_start_loc[pid] = static_cast<LenT>(start);
}

template <typename LenT>
__global__ void clamp_position_kernel(const LenT* __restrict__ seq_lens, int64_t* __restrict__ out,
                                      int64_t n) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const int64_t v = static_cast<int64_t>(seq_lens[i]) - 1;
  out[i] = v < 0 ? 0 : v;
}

// paged.py:45-102 (alloc_extend_naive) / allocator.py alloc_extend_kernel
__global__ void alloc_extend_kernel(const int64_t* __restrict__ prefix_lens,
                                    const int64_t* __restrict__ seq_lens,
                                    const int64_t* __restrict__ last_loc,
                                    const int64_t* __restrict__ free_pages,
                                    int64_t* __restrict__ out_indices, int64_t page_size) {
  __shared__ int64_t scratch[16];
  const int pid = blockIdx.x;
  const int64_t ps = page_size;
  const int64_t out_start =
      block_prefix(pid, [&](int i) { return seq_lens[i] - prefix_lens[i]; }, scratch);
  const int64_t page_start = block_prefix(
      pid, [&](int i) { return (seq_lens[i] + ps - 1) / ps - (prefix_lens[i] + ps - 1) / ps; },
      scratch);
  const int64_t pre = prefix_lens[pid], seq = seq_lens[pid];
  const int64_t n_new_pages = (seq + ps - 1) / ps - (pre + ps - 1) / ps;
  const int64_t pre_up = (pre + ps - 1) / ps * ps;
  // part 1: fill the partially used last page of the prefix
  const int64_t num1 = (seq < pre_up ? seq : pre_up) - pre;
  const int64_t ll = last_loc[pid];
  for (int64_t i = threadIdx.x; i < num1; i += blockDim.x) out_indices[out_start + i] = ll + 1 + i;
  if (pre + num1 == seq) return;
  // part 2: whole new pages
  const int64_t num2 = seq / ps * ps - pre_up;
  for (int64_t i = threadIdx.x; i < num2; i += blockDim.x)
    out_indices[out_start + num1 + i] = free_pages[page_start + i / ps] * ps + i % ps;
  if (pre + num1 + num2 == seq) return;
  // part 3: the new partial page
  const int64_t num3 = seq - seq / ps * ps;
  const int64_t pg = free_pages[page_start + n_new_pages - 1];
  for (int64_t i = threadIdx.x; i < num3; i += blockDim.x)
    out_indices[out_start + num1 + num2 + i] = pg * ps + i;
}

// allocator.py alloc_decode_kernel: seq_lens are the lengths AFTER the new token.
__global__ void alloc_decode_kernel(const int64_t* __restrict__ seq_lens,
                                    const int64_t* __restrict__ last_loc,
                                    const int64_t* __restrict__ free_pages,
                                    int64_t* __restrict__ out_indices, int64_t page_size) {
  __shared__ int64_t scratch[16];
  const int pid = blockIdx.x;
  const int64_t ps = page_size;
  auto new_pages = [&](int i) {
    const int64_t s = seq_lens[i];
    return (s + ps - 1) / ps - (s - 1 + ps - 1) / ps;
  };
  const int64_t page_start = block_prefix(pid, new_pages, scratch);
  if (threadIdx.x != 0) return;
  if (new_pages(pid) == 0) out_indices[pid] = last_loc[pid] + 1;
  else out_indices[pid] = free_pages[page_start] * ps;
}

}  // namespace

extern "C" {

int sgl_amd_create_kv_indices(const int32_t* req_to_token, int64_t req_to_token_stride,
                              const void* req_pool_indices, int req_pool_indices_is_i64,
                              const int32_t* kernel_lens, const int32_t* kv_indptr,
                              const int32_t* kv_start_idx, void* kv_indices, int kv_indices_is_i64,
                              int64_t batch, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(batch <= 0x7fffffffLL, "create_kv_indices: batch too large");
  if (batch == 0) return 0;
  const int64_t* r64 = req_pool_indices_is_i64 ? static_cast<const int64_t*>(req_pool_indices) : nullptr;
  const int32_t* r32 = req_pool_indices_is_i64 ? nullptr : static_cast<const int32_t*>(req_pool_indices);
  dim3 grid(batch, 4);
  if (kv_indices_is_i64)
    hipLaunchKernelGGL(kv_indices_kernel<int64_t>, grid, dim3(256), 0, as_stream(stream),
                       req_to_token, r64, r32, kernel_lens, kv_indptr, kv_start_idx,
                       static_cast<int64_t*>(kv_indices), req_to_token_stride);
  else
    hipLaunchKernelGGL(kv_indices_kernel<int32_t>, grid, dim3(256), 0, as_stream(stream),
                       req_to_token, r64, r32, kernel_lens, kv_indptr, kv_start_idx,
                       static_cast<int32_t*>(kv_indices), req_to_token_stride);
  SGL_CHECK_LAUNCH("create_kv_indices");
  return 0;
}

int sgl_amd_decode_advance(int32_t* req_to_token, int64_t req_to_token_stride, const int64_t* req_pool_indices,
                           int32_t* seq_lens, const int64_t* new_slots, int64_t* out_cache_loc, int64_t batch,
                           void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(req_to_token && req_pool_indices && seq_lens && new_slots && out_cache_loc, "decode_advance: null argument");
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(decode_advance_kernel, dim3(static_cast<unsigned>((batch + 255) / 256)), dim3(256), 0, as_stream(stream),
                     req_to_token, req_to_token_stride, req_pool_indices, seq_lens, new_slots, out_cache_loc, batch);
  SGL_CHECK_LAUNCH("decode_advance");
  return 0;
}

int sgl_amd_write_req_to_token(int32_t* req_to_token, int64_t req_to_token_stride,
                               const int64_t* req_pool_indices, const void* prefix_ptrs,
                               const int64_t* prefix_lens, const int64_t* seq_lens,
                               const int64_t* extend_lens, const int64_t* out_cache_loc,
                               int64_t batch, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(batch <= 0x7fffffffLL, "write_req_to_token: batch too large");
  if (batch == 0) return 0;
  hipLaunchKernelGGL(write_req_to_token_kernel, dim3(batch), dim3(256), 0, as_stream(stream),
                     req_to_token, req_pool_indices,
                     static_cast<const int64_t* const*>(prefix_ptrs), prefix_lens, seq_lens,
                     extend_lens, out_cache_loc, req_to_token_stride);
  SGL_CHECK_LAUNCH("write_req_to_token");
  return 0;
}

int sgl_amd_get_last_loc(const int32_t* req_to_token, int64_t req_to_token_stride,
                         const int64_t* req_pool_indices, const int64_t* prefix_lens,
                         int64_t* last_loc, int64_t batch, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  if (batch == 0) return 0;
  hipLaunchKernelGGL(get_last_loc_kernel, dim3((batch + 255) / 256), dim3(256), 0,
                     as_stream(stream), req_to_token, req_pool_indices, prefix_lens, last_loc,
                     batch, req_to_token_stride);
  SGL_CHECK_LAUNCH("get_last_loc");
  return 0;
}

int sgl_amd_compute_position(const void* extend_prefix_lens, const void* extend_seq_lens,
                             int lens_are_i64, int64_t* positions, void* extend_start_loc,
                             int64_t batch, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(batch <= 0x7fffffffLL, "compute_position: batch too large");
  if (batch == 0) return 0;
  if (lens_are_i64)
    hipLaunchKernelGGL(compute_position_kernel<int64_t>, dim3(batch), dim3(256), 0,
                       as_stream(stream), static_cast<const int64_t*>(extend_prefix_lens),
                       static_cast<const int64_t*>(extend_seq_lens), positions,
                       static_cast<int64_t*>(extend_start_loc));
  else
    hipLaunchKernelGGL(compute_position_kernel<int32_t>, dim3(batch), dim3(256), 0,
                       as_stream(stream), static_cast<const int32_t*>(extend_prefix_lens),
                       static_cast<const int32_t*>(extend_seq_lens), positions,
                       static_cast<int32_t*>(extend_start_loc));
  SGL_CHECK_LAUNCH("compute_position");
  return 0;
}

int sgl_amd_clamp_position(const void* seq_lens, int lens_are_i64, int64_t* positions,
                           int64_t batch, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  if (batch == 0) return 0;
  const dim3 grid((batch + 255) / 256);
  if (lens_are_i64)
    hipLaunchKernelGGL(clamp_position_kernel<int64_t>, grid, dim3(256), 0, as_stream(stream),
                       static_cast<const int64_t*>(seq_lens), positions, batch);
  else
    hipLaunchKernelGGL(clamp_position_kernel<int32_t>, grid, dim3(256), 0, as_stream(stream),
                       static_cast<const int32_t*>(seq_lens), positions, batch);
  SGL_CHECK_LAUNCH("clamp_position");
  return 0;
}

int sgl_amd_alloc_extend(const int64_t* prefix_lens, const int64_t* seq_lens,
                         const int64_t* last_loc, const int64_t* free_pages, int64_t* out_indices,
                         int64_t batch, int64_t page_size, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(page_size >= 1, "alloc_extend: bad page_size");
  SGL_CHECK_ARG(batch <= 0x7fffffffLL, "alloc_extend: batch too large");
  if (batch == 0) return 0;
  hipLaunchKernelGGL(alloc_extend_kernel, dim3(batch), dim3(256), 0, as_stream(stream),
                     prefix_lens, seq_lens, last_loc, free_pages, out_indices, page_size);
  SGL_CHECK_LAUNCH("alloc_extend");
  return 0;
}

int sgl_amd_alloc_decode(const int64_t* seq_lens, const int64_t* last_loc,
                         const int64_t* free_pages, int64_t* out_indices, int64_t batch,
                         int64_t page_size, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(page_size >= 1, "alloc_decode: bad page_size");
  SGL_CHECK_ARG(batch <= 0x7fffffffLL, "alloc_decode: batch too large");
  if (batch == 0) return 0;
  hipLaunchKernelGGL(alloc_decode_kernel, dim3(batch), dim3(64), 0, as_stream(stream), seq_lens,
                     last_loc, free_pages, out_indices, page_size);
  SGL_CHECK_LAUNCH("alloc_decode");
  return 0;
}

}  // extern "C"

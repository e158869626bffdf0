// Mixture-of-experts routing / bookkeeping kernels for gfx950.
//
// Replaces (reference, /root/reference/python/sglang):
//   kernels/aot/csrc/moe/moe_topk_softmax_kernels.cu (sgl_kernel.topk_softmax; torch spec
//     srt/layers/moe/topk.py:690-736 fused_topk_torch_native),
//   kernels/aot/csrc/moe/moe_align_kernel.cu (sgl_kernel.moe_align_block_size; integer spec
//     test/registered/kernels/ops/moe/test_moe_align_block_size.py:23-140),
//   kernels/aot/csrc/moe/moe_sum_reduce.cu (sgl_kernel.moe_sum_reduce;
//     kernels/ops/moe/fused_moe_triton_kernels.py:1165 _moe_sum_reduce_kernel).
// The grouped GEMM itself lives in skinny_gemm.hip.
#include "common.hpp"
#include "sglang_amd.h"

using namespace sgl_amd;

namespace {

// ---- router: softmax -> top-k -> renormalise -------------------------------------------
// One wave per token; lane l owns experts l, l+64, ... (E <= 256).
template <bool BF16>
__global__ __launch_bounds__(256) void topk_softmax_kernel(const void* __restrict__ gating,
                                                            float* __restrict__ topk_weights,
                                                            int32_t* __restrict__ topk_ids, int64_t M, int E,
                                                            int topk, int64_t row_stride, int renormalize) {
  const int lane = threadIdx.x & 63;
  const int64_t tok = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (tok >= M) return;
  float v[4];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = lane + 64 * j;
    float x = -INFINITY;
    if (e < E) {
      x = BF16 ? bf2f(static_cast<const uint16_t*>(gating)[tok * row_stride + e])
               : static_cast<const float*>(gating)[tok * row_stride + e];
    }
    v[j] = x;
    mx = fmaxf(mx, x);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = (lane + 64 * j < E) ? expf(v[j] - mx) : 0.f;
    sum += v[j];
  }
  sum = wave_sum(sum);
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = v[j] / sum;   // torch.softmax: exp(x - max) / sum

  // k rounds of (max value, lowest expert id) selection; picked entries drop to -1
  float wsum = 0.f;
  float my_w = 0.f;       // lane i < topk keeps the i-th pick
  int my_id = 0;
  for (int i = 0; i < topk; ++i) {
    float bv = -1.f;
    int be = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = lane + 64 * j;
      if (e < E && (v[j] > bv)) { bv = v[j]; be = e; }   // ascending e per lane: first max kept
    }
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const int oe = __shfl_xor(be, off, 64);
      if (ov > bv || (ov == bv && oe < be)) { bv = ov; be = oe; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (be == lane + 64 * j) v[j] = -1.f;
    if (lane == i) { my_w = bv; my_id = be; }
    wsum += bv;
  }
  if (lane < topk) {
    float w = my_w;
    if (renormalize) w = w / (wsum + 1e-20f);      // topk.py: _RENORMALIZE_SUM_EPSILON
    topk_weights[tok * topk + lane] = w;
    topk_ids[tok * topk + lane] = my_id;
  }
}

// ---- moe_align_block_size: stable counting sort of the (token, k) pairs by expert ---------
constexpr int kAlignThreads = 1024;
constexpr int kAlignWaves = kAlignThreads / 64;
constexpr int kMaxBuckets = 512;   // experts + 1 (bucket 0 = filtered pairs, id -1)

template <bool I64>
__global__ __launch_bounds__(kAlignThreads) void moe_align_kernel(const void* __restrict__ topk_ids, int64_t numel,
                                                                   int num_buckets, int block_size,
                                                                   int32_t* __restrict__ sorted_ids,
                                                                   int32_t* __restrict__ expert_ids,
                                                                   int32_t* __restrict__ num_post_pad,
                                                                   int64_t sorted_capacity, int64_t expert_capacity) {
  __shared__ int cnt[kMaxBuckets];
  __shared__ int base[kMaxBuckets + 1];
  __shared__ int wave_cnt[kAlignWaves][kMaxBuckets];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  auto bucket_of = [&](int64_t i) -> int {
    const int64_t e = I64 ? static_cast<const int64_t*>(topk_ids)[i] : static_cast<const int32_t*>(topk_ids)[i];
    return static_cast<int>(e) + 1;
  };
  for (int b = tid; b < num_buckets; b += kAlignThreads) cnt[b] = 0;
  // pad value everywhere first (pad_sorted_token_ids=True in the reference call)
  for (int64_t i = tid; i < sorted_capacity; i += kAlignThreads) sorted_ids[i] = static_cast<int32_t>(numel);
  __syncthreads();
  for (int64_t i = tid; i < numel; i += kAlignThreads) atomicAdd(&cnt[bucket_of(i)], 1);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int b = 0; b < num_buckets; ++b) {
      base[b] = run;
      run += (cnt[b] + block_size - 1) / block_size * block_size;
    }
    base[num_buckets] = run;
    num_post_pad[0] = run;
  }
  __syncthreads();
  // expert id of every row block (bucket - 1; -1 for the filtered bucket); unused blocks get -1
  for (int64_t blk = tid; blk < expert_capacity; blk += kAlignThreads) {
    const int64_t pos = blk * block_size;
    int e = -1;
    if (pos < base[num_buckets]) {
      int lo = 0, hi = num_buckets;     // largest b with base[b] <= pos
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (base[mid] <= pos) lo = mid; else hi = mid;
      }
      e = lo - 1;
    }
    expert_ids[blk] = e;
  }
  // stable placement, one tile of 1024 pairs at a time
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int64_t i0 = 0; i0 < numel; i0 += kAlignThreads) {
    const int64_t i = i0 + tid;
    const bool act = i < numel;
    const int b = act ? bucket_of(i) : 0;
    for (int z = tid; z < kAlignWaves * num_buckets; z += kAlignThreads) wave_cnt[z / num_buckets][z % num_buckets] = 0;
    unsigned long long m = __ballot(act);
#pragma unroll
    for (int bit = 0; bit < 9; ++bit) {
      const unsigned long long bb = __ballot((b >> bit) & 1);
      m &= ((b >> bit) & 1) ? bb : ~bb;
    }
    __syncthreads();
    if (act && (m & lt_mask) == 0ull) wave_cnt[wid][b] = __popcll(m);
    __syncthreads();
    if (act) {
      int off = base[b];
      for (int w = 0; w < wid; ++w) off += wave_cnt[w][b];
      sorted_ids[off + __popcll(m & lt_mask)] = static_cast<int32_t>(i);
    }
    __syncthreads();
    for (int bb = tid; bb < num_buckets; bb += kAlignThreads) {
      int tot = 0;
      for (int w = 0; w < kAlignWaves; ++w) tot += wave_cnt[w][bb];
      base[bb] += tot;
    }
    __syncthreads();
  }
}

// ---- moe_sum_reduce: out[m, :] = scale * sum_k in[m, k, :] ----------------------------------
template <bool IN_F32>
__global__ __launch_bounds__(256) void moe_sum_reduce_kernel(const void* __restrict__ in, uint16_t* __restrict__ out,
                                                              int64_t M, int topk, int H, int64_t in_tok_stride,
                                                              int64_t in_k_stride, int64_t out_stride, float scale) {
  const int hblocks = (H / 8 + 255) / 256;
  const int64_t m = blockIdx.x / hblocks;
  const int h0 = ((blockIdx.x % hblocks) * 256 + threadIdx.x) * 8;
  if (h0 >= H) return;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int k = 0; k < topk; ++k) {
    if (IN_F32) {
      const float* p = static_cast<const float*>(in) + m * in_tok_stride + k * in_k_stride + h0;
      const float4 a = *reinterpret_cast<const float4*>(p);
      const float4 b = *reinterpret_cast<const float4*>(p + 4);
      acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
      acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
    } else {
      const U4 v = ld16(static_cast<const uint16_t*>(in) + m * in_tok_stride + k * in_k_stride + h0);
      acc[0] += bf_lo(v.x); acc[1] += bf_hi(v.x); acc[2] += bf_lo(v.y); acc[3] += bf_hi(v.y);
      acc[4] += bf_lo(v.z); acc[5] += bf_hi(v.z); acc[6] += bf_lo(v.w); acc[7] += bf_hi(v.w);
    }
  }
  U4 o;
  o.x = pack_bf2(acc[0] * scale, acc[1] * scale);
  o.y = pack_bf2(acc[2] * scale, acc[3] * scale);
  o.z = pack_bf2(acc[4] * scale, acc[5] * scale);
  o.w = pack_bf2(acc[6] * scale, acc[7] * scale);
  st16(out + m * out_stride + h0, o);
}

}  // namespace

extern "C" {

int sgl_amd_topk_softmax(const void* gating_output, int gating_is_bf16, float* topk_weights, int32_t* topk_ids,
                         int64_t num_tokens, int num_experts, int topk, int64_t gating_row_stride,
                         int renormalize, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(num_experts >= 1 && num_experts <= 256, "topk_softmax: num_experts=%d (supported: 1..256)", num_experts);
  SGL_CHECK_ARG(topk >= 1 && topk <= num_experts && topk <= 64, "topk_softmax: bad topk=%d", topk);
  if (num_tokens == 0) return 0;
  const dim3 grid(static_cast<unsigned>((num_tokens + 3) / 4));
  if (gating_is_bf16)
    hipLaunchKernelGGL(topk_softmax_kernel<true>, grid, dim3(256), 0, as_stream(stream), gating_output, topk_weights,
                       topk_ids, num_tokens, num_experts, topk, gating_row_stride, renormalize);
  else
    hipLaunchKernelGGL(topk_softmax_kernel<false>, grid, dim3(256), 0, as_stream(stream), gating_output, topk_weights,
                       topk_ids, num_tokens, num_experts, topk, gating_row_stride, renormalize);
  SGL_CHECK_LAUNCH("topk_softmax");
  return 0;
}

int sgl_amd_moe_align_block_size(const void* topk_ids, int ids_are_i64, int64_t numel, int num_experts,
                                 int block_size, int32_t* sorted_token_ids, int32_t* expert_ids,
                                 int32_t* num_tokens_post_pad, int64_t sorted_capacity, int64_t expert_capacity,
                                 void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(num_experts >= 1 && num_experts + 1 <= kMaxBuckets, "moe_align_block_size: num_experts=%d (supported: < %d)",
                num_experts, kMaxBuckets);
  SGL_CHECK_ARG(block_size >= 1, "moe_align_block_size: bad block_size");
  SGL_CHECK_ARG(numel >= 0 && numel < 0x7fffffffLL, "moe_align_block_size: bad numel");
  const int64_t worst = numel + static_cast<int64_t>(num_experts + 1) * (block_size - 1);
  SGL_CHECK_ARG(sorted_capacity >= (numel < num_experts + 1 ? numel * block_size : worst) || sorted_capacity >= worst,
                "moe_align_block_size: sorted_token_ids too small");
  if (ids_are_i64)
    hipLaunchKernelGGL(moe_align_kernel<true>, dim3(1), dim3(kAlignThreads), 0, as_stream(stream), topk_ids, numel,
                       num_experts + 1, block_size, sorted_token_ids, expert_ids, num_tokens_post_pad, sorted_capacity,
                       expert_capacity);
  else
    hipLaunchKernelGGL(moe_align_kernel<false>, dim3(1), dim3(kAlignThreads), 0, as_stream(stream), topk_ids, numel,
                       num_experts + 1, block_size, sorted_token_ids, expert_ids, num_tokens_post_pad, sorted_capacity,
                       expert_capacity);
  SGL_CHECK_LAUNCH("moe_align_block_size");
  return 0;
}

int sgl_amd_moe_sum_reduce(const void* input, int input_is_f32, void* output, int64_t num_tokens, int topk,
                           int hidden, int64_t in_token_stride, int64_t in_k_stride, int64_t out_row_stride,
                           float routed_scaling_factor, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(hidden > 0 && hidden % 8 == 0, "moe_sum_reduce: hidden=%d must be a positive multiple of 8", hidden);
  SGL_CHECK_ARG(in_token_stride % 8 == 0 && in_k_stride % 8 == 0 && out_row_stride % 8 == 0,
                "moe_sum_reduce: strides must be multiples of 8 elements");
  const int64_t hblocks = (hidden / 8 + 255) / 256;
  SGL_CHECK_ARG(num_tokens * hblocks <= 0x7fffffffLL, "moe_sum_reduce: too many tokens per launch (%lld)", (long long)num_tokens);
  if (num_tokens == 0) return 0;
  const dim3 grid(static_cast<unsigned>(num_tokens * hblocks));
  if (input_is_f32)
    hipLaunchKernelGGL(moe_sum_reduce_kernel<true>, grid, dim3(256), 0, as_stream(stream), input,
                       static_cast<uint16_t*>(output), num_tokens, topk, hidden, in_token_stride, in_k_stride,
                       out_row_stride, routed_scaling_factor);
  else
    hipLaunchKernelGGL(moe_sum_reduce_kernel<false>, grid, dim3(256), 0, as_stream(stream), input,
                       static_cast<uint16_t*>(output), num_tokens, topk, hidden, in_token_stride, in_k_stride,
                       out_row_stride, routed_scaling_factor);
  SGL_CHECK_LAUNCH("moe_sum_reduce");
  return 0;
}

}  // extern "C"

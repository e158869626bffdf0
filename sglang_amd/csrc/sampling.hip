// Sampling kernels for gfx950: greedy argmax, temperature softmax, and the
// top-k / top-p / min-p sampler with the reference's deterministic
// (murmur-hash gumbel) mode.
//
// Replaces (reference, /root/reference/python/sglang):
//   srt/layers/sampler.py:133-141 (greedy argmax), :211-216 (div_ + softmax),
//   :567-612 top_k_top_p_min_p_sampling_from_probs_torch,
//   :688-729 multinomial_with_seed, :732-750 sampling_from_probs_torch,
//   kernels/ops/sampling/murmur_hash.py:51-121 murmur_hash32.
//
// One workgroup of 1024 threads per row; a row (vocab x fp32, ~0.5 MB) stays in
// L2 between the passes, so the passes cost L2 bandwidth, not HBM.
#include "common.hpp"
#include "sglang_amd.h"

using namespace sgl_amd;

namespace {

constexpr int kRowThreads = 1024;

template <bool BF16>
__device__ __forceinline__ float load_logit(const void* base, int64_t i) {
  if (BF16) return bf2f(static_cast<const uint16_t*>(base)[i]);
  return static_cast<const float*>(base)[i];
}

// torch.argmax semantics: first index of the maximum; NaN counts as maximal.
template <bool BF16>
__global__ __launch_bounds__(kRowThreads) void argmax_kernel(const void* __restrict__ logits,
                                                              int64_t* __restrict__ ids,
                                                              int64_t vocab, int64_t row_stride) {
  __shared__ float s_val[16];
  __shared__ int64_t s_idx[16];
  const int64_t row = blockIdx.x;
  const void* base = BF16 ? static_cast<const void*>(static_cast<const uint16_t*>(logits) + row * row_stride)
                          : static_cast<const void*>(static_cast<const float*>(logits) + row * row_stride);
  float best = -INFINITY;
  int64_t best_i = INT64_MAX;
  bool best_nan = false;
  auto take = [&](float v, int64_t i) {
    const bool is_nan = v != v;
    // strictly-greater keeps the first index inside a thread (indices ascend)
    if (!best_nan && (is_nan || v > best || best_i == INT64_MAX)) {
      best = v; best_i = i; best_nan = is_nan;
    }
  };
  // 16-byte loads (8 bf16 / 4 fp32 per lane) over the aligned body, scalar head and tail
  constexpr int EPV = BF16 ? 8 : 4;
  constexpr int ESZ = BF16 ? 2 : 4;
  const uintptr_t addr = reinterpret_cast<uintptr_t>(base);
  int64_t head = ((16 - (addr & 15)) & 15) / ESZ;
  if (head > vocab) head = vocab;
  const int64_t nvec = (vocab - head) / EPV;
  for (int64_t i = threadIdx.x; i < head; i += blockDim.x) take(load_logit<BF16>(base, i), i);
  const U4* vbase = reinterpret_cast<const U4*>(static_cast<const char*>(base) + head * ESZ);
  for (int64_t j = threadIdx.x; j < nvec; j += blockDim.x) {
    const U4 v = vbase[j];
    const int64_t i0 = head + j * EPV;
    if (BF16) {
      take(bf_lo(v.x), i0 + 0); take(bf_hi(v.x), i0 + 1); take(bf_lo(v.y), i0 + 2); take(bf_hi(v.y), i0 + 3);
      take(bf_lo(v.z), i0 + 4); take(bf_hi(v.z), i0 + 5); take(bf_lo(v.w), i0 + 6); take(bf_hi(v.w), i0 + 7);
    } else {
      take(__uint_as_float(v.x), i0 + 0); take(__uint_as_float(v.y), i0 + 1);
      take(__uint_as_float(v.z), i0 + 2); take(__uint_as_float(v.w), i0 + 3);
    }
  }
  for (int64_t i = head + nvec * EPV + threadIdx.x; i < vocab; i += blockDim.x) take(load_logit<BF16>(base, i), i);
  auto better = [](float av, int64_t ai, float bv, int64_t bi) {
    // true if (bv, bi) should replace (av, ai)
    if (bi == INT64_MAX) return false;
    if (ai == INT64_MAX) return true;
    const bool an = av != av, bn = bv != bv;
    if (an != bn) return bn;
    if (!an && bv != av) return bv > av;
    return bi < ai;
  };
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int64_t oi = __shfl_xor(best_i, off, 64);
    if (better(best, best_i, ov, oi)) { best = ov; best_i = oi; }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { s_val[wid] = best; s_idx[wid] = best_i; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 1; w < nw; ++w)
      if (better(best, best_i, s_val[w], s_idx[w])) { best = s_val[w]; best_i = s_idx[w]; }
    ids[row] = best_i == INT64_MAX ? 0 : best_i;
  }
}

// logits <- softmax(logits / T), fp32, in place (sampler.py:211-216).
__global__ __launch_bounds__(kRowThreads) void softmax_temperature_kernel(
    float* __restrict__ logits, const float* __restrict__ temperatures, int64_t vocab,
    int64_t row_stride) {
  __shared__ float scratch[16];
  const int64_t row = blockIdx.x;
  float* x = logits + row * row_stride;
  const float t = temperatures[row];
  float mx = -INFINITY;
  for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) mx = fmaxf(mx, x[i] / t);
  mx = block_max(mx, scratch);
  float sum = 0.f;
  for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) sum += expf(x[i] / t - mx);
  sum = block_sum(sum, scratch);
  for (int64_t i = threadIdx.x; i < vocab; i += blockDim.x) x[i] = expf(x[i] / t - mx) / sum;
}

}  // namespace

extern "C" {

int sgl_amd_argmax(const void* logits, int logits_is_bf16, int64_t* ids, int64_t batch,
                   int64_t vocab, int64_t row_stride, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(vocab > 0, "argmax: vocab must be positive");
  SGL_CHECK_ARG(batch <= 0x7fffffffLL, "argmax: batch too large");
  if (batch == 0) return 0;
  if (logits_is_bf16)
    hipLaunchKernelGGL(argmax_kernel<true>, dim3(batch), dim3(kRowThreads), 0, as_stream(stream),
                       logits, ids, vocab, row_stride);
  else
    hipLaunchKernelGGL(argmax_kernel<false>, dim3(batch), dim3(kRowThreads), 0, as_stream(stream),
                       logits, ids, vocab, row_stride);
  SGL_CHECK_LAUNCH("argmax");
  return 0;
}

int sgl_amd_softmax_temperature(float* logits, const float* temperatures, int64_t batch,
                                int64_t vocab, int64_t row_stride, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(vocab > 0, "softmax_temperature: vocab must be positive");
  SGL_CHECK_ARG(batch <= 0x7fffffffLL, "softmax_temperature: batch too large");
  if (batch == 0) return 0;
  hipLaunchKernelGGL(softmax_temperature_kernel, dim3(batch), dim3(kRowThreads), 0,
                     as_stream(stream), logits, temperatures, vocab, row_stride);
  SGL_CHECK_LAUNCH("softmax_temperature");
  return 0;
}

}  // extern "C"

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
#include "softmax_ranges.hpp"

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

// The same arg-max with every row cut into `splits` column ranges, one workgroup each (a decode batch of 64 rows on
// one-workgroup-per-row keeps 64 of 256 CUs busy: 21 us for 16 MB).  A workgroup folds its range's winner into the
// row's 64-bit key  (order-preserving value bits << 32) | (2^32 - 1 - index)  with one atomicMax -- the larger key is
// the larger value, on ties the smaller index, NaN above everything, -0 == +0 -- and takes a ticket; the last arriver
// reads the key back, writes the id and re-arms key and ticket (the workspace is zero once, never again by the host).
__device__ __forceinline__ unsigned long long argmax_key(float v, uint32_t idx) {
  uint32_t u = __float_as_uint(v + 0.0f);                 // -0 -> +0
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  if (v != v) u = 0xffffffffu;
  return (static_cast<unsigned long long>(u) << 32) | (0xffffffffu - idx);
}

template <bool BF16>
__global__ __launch_bounds__(kRowThreads) void argmax_split_kernel(const void* __restrict__ logits, int64_t* __restrict__ ids,
                                                                    int64_t vocab, int64_t row_stride, int splits,
                                                                    unsigned long long* __restrict__ keys,
                                                                    unsigned int* __restrict__ tickets) {
  __shared__ unsigned long long s_key[16];
  const int64_t row = blockIdx.y;
  constexpr int EPV = BF16 ? 8 : 4;
  // ranges in whole 16-byte vectors (the host checked the row alignment); the last range takes the scalar tail
  const int64_t nvec = vocab / EPV;
  const int64_t per = (nvec + splits - 1) / splits;
  const int64_t v0 = blockIdx.x * per, v1 = (v0 + per < nvec) ? v0 + per : nvec;
  const char* base = static_cast<const char*>(logits) + row * row_stride * (BF16 ? 2 : 4);
  const U4* vbase = reinterpret_cast<const U4*>(base);
  unsigned long long best = 0ull;
  auto take = [&](float v, int64_t i) {
    const unsigned long long k = argmax_key(v, static_cast<uint32_t>(i));
    best = k > best ? k : best;
  };
  // Fast path per 16-byte vector: its maximum and its sum (a NaN -- or inf - inf -- poisons the sum); only a vector
  // that beats the thread's running maximum, or may hold a NaN, is looked at element by element with the full key
  // (~2 VALU ops per element instead of ~12: at 12 the kernel was VALU-bound, 18.7 us for 16 MB).
  float run_max = -INFINITY;
  bool any = false;
  auto take_vec = [&](const U4& v, int64_t i0) {
    float e[8];
    if (BF16) {
      e[0] = bf_lo(v.x); e[1] = bf_hi(v.x); e[2] = bf_lo(v.y); e[3] = bf_hi(v.y);
      e[4] = bf_lo(v.z); e[5] = bf_hi(v.z); e[6] = bf_lo(v.w); e[7] = bf_hi(v.w);
    } else {
      e[0] = __uint_as_float(v.x); e[1] = __uint_as_float(v.y); e[2] = __uint_as_float(v.z); e[3] = __uint_as_float(v.w);
    }
    float m = e[0], sum = e[0];
#pragma unroll
    for (int q = 1; q < EPV; ++q) { m = fmaxf(m, e[q]); sum += e[q]; }
    if (m > run_max || sum != sum || !any) {
#pragma unroll
      for (int q = 0; q < EPV; ++q) take(e[q], i0 + q);
      run_max = fmaxf(run_max, m);
      any = true;
    }
  };
  const int nthr = blockDim.x;                           // 1024 (one-launch form) or 256 (two-launch form: eight vectors per thread)
  int64_t j = v0 + threadIdx.x;
  for (; j + 3 * nthr < v1; j += 4 * nthr) {             // four loads in flight
    const U4 a = vbase[j], b2 = vbase[j + nthr], c2 = vbase[j + 2 * nthr], d2 = vbase[j + 3 * nthr];
    take_vec(a, j * EPV);
    take_vec(b2, (j + nthr) * EPV);
    take_vec(c2, (j + 2 * nthr) * EPV);
    take_vec(d2, (j + 3 * nthr) * EPV);
  }
  for (; j < v1; j += nthr) take_vec(vbase[j], j * EPV);
  if (blockIdx.x == splits - 1)
    for (int64_t i = nvec * EPV + threadIdx.x; i < vocab; i += nthr) take(load_logit<BF16>(base, i), i);
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(best, off, 64);
    best = o > best ? o : best;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) s_key[wid] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nthr / 64; ++w) best = s_key[w] > best ? s_key[w] : best;
    if (tickets == nullptr) {
      // two-launch form (round 6): the range's key goes to its own word, a second launch takes the row maxima.  Inside the decode
      // step the one-launch form below measured 15.9 us for 16 MB: its two fences (agent-scope: an L2 write-back each) and the
      // returning atomics sit behind a stream of 1 GB of lm_head weights; two plain launches are ~9 us
      keys[row * splits + blockIdx.x] = best;
      return;
    }
    atomicMax(&keys[row], best);
    __threadfence();
    if (atomicAdd(&tickets[row], 1u) == static_cast<unsigned>(splits - 1)) {
      __threadfence();
      const unsigned long long k = atomicMax(&keys[row], 0ull);
      ids[row] = k ? static_cast<int64_t>(0xffffffffu - static_cast<uint32_t>(k & 0xffffffffu)) : 0;
      atomicExch(&keys[row], 0ull);
      atomicExch(&tickets[row], 0u);
    }
  }
}

__global__ __launch_bounds__(256) void argmax_merge_kernel(const unsigned long long* __restrict__ keys, int64_t* __restrict__ ids,
                                                           int64_t batch, int splits) {
  const int64_t row = blockIdx.x * 256ll + threadIdx.x;
  if (row >= batch) return;
  unsigned long long k = 0ull;
  for (int s = 0; s < splits; ++s) {
    const unsigned long long o = keys[row * splits + s];
    k = o > k ? o : k;
  }
  ids[row] = k ? static_cast<int64_t>(0xffffffffu - static_cast<uint32_t>(k & 0xffffffffu)) : 0;
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

// The same softmax for decode-sized batches of wide rows: a row is cut into `splits` column ranges so that the whole chip
// works on 64 rows (one 1024-thread workgroup per row reads its 0.5 MB three times at one CU's bandwidth: 141 us for
// [64, 128256]).  Pass 1: every range's maximum and sum of exp(x / t - that maximum); pass 2: every workgroup merges the
// row's partials in range order (deterministic) and normalises its range.  16-byte loads; ranges are multiples of four.
// `IN` = float (in place: out == logits) or uint16_t (bf16 logits widened on the fly: what `logits.float()` + the fp32 kernel compute,
// without the 16 MB read + 33 MB write of the separate widening pass and with half the bytes in both passes here).
// The pieces live in softmax_ranges.hpp (shared with the sampler that evaluates the probabilities of its candidates only).
template <typename IN>
__global__ __launch_bounds__(kSplitThreads) void softmax_partials_kernel(const IN* __restrict__ logits, const float* __restrict__ temperatures,
                                                                         int64_t vocab, int64_t row_stride, int splits,
                                                                         float* __restrict__ partials) {
  __shared__ float scratch[16];
  const int64_t row = blockIdx.y;
  const IN* x = logits + row * row_stride;
  const float t = temperatures[row];
  int64_t b, e;
  split_range_of(vocab, splits, blockIdx.x, &b, &e);
  float mx, sum;
  range_partial<IN>(x, t, b, e, scratch, &mx, &sum);
  if (threadIdx.x == 0) {
    partials[(row * splits + blockIdx.x) * 2 + 0] = mx;
    partials[(row * splits + blockIdx.x) * 2 + 1] = sum;
  }
}

template <typename IN>
__global__ __launch_bounds__(kSplitThreads) void softmax_normalize_kernel(const IN* __restrict__ logits, float* __restrict__ out,
                                                                          const float* __restrict__ temperatures, int64_t vocab,
                                                                          int64_t row_stride, int64_t out_stride, int splits,
                                                                          const float* __restrict__ partials) {
  const int64_t row = blockIdx.y;
  const IN* x = logits + row * row_stride;
  float* y = out + row * out_stride;
  const float t = temperatures[row];
  float mx, sum;
  merge_partials(partials + row * splits * 2, splits, &mx, &sum);
  int64_t b, e;
  split_range_of(vocab, splits, blockIdx.x, &b, &e);
  const int64_t e4 = b + (e - b) / 4 * 4;
  for (int64_t i = b + 4 * threadIdx.x; i < e4; i += 4 * kSplitThreads) {
    float v[4];
    ld4<IN>(x + i, v);
    float4 o;
    o.x = softmax_prob(v[0], t, mx, sum); o.y = softmax_prob(v[1], t, mx, sum); o.z = softmax_prob(v[2], t, mx, sum); o.w = softmax_prob(v[3], t, mx, sum);
    *reinterpret_cast<float4*>(y + i) = o;
  }
  for (int64_t i = e4 + threadIdx.x; i < e; i += kSplitThreads) y[i] = softmax_prob(ld1(x + i), t, mx, sum);
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

int64_t sgl_amd_argmax_split_workspace_bytes(int64_t batch) { return batch * 64 * 8; }   /* a key per (row, range <= 64) */

int sgl_amd_argmax_split(const void* logits, int logits_is_bf16, int64_t* ids, int64_t batch, int64_t vocab,
                         int64_t row_stride, int num_splits, void* workspace, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(vocab > 0 && vocab < (int64_t{1} << 32), "argmax_split: vocab must be in [1, 2^32)");
  SGL_CHECK_ARG(batch <= 65535 && num_splits >= 1 && num_splits <= 64, "argmax_split: batch <= 65535, 1..64 splits");
  SGL_CHECK_ARG(workspace && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && (row_stride * (logits_is_bf16 ? 2 : 4)) % 16 == 0,
                "argmax_split: needs the workspace and 16-byte aligned rows");
  if (batch == 0) return 0;
  unsigned long long* keys = static_cast<unsigned long long*>(workspace);
  const dim3 grid(num_splits, static_cast<unsigned>(batch));
  if (logits_is_bf16)
    hipLaunchKernelGGL(argmax_split_kernel<true>, grid, dim3(256), 0, as_stream(stream), logits, ids, vocab, row_stride,
                       num_splits, keys, static_cast<unsigned int*>(nullptr));
  else
    hipLaunchKernelGGL(argmax_split_kernel<false>, grid, dim3(256), 0, as_stream(stream), logits, ids, vocab, row_stride,
                       num_splits, keys, static_cast<unsigned int*>(nullptr));
  hipLaunchKernelGGL(argmax_merge_kernel, dim3(static_cast<unsigned>((batch + 255) / 256)), dim3(256), 0, as_stream(stream), keys, ids,
                     batch, num_splits);
  SGL_CHECK_LAUNCH("argmax_split");
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

int64_t sgl_amd_softmax_temperature_split_workspace_bytes(int64_t batch, int num_splits) { return batch * num_splits * 8; }

int sgl_amd_softmax_temperature_split(float* logits, const float* temperatures, int64_t batch, int64_t vocab, int64_t row_stride,
                                      int num_splits, void* workspace, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(vocab > 0 && batch <= 65535 && num_splits >= 1 && num_splits <= 64, "softmax_temperature_split: batch <= 65535, 1..64 splits");
  SGL_CHECK_ARG(workspace && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && row_stride % 4 == 0,
                "softmax_temperature_split: needs the workspace and 16-byte aligned rows");
  if (batch == 0) return 0;
  const dim3 grid(num_splits, static_cast<unsigned>(batch));
  float* partials = static_cast<float*>(workspace);
  hipLaunchKernelGGL(softmax_partials_kernel<float>, grid, dim3(kSplitThreads), 0, as_stream(stream), static_cast<const float*>(logits), temperatures,
                     vocab, row_stride, num_splits, partials);
  hipLaunchKernelGGL(softmax_normalize_kernel<float>, grid, dim3(kSplitThreads), 0, as_stream(stream), static_cast<const float*>(logits), logits,
                     temperatures, vocab, row_stride, row_stride, num_splits, partials);
  SGL_CHECK_LAUNCH("softmax_temperature_split");
  return 0;
}

int sgl_amd_softmax_temperature_split_bf16(const void* logits_bf16, float* probs, const float* temperatures, int64_t batch, int64_t vocab,
                                           int64_t logits_row_stride, int64_t probs_row_stride, int num_splits, void* workspace, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(vocab > 0 && batch <= 65535 && num_splits >= 1 && num_splits <= 64, "softmax_temperature_split_bf16: batch <= 65535, 1..64 splits");
  SGL_CHECK_ARG(workspace && logits_bf16 && probs && (reinterpret_cast<uintptr_t>(logits_bf16) & 7) == 0 && logits_row_stride % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(probs) & 15) == 0 && probs_row_stride % 4 == 0,
                "softmax_temperature_split_bf16: needs the workspace, 8-byte aligned bf16 rows and 16-byte aligned fp32 rows");
  if (batch == 0) return 0;
  const dim3 grid(num_splits, static_cast<unsigned>(batch));
  float* partials = static_cast<float*>(workspace);
  const uint16_t* x = static_cast<const uint16_t*>(logits_bf16);
  hipLaunchKernelGGL(softmax_partials_kernel<uint16_t>, grid, dim3(kSplitThreads), 0, as_stream(stream), x, temperatures, vocab, logits_row_stride,
                     num_splits, partials);
  hipLaunchKernelGGL(softmax_normalize_kernel<uint16_t>, grid, dim3(kSplitThreads), 0, as_stream(stream), x, probs, temperatures, vocab,
                     logits_row_stride, probs_row_stride, num_splits, partials);
  SGL_CHECK_LAUNCH("softmax_temperature_split_bf16");
  return 0;
}

}  // extern "C"

// Column-range softmax pieces shared by sampling.hip (the two softmax launches) and sampling_topk.hip (the sampler that never
// writes the probabilities): ONE definition of the range geometry, of a range's (maximum, sum) partial, of the merge of a row's
// partials and of a probability, so that every launch that evaluates p(x) = expf(x / t - max) / sum produces the same bits.
// (sampler.py:211-216: logits.div_(temperatures); softmax(logits, dim=-1).)
#pragma once
#include "common.hpp"

namespace sgl_amd {

constexpr int kSplitThreads = 256;

// range `r` of `splits` over a row of `vocab` columns: multiples of four columns
__device__ __forceinline__ void split_range_of(int64_t vocab, int splits, int r, int64_t* begin, int64_t* end) {
  const int64_t per = ((vocab + splits - 1) / splits + 3) / 4 * 4;
  int64_t b = per * r, e = b + per;
  if (b > vocab) b = vocab;
  if (e > vocab) e = vocab;
  *begin = b; *end = e;
}

// `IN` = float or uint16_t (bf16 logits widened on the fly: what `logits.float()` computes, exactly)
template <typename IN>
__device__ __forceinline__ void ld4(const IN* p, float (&v)[4]);
template <>
__device__ __forceinline__ void ld4<float>(const float* p, float (&v)[4]) {
  const float4 q = *reinterpret_cast<const float4*>(p);
  v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
template <>
__device__ __forceinline__ void ld4<uint16_t>(const uint16_t* p, float (&v)[4]) {
  const uint2 q = *reinterpret_cast<const uint2*>(p);
  v[0] = bf_lo(q.x); v[1] = bf_hi(q.x); v[2] = bf_lo(q.y); v[3] = bf_hi(q.y);
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const uint16_t* p) { return bf2f(*p); }

// (maximum of x / t, sum of expf(x / t - that maximum)) over columns [b, e) of one row, by a kSplitThreads workgroup;
// every thread returns the pair.  `scratch` = 16 floats of LDS.
template <typename IN>
__device__ __forceinline__ void range_partial(const IN* x, float t, int64_t b, int64_t e, float* scratch, float* out_max, float* out_sum) {
  const int64_t e4 = b + (e - b) / 4 * 4;
  float mx = -INFINITY;
  for (int64_t i = b + 4 * threadIdx.x; i < e4; i += 4 * kSplitThreads) {
    float v[4];
    ld4<IN>(x + i, v);
    mx = fmaxf(fmaxf(mx, v[0] / t), fmaxf(v[1] / t, fmaxf(v[2] / t, v[3] / t)));
  }
  for (int64_t i = e4 + threadIdx.x; i < e; i += kSplitThreads) mx = fmaxf(mx, ld1(x + i) / t);
  mx = block_max(mx, scratch);
  float sum = 0.f;
  if (mx > -INFINITY) {
    for (int64_t i = b + 4 * threadIdx.x; i < e4; i += 4 * kSplitThreads) {
      float v[4];
      ld4<IN>(x + i, v);
      sum += expf(v[0] / t - mx) + expf(v[1] / t - mx) + expf(v[2] / t - mx) + expf(v[3] / t - mx);
    }
    for (int64_t i = e4 + threadIdx.x; i < e; i += kSplitThreads) sum += expf(ld1(x + i) / t - mx);
  }
  sum = block_sum(sum, scratch);
  *out_max = mx;
  *out_sum = sum;
}

// the row's maximum and sum from its `splits` partials, in range order: the same bits in every workgroup
__device__ __forceinline__ void merge_partials(const float* pr, int splits, float* out_max, float* out_sum) {
  float mx = -INFINITY;
  for (int s = 0; s < splits; ++s) mx = fmaxf(mx, pr[2 * s]);
  float sum = 0.f;
  for (int s = 0; s < splits; ++s) {
    const float ms = pr[2 * s];
    if (ms > -INFINITY) sum = __fmaf_rn(pr[2 * s + 1], expf(ms - mx), sum);   // (an explicit fma: the same bits wherever this chain is evaluated)
  }
  *out_max = mx;
  *out_sum = sum;
}

__device__ __forceinline__ float softmax_prob(float x, float t, float mx, float sum) { return expf(x / t - mx) / sum; }

}  // namespace sgl_amd

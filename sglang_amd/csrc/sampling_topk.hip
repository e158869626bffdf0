// Top-k / top-p / min-p sampling for gfx950 with the reference's deterministic
// (murmur-hash gumbel) mode, without sorting the vocabulary.
//
// Replaces (reference, /root/reference/python/sglang):
//   srt/layers/sampler.py:567-612 top_k_top_p_min_p_sampling_from_probs_torch
//     (sort desc -> cumsum -> zero rank >= top_k -> zero (cumsum - p) > top_p ->
//      [min_p] -> gumbel argmax over the SORTED ranks),
//   :688-729 multinomial_with_seed, :732-750 sampling_from_probs_torch,
//   :753-762 top_p_normalize_probs_torch,
//   kernels/ops/sampling/murmur_hash.py:51-121 murmur_hash32,
//   sgl_kernel.top_k_renorm_prob / top_p_renorm_prob (kernels/aot/python/sgl_kernel/sampling.py).
//
// One 1024-thread workgroup per row.  The torch path sorts all V probabilities;
// only the kept prefix of that order matters, so the kernel
//   1. finds the kept prefix with a 4-level MSB radix select over the fp32 bit
//      patterns (256-bin count + fixed-point sum histograms in LDS): the result is a
//      threshold value, the number of strictly larger elements and how many of the
//      elements equal to the threshold are kept.  Sums are 2^-40 fixed point, so the
//      result is independent of the order of the atomics (deterministic);
//   2. compacts the kept (value, token) pairs in token order (two ballot scans per tile);
//   3. ranks them: descending value, ties by token id.  Up to kLdsKeep pairs are ranked
//      in LDS by counting; larger nuclei take a stable 4 x 8-bit LSD radix sort in a
//      caller-owned global workspace;
//   4. scores rank j with  log(p_j) + gumbel(murmur(seed, position, j))  in fp64 and
//      takes the arg max (first maximum), exactly the reference's arithmetic.
// The row (V x 4 B, ~0.5 MB) is read 5-6 times but stays in L2.
#include "common.hpp"
#include "sglang_amd.h"
#include "softmax_ranges.hpp"

using namespace sgl_amd;

namespace {

constexpr int kT = 1024;          // threads per row
constexpr int kNW = kT / 64;      // waves per row
constexpr int kLdsKeep = 2048;    // nucleus size ranked in LDS
constexpr double kFix = 1099511627776.0;  // 2^40

// ---- murmur3 (murmur_hash.py:17-99) -------------------------------------------
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t mm_mix(uint32_t h, uint32_t k) {
  k *= 0xCC9E2D51u;
  k = rotl32(k, 15);
  k *= 0x1B873593u;
  h ^= k;
  h = rotl32(h, 13);
  return h * 5u + 0xE6546B64u;
}
__device__ __forceinline__ uint32_t mm_fmix(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t murmur_hash32(uint32_t h_seed_pos, uint32_t col) {
  uint32_t h = mm_mix(h_seed_pos, col);
  h ^= 16u;
  return mm_fmix(h);
}
__device__ __forceinline__ uint32_t murmur_prefix(uint64_t seed, uint32_t pos) {
  uint32_t h = 0;
  h = mm_mix(h, static_cast<uint32_t>(seed & 0xffffffffull));
  h = mm_mix(h, static_cast<uint32_t>(seed >> 32));
  return mm_mix(h, pos);
}
// sampler.py:716-723: x = hash / (2^32-1); g = -log(clamp(-log(x), 2^-32, DBL_MAX)).
__device__ __forceinline__ double gumbel_from_hash(uint32_t h) {
  const double x = static_cast<double>(h) / 4294967295.0;
  double lx = log(x);                       // h == 0 -> -inf
  if (lx < -1.7976931348623157e308) lx = -1.7976931348623157e308;
  if (lx > -2.3283064365386963e-10) lx = -2.3283064365386963e-10;
  return -log(-lx);
}

struct Best {
  double score;
  int rank;    // sorted rank (tie-break: first)
  int token;
};
__device__ __forceinline__ bool best_better(const Best& a, const Best& b) {
  // true if b should replace a.  torch.argmax: NaN is maximal, first index wins ties.
  if (b.rank < 0) return false;
  if (a.rank < 0) return true;
  const bool an = a.score != a.score, bn = b.score != b.score;
  if (an != bn) return bn;
  if (!an && b.score != a.score) return b.score > a.score;
  return b.rank < a.rank;
}
__device__ __forceinline__ Best block_best(Best v, double* s_score, int* s_rank, int* s_tok) {
  for (int off = 32; off > 0; off >>= 1) {
    Best o;
    o.score = __shfl_xor(v.score, off, 64);
    o.rank = __shfl_xor(v.rank, off, 64);
    o.token = __shfl_xor(v.token, off, 64);
    if (best_better(v, o)) v = o;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { s_score[wid] = v.score; s_rank[wid] = v.rank; s_tok[wid] = v.token; }
  __syncthreads();
  // every wave merges the kNW partials redundantly with the same butterfly (a serial scalar
  // loop over LDS here was mis-compiled by hipcc 7.2: the token of partial 1 got dropped)
  Best r{0.0, -1, 0};
  if (lane < kNW) { r.score = s_score[lane]; r.rank = s_rank[lane]; r.token = s_tok[lane]; }
  for (int off = kNW / 2; off > 0; off >>= 1) {
    Best o;
    o.score = __shfl_xor(r.score, off, 64);
    o.rank = __shfl_xor(r.rank, off, 64);
    o.token = __shfl_xor(r.token, off, 64);
    if (best_better(r, o)) r = o;
  }
  r.score = __shfl(r.score, 0, 64);
  r.rank = __shfl(r.rank, 0, 64);
  r.token = __shfl(r.token, 0, 64);
  return r;
}

__device__ __forceinline__ uint64_t to_fix(float p) {
  const double d = static_cast<double>(p) * kFix;
  return d > 0.0 ? static_cast<uint64_t>(d + 0.5) : 0ull;
}

struct Select {
  uint32_t thr_key;   // bit pattern of the smallest kept value
  int n_eq_keep;      // how many elements == thr are kept (lowest token ids first)
};

constexpr int kCand = 4096;       // candidate list capacity (4 per thread: one tile)
constexpr int kDirect = 2048;     // preloaded candidate lists up to this length (= kLdsKeep) are ranked whole (sample_row)
constexpr int kHistCopies = 16;   // lane-group private copies of the select histograms
constexpr int kHistStride = 257;  // ... one bank apart

struct RowSmem {
  uint32_t hist_cnt[256];
  unsigned long long hist_sum[256];
  // A softmax row lands in a handful of top-byte bins: 64 lanes of a wave adding to ONE LDS word serialise (two atomics
  // per element: ~0.1 ms of the 0.18 ms kernel at V = 128 K).  Lane l adds to copy l & 15, the copies are summed into
  // hist_cnt / hist_sum before the scan: integer and fixed-point sums, so the result does not depend on the split.
  uint32_t part_cnt[kHistCopies * kHistStride];
  unsigned long long part_sum[kHistCopies * kHistStride];
  // candidates of the cut (sample_kernel): every element of the first-level bin the cut falls into and of the bins above
  // it, in token order -- the later levels and the compaction run on this list instead of the row
  alignas(16) uint32_t cand_val[kCand];
  int cand_tok[kCand];
  int wave_a[kNW];
  int wave_b[kNW];
  double s_score[kNW];
  int s_rank[kNW];
  int s_tok[kNW];
  // select state (written by wave 0, read by all)
  uint32_t prefix;
  int c_above;
  unsigned long long s_above;
  int found_bin;
  int n_eq_keep;
  int done_all;
  int n_keep;
  float red[16];
  uint32_t keys[kLdsKeep];
  int toks[kLdsKeep];
  // radix-sort scratch (aliases nothing: only used on the large path)
  uint32_t digit_cnt[kNW][256];
  uint32_t digit_base[256];
};

// rows whose start is 16-byte aligned are read four values at a time up to this index (the tail, and unaligned rows
// altogether, one by one)
__device__ __forceinline__ int vec4_len(const float* x, int V) {
  return (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? (V & ~3) : 0;
}

// probabilities are >= 0: the fp32 bit pattern orders like the value.  -0.0 and NaN
// are mapped to 0 / +inf bits so the order stays total.
__device__ __forceinline__ uint32_t key_of(float p) {
  if (!(p > 0.f)) return (p != p) ? 0x7f800000u : 0u;
  return __float_as_uint(p);
}

// ---- 1. radix select ------------------------------------------------------------
// top_k < 0 or >= V behaves as "all"; top_p >= 1 keeps everything the cumsum allows.
// top_p >= 1 disables the filter: the exclusive cumsum of a softmax row never exceeds 1 in the
// reference's fp32 arithmetic, while the exact fixed-point sum can be a few 2^-40 above it.
__device__ __forceinline__ unsigned long long top_p_fix(float top_p) {
  return top_p >= 1.0f ? ~0ull >> 1
                       : static_cast<unsigned long long>(top_p > 0.f ? static_cast<double>(top_p) * kFix : 0.0);
}

// The sampler's cut is decided EXACTLY afterwards (exact_top_p_keep): the select only has to return a prefix of the sorted
// order that contains every element the reference keeps.  The reference's test value fl32(fl32(S_i) - p_i) is the
// exclusive prefix sum within 2^-24, the fixed-point sums are it within V * 2^-41 <= 2^-24: a margin of 2^-22 covers both.
__device__ __forceinline__ unsigned long long top_p_fix_superset(float top_p) {
  if (top_p > 1.001f) return ~0ull >> 1;             // a softmax row's prefix sums never get there
  const double t = (top_p > 0.f ? static_cast<double>(top_p) : 0.0) + 2.384185791015625e-07;   // + 2^-22
  return static_cast<unsigned long long>(t * kFix);
}

// sampler.py:577-580 on the torch CPU path, for sorted rank i with inclusive prefix sum S_i (torch.cumsum accumulates a
// float row in double and rounds every prefix to float): the element is zeroed when  fl32(fl32(S_i) - p_i) > top_p.
__device__ __forceinline__ bool exact_top_p_keep(double s_incl, float p, float top_p) {
  const float cs = static_cast<float>(s_incl);
  return !(__fsub_rn(cs, p) > top_p);
}

// inclusive scan of one double per thread over the workgroup, in thread order (fixed association: deterministic)
__device__ __forceinline__ double block_scan_incl(double v, double* s_part /* kNW */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double o = __shfl_up(v, off, 64);
    if (lane >= off) v += o;
  }
  __syncthreads();
  if (lane == 63) s_part[wid] = v;
  __syncthreads();
  double base = 0.0;
  for (int w = 0; w < wid; ++w) base += s_part[w];
  __syncthreads();
  return base + v;
}

template <class SM>
__device__ __forceinline__ void select_begin(SM& sm) {
  if (threadIdx.x == 0) { sm.prefix = 0; sm.c_above = 0; sm.s_above = 0; sm.done_all = 0; sm.n_eq_keep = 0; }
}

// The decision of one level, from the level's histogram in sm.hist_cnt / hist_sum (bins of the values that match `prefix`): walks
// the bins from the top, finds the bin the cut falls into and updates the select state.  Wave 0 only; the caller's barrier follows.
template <class SM>
__device__ __forceinline__ void select_decide(int level, int shift, uint32_t prefix, int64_t top_k, unsigned long long p_fix, SM& sm) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    // lane l owns bins 255-4l .. 252-4l (descending); inclusive scan over lanes.
    uint32_t c4[4];
    unsigned long long s4[4];
    uint32_t lc = 0;
    unsigned long long ls = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c4[j] = sm.hist_cnt[255 - 4 * tid - j];
      s4[j] = sm.hist_sum[255 - 4 * tid - j];
      lc += c4[j];
      ls += s4[j];
    }
    uint32_t ic = lc;
    unsigned long long is = ls;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t oc = __shfl_up(ic, off, 64);
      const unsigned long long os = __shfl_up(is, off, 64);
      if (tid >= off) { ic += oc; is += os; }
    }
    long long c = static_cast<long long>(sm.c_above) + (ic - lc);
    unsigned long long s = sm.s_above + (is - ls);
    int fail = -1;
    long long c_b = 0;
    unsigned long long s_b = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (fail < 0) {
        const bool whole = (c + c4[j] <= top_k) && (s + s4[j] <= p_fix);
        if (whole || c4[j] == 0) { c += c4[j]; s += s4[j]; }
        else { fail = 255 - 4 * tid - j; c_b = c; s_b = s; }
      }
    }
    const unsigned long long bal = __ballot(fail >= 0);
    if (bal == 0ull) {
      if (tid == 0) sm.done_all = 1;   // everything at this prefix is kept (only possible at level 0)
    } else {
      const int first = __ffsll(static_cast<long long>(bal)) - 1;
      if (tid == first) {
        sm.found_bin = fail;
        sm.c_above = static_cast<int>(c_b);
        sm.s_above = s_b;
        sm.prefix = prefix | (static_cast<uint32_t>(fail) << shift);
        if (level == 3) {
          // single value v, m copies: keep j = 0.. while c_b + j < top_k and s_b + j*v <= p_fix
          const uint32_t m = sm.hist_cnt[fail];
          const unsigned long long v = sm.hist_sum[fail] / (m ? m : 1);
          long long nk = top_k - c_b;
          if (nk < 0) nk = 0;
          long long np;
          if (s_b > p_fix) np = 0;
          else if (v == 0) np = m;
          else np = static_cast<long long>((p_fix - s_b) / v) + 1;
          long long n = m;
          if (nk < n) n = nk;
          if (np < n) n = np;
          sm.n_eq_keep = static_cast<int>(n);
        }
      }
    }
  }
}

// Levels [level_begin, level_end) of the select over the values x[0 .. V) (a row, or the candidate list in LDS); the
// state (prefix, counts above, done_all, n_eq_keep) lives in `sm`.  Ends on a barrier.
__device__ void select_levels(const float* __restrict__ x, int V, int level_begin, int level_end, int64_t top_k,
                              unsigned long long p_fix, RowSmem& sm) {
  const int tid = threadIdx.x;
  uint32_t mask = level_begin > 0 ? 0xffffffffu << (32 - 8 * level_begin) : 0u;
  for (int level = level_begin; level < level_end; ++level) {
    const int shift = 24 - 8 * level;
    for (int z = tid; z < kHistCopies * kHistStride; z += kT) { sm.part_cnt[z] = 0; sm.part_sum[z] = 0; }
    __syncthreads();
    const uint32_t prefix = sm.prefix;
    const int copy = (tid & (kHistCopies - 1)) * kHistStride;
    auto tally = [&](float p) {
      const uint32_t key = key_of(p);
      if ((key & mask) == prefix) {
        const int b = copy + ((key >> shift) & 255);
        atomicAdd(&sm.part_cnt[b], 1u);
        atomicAdd(&sm.part_sum[b], static_cast<unsigned long long>(to_fix(p)));
      }
    };
    // four values per load: a pass of one value per thread and iteration is 125 dependent L2 round trips at V = 128 K
    const int V4 = vec4_len(x, V);
    for (int i = 4 * tid; i < V4; i += 4 * kT) {
      const float4 v = *reinterpret_cast<const float4*>(x + i);
      tally(v.x); tally(v.y); tally(v.z); tally(v.w);
    }
    for (int i = V4 + tid; i < V; i += kT) tally(x[i]);
    __syncthreads();
    if (tid < 256) {
      uint32_t c = 0;
      unsigned long long sfix = 0;
#pragma unroll
      for (int k = 0; k < kHistCopies; ++k) { c += sm.part_cnt[k * kHistStride + tid]; sfix += sm.part_sum[k * kHistStride + tid]; }
      sm.hist_cnt[tid] = c;
      sm.hist_sum[tid] = sfix;
    }
    __syncthreads();
    select_decide(level, shift, prefix, top_k, p_fix, sm);
    __syncthreads();
    if (sm.done_all) break;
    mask |= 0xffu << shift;
  }
}

__device__ __forceinline__ Select select_result(RowSmem& sm) {
  Select r;
  if (sm.done_all) { r.thr_key = 0; r.n_eq_keep = 0x7fffffff; }
  else { r.thr_key = sm.prefix; r.n_eq_keep = sm.n_eq_keep; }
  __syncthreads();
  return r;
}

__device__ Select radix_select(const float* __restrict__ x, int V, int64_t top_k, float top_p, RowSmem& sm) {
  select_begin(sm);
  select_levels(x, V, 0, 4, top_k, top_p_fix(top_p), sm);
  return select_result(sm);
}

// The candidates of the cut, in token order: every element whose first-level digit is >= `bin` (the bin the cut falls
// into) -> (value bits, token) in sm.cand_*.  The caller has checked that they fit.
__device__ void collect_candidates(const float* __restrict__ x, int V, int bin, RowSmem& sm) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t lo = static_cast<uint32_t>(bin) << 24;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int V4 = vec4_len(x, V);
  int base = 0;
  for (int i0 = 0; i0 < V4; i0 += 4 * kT) {
    const int i = i0 + 4 * tid;
    float val[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < V4) {
      const float4 v = *reinterpret_cast<const float4*>(x + i);
      val[0] = v.x; val[1] = v.y; val[2] = v.z; val[3] = v.w;
    }
    bool in[4];
    int before = 0, wave = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      in[j] = i < V4 && key_of(val[j]) >= lo;
      const unsigned long long b = __ballot(in[j]);
      before += __popcll(b & lt_mask);
      wave += __popcll(b);
    }
    __syncthreads();
    if (lane == 0) sm.wave_a[wid] = wave;
    __syncthreads();
    int o = 0, t = 0;
    for (int w = 0; w < kNW; ++w) {
      const int a = sm.wave_a[w];
      if (w < wid) o += a;
      t += a;
    }
    int pos = base + o + before;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (in[j]) { sm.cand_val[pos] = __float_as_uint(val[j]); sm.cand_tok[pos] = i + j; ++pos; }
    base += t;
  }
  for (int i0 = V4; i0 < V; i0 += kT) {
    const int i = i0 + tid;
    const float v = i < V ? x[i] : 0.f;
    const bool in = i < V && key_of(v) >= lo;
    const unsigned long long b = __ballot(in);
    __syncthreads();
    if (lane == 0) sm.wave_a[wid] = __popcll(b);
    __syncthreads();
    int o = 0, t = 0;
    for (int w = 0; w < kNW; ++w) {
      const int a = sm.wave_a[w];
      if (w < wid) o += a;
      t += a;
    }
    if (in) { const int pos = base + o + __popcll(b & lt_mask); sm.cand_val[pos] = __float_as_uint(v); sm.cand_tok[pos] = i; }
    base += t;
  }
  __syncthreads();
}

// ---- 2. compaction ---------------------------------------------------------------
// Kept = key > thr (and key >= min_key)  ||  key == thr with equal-rank < n_eq_keep.
// Emits pairs through `emit(pos, key, token)`; greater elements first (token order), then
// the kept ties.  Returns n_keep.  n_gt is computed in a first counting sweep.
template <typename Emit>
__device__ int compact_kept(const float* __restrict__ x, int V, Select sel, uint32_t min_key,
                            RowSmem& sm, Emit emit) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  uint32_t thr = sel.thr_key;
  int n_eq_keep = sel.n_eq_keep;
  if (min_key > thr) { thr = min_key; n_eq_keep = 0x7fffffff; }   // min_p cut is above the top-k/p cut: keep >= min_key
  // count strictly-greater elements (the ties go after them)
  const int V4 = vec4_len(x, V);
  int cnt = 0;
  for (int i = 4 * tid; i < V4; i += 4 * kT) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    cnt += (key_of(v.x) > thr) + (key_of(v.y) > thr) + (key_of(v.z) > thr) + (key_of(v.w) > thr);
  }
  for (int i = V4 + tid; i < V; i += kT) cnt += key_of(x[i]) > thr;
  cnt = static_cast<int>(wave_sum(static_cast<float>(cnt)) + 0.5f);
  __syncthreads();
  if (lane == 0) sm.wave_a[wid] = cnt;
  __syncthreads();
  int n_gt = 0;
  for (int w = 0; w < kNW; ++w) n_gt += sm.wave_a[w];
  int base_gt = 0, base_eq = 0;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // token order inside a tile of 4 kT values: thread, then the four values of its load (index i0 + 4 tid + j)
  for (int i0 = 0; i0 < V4; i0 += 4 * kT) {
    const int i = i0 + 4 * tid;
    uint32_t key[4] = {0u, 0u, 0u, 0u};
    if (i < V4) {
      const float4 v = *reinterpret_cast<const float4*>(x + i);
      key[0] = key_of(v.x); key[1] = key_of(v.y); key[2] = key_of(v.z); key[3] = key_of(v.w);
    }
    int before_g = 0, before_e = 0, wave_g = 0, wave_e = 0;
    bool gt[4], eq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gt[j] = i < V4 && key[j] > thr;
      eq[j] = i < V4 && key[j] == thr;
      const unsigned long long bg = __ballot(gt[j]), be = __ballot(eq[j]);
      before_g += __popcll(bg & lt_mask); before_e += __popcll(be & lt_mask);
      wave_g += __popcll(bg); wave_e += __popcll(be);
    }
    __syncthreads();
    if (lane == 0) { sm.wave_a[wid] = wave_g; sm.wave_b[wid] = wave_e; }
    __syncthreads();
    int og = 0, oe = 0, tg = 0, te = 0;
    for (int w = 0; w < kNW; ++w) {
      const int a = sm.wave_a[w], b = sm.wave_b[w];
      if (w < wid) { og += a; oe += b; }
      tg += a; te += b;
    }
    int pg = base_gt + og + before_g, pe = base_eq + oe + before_e;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (gt[j]) emit(pg++, key[j], i + j);
      if (eq[j]) {
        if (pe < n_eq_keep) emit(n_gt + pe, key[j], i + j);
        ++pe;
      }
    }
    base_gt += tg;
    base_eq += te;
  }
  for (int i0 = V4; i0 < V; i0 += kT) {
    const int i = i0 + tid;
    uint32_t key = 0;
    bool gt = false, eq = false;
    if (i < V) {
      key = key_of(x[i]);
      gt = key > thr;
      eq = key == thr;
    }
    const unsigned long long bg = __ballot(gt), be = __ballot(eq);
    __syncthreads();
    if (lane == 0) { sm.wave_a[wid] = __popcll(bg); sm.wave_b[wid] = __popcll(be); }
    __syncthreads();
    int og = 0, oe = 0, tg = 0, te = 0;
    for (int w = 0; w < kNW; ++w) {
      const int a = sm.wave_a[w], b = sm.wave_b[w];
      if (w < wid) { og += a; oe += b; }
      tg += a; te += b;
    }
    if (gt) emit(base_gt + og + __popcll(bg & lt_mask), key, i);
    if (eq) {
      const int r = base_eq + oe + __popcll(be & lt_mask);
      if (r < n_eq_keep) emit(n_gt + r, key, i);
    }
    base_gt += tg;
    base_eq += te;
  }
  const int kept_eq = base_eq < n_eq_keep ? base_eq : n_eq_keep;
  __syncthreads();
  return n_gt + kept_eq;
}

// ---- 3b. stable LSD radix sort (descending) of n (key, token) pairs in global memory --
// ping-pongs between (k0,t0) and (k1,t1); after 4 passes the result is back in (k0,t0).
__device__ void radix_sort_desc(uint32_t* k0, int* t0, uint32_t* k1, int* t1, int n, RowSmem& sm) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t* ks = k0; int* ts = t0; uint32_t* kd = k1; int* td = t1;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 8 * pass;
    if (tid < 256) sm.hist_cnt[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kT) atomicAdd(&sm.hist_cnt[(ks[i] >> shift) & 255], 1u);
    __syncthreads();
    if (tid == 0) {   // descending: digit 255 first
      uint32_t run = 0;
      for (int d = 255; d >= 0; --d) { sm.digit_base[d] = run; run += sm.hist_cnt[d]; }
    }
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += kT) {
      const int i = i0 + tid;
      const bool act = i < n;
      uint32_t key = 0; int tok = 0; int d = 0;
      if (act) { key = ks[i]; tok = ts[i]; d = (key >> shift) & 255; }
      for (int z = tid; z < kNW * 256; z += kT) (&sm.digit_cnt[0][0])[z] = 0;
      // lanes of this wave with the same digit
      unsigned long long m = __ballot(act);
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
        const unsigned long long bb = __ballot((d >> bit) & 1);
        m &= ((d >> bit) & 1) ? bb : ~bb;
      }
      __syncthreads();
      if (act && (m & lt_mask) == 0ull) sm.digit_cnt[wid][d] = __popcll(m);   // group leader
      __syncthreads();
      uint32_t off = 0;
      if (act) {
        off = sm.digit_base[d];
        for (int w = 0; w < wid; ++w) off += sm.digit_cnt[w][d];
        off += __popcll(m & lt_mask);
        kd[off] = key;
        td[off] = tok;
      }
      __syncthreads();
      if (tid < 256) {
        uint32_t tot = 0;
        for (int w = 0; w < kNW; ++w) tot += sm.digit_cnt[w][tid];
        sm.digit_base[tid] += tot;
      }
      __syncthreads();
    }
    // make this pass's global writes visible to the whole workgroup before the next pass reads them
    __threadfence_block();
    __syncthreads();
    uint32_t* tk = ks; ks = kd; kd = tk;
    int* tt = ts; ts = td; td = tt;
  }
}

struct SampleParams {
  const float* probs;          // [B, V]
  int64_t row_stride;
  int V;
  const int32_t* top_ks;       // [B] or null (all)
  const float* top_ps;         // [B] or null (1.0)
  const float* min_ps;         // [B] or null
  const int64_t* seeds;        // [B]
  const int64_t* positions;    // [B] or null (0)
  int32_t* out_ids;            // [B]
  uint32_t* ws_keys;           // [B, 2, V] (large nuclei only)
  int32_t* ws_toks;            // [B, 2, V]
  int32_t* out_n_keep;         // optional [B] (tests)
  int filtered;                // 0: sampling_from_probs (no filter, col = token id)
};

// One row of the sampler, by one 1024-thread workgroup.  n_preloaded < 0: from the row itself (the first radix level and the
// candidate collection are two full-row passes).  n_preloaded >= 0 (the column-range launches below did those passes over the
// whole chip): the select state behind level 0 is in `sm` and sm.cand_* hold the n_preloaded candidates in token order.
// The candidate list of sample_from_logits (below) is exact only above a FRONTIER: the largest probability among the elements
// its column ranges did not emit.  Ranks and prefix sums of list elements above it are the reference's; the element the reference
// has at the first rank at or below it has exactly that probability.  If the three rules would keep THAT element the list is
// not enough (`*unusable` is set: the row is redone from the full row); if they do not, nothing at or behind it is kept
// (rank >= top_k, value < the min-p threshold and exclusive prefix sum > top_p are all monotone along the sorted order).
struct Frontier {
  uint32_t key;        // key_of(frontier probability)
  float prob;
  int* unusable;       // LDS flag, cleared by the caller
};

__device__ void sample_row(const SampleParams& p, const int row, RowSmem& sm, const int n_preloaded, const Frontier* fr = nullptr) {
  const int tid = threadIdx.x;
  const float* x = p.probs + static_cast<int64_t>(row) * p.row_stride;
  const int V = p.V;
  const uint64_t seed = static_cast<uint64_t>(p.seeds[row]);
  const uint32_t pos = p.positions ? static_cast<uint32_t>(p.positions[row] & 0xffffffffll) : 0u;
  const uint32_t hpre = murmur_prefix(seed, pos);
  Best best{0.0, -1, 0};

  if (!p.filtered) {
    // sampler.py:744-747: multinomial_with_seed(torch.log(probs)) -- fp32 log, col = token id
    for (int i = tid; i < V; i += kT) {
      const double sc = static_cast<double>(logf(x[i])) + gumbel_from_hash(murmur_hash32(hpre, i));
      Best c{sc, i, i};
      if (best_better(best, c)) best = c;
    }
    best = block_best(best, sm.s_score, sm.s_rank, sm.s_tok);
    if (tid == 0) {
      p.out_ids[row] = best.rank < 0 ? 0 : best.token;
      if (p.out_n_keep) p.out_n_keep[row] = V;
    }
    return;
  }

  int64_t top_k = p.top_ks ? p.top_ks[row] : V;
  if (top_k > V) top_k = V;
  if (top_k < 0) top_k = 0;
  const float top_p = p.top_ps ? p.top_ps[row] : 1.0f;
  // The first radix level reads the row; when the bin the cut falls into and the bins above it hold <= kCand elements
  // (a softmax row with top-k 50 / top-p 0.9: a few hundred) they are collected into LDS in ONE more pass and the other
  // three levels, the min-p maximum and the compaction run on that list: two full-row passes instead of six.
  // (the select returns a SUPERSET prefix of the reference's kept set; the top-p rule itself is applied exactly below)
  const unsigned long long p_fix = top_p_fix_superset(top_p);
  const float* src = x;                              // what the rest of the selection reads: the row, or the candidates
  int n_src = V;
  bool from_cand = false;
  // A short preloaded candidate list is ranked WHOLE and the three rules of sampler.py:574-591 are applied to every ranked
  // element (rank < top_k; the exact top-p test; p >= max * min_p): the candidates are everything from the cut's first-level bin
  // upwards, i.e. a prefix of the sorted order that contains the kept set -- the radix levels 1-3 and the compaction only ever
  // shortened that prefix (a dozen workgroup barriers; 20 -> 7 us for the finish launch at [64, 128256], top-k 50 / top-p 0.9).
  const bool direct = n_preloaded >= 0 && n_preloaded <= kDirect;
  if (n_preloaded >= 0) {
    src = reinterpret_cast<const float*>(sm.cand_val);
    n_src = n_preloaded;
    from_cand = true;
    if (!direct) select_levels(src, n_src, 1, 4, top_k, p_fix, sm);
  } else {
    select_begin(sm);
    select_levels(x, V, 0, 1, top_k, p_fix, sm);
  }
  if (n_preloaded < 0 && !sm.done_all) {
    const int bin = sm.found_bin;
    const long long n_cand = static_cast<long long>(sm.c_above) + sm.hist_cnt[bin];
    __syncthreads();                                 // (hist_cnt is rewritten by the next level)
    if (n_cand <= kCand) {
      collect_candidates(x, V, bin, sm);
      src = reinterpret_cast<const float*>(sm.cand_val);
      n_src = static_cast<int>(n_cand);
      from_cand = true;
    }
    select_levels(src, n_src, 1, 4, top_k, p_fix, sm);
  }
  Select sel{0u, 0};
  if (!direct) sel = select_result(sm);

  uint32_t min_key = 0;
  if (p.min_ps) {
    // sampler.py:590-591: threshold = probs_sort[:, 0] * min_p  (fp32), drop p < threshold (the row maximum is a candidate)
    float mx = 0.f;
    const int V4 = vec4_len(src, n_src);
    for (int i = 4 * tid; i < V4; i += 4 * kT) {
      const float4 v = *reinterpret_cast<const float4*>(src + i);
      mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    for (int i = V4 + tid; i < n_src; i += kT) mx = fmaxf(mx, src[i]);
    mx = block_max(mx, sm.red);
    min_key = key_of(mx * p.min_ps[row]);
  }

  // small nuclei are compacted into LDS, larger ones into the global workspace
  uint32_t* gk0 = p.ws_keys ? p.ws_keys + static_cast<int64_t>(row) * 2 * V : nullptr;
  int32_t* gt0 = p.ws_toks ? p.ws_toks + static_cast<int64_t>(row) * 2 * V : nullptr;
  int n_keep;
  if (direct) {
    __syncthreads();
    for (int i = tid; i < n_src; i += kT) { sm.keys[i] = key_of(src[i]); sm.toks[i] = sm.cand_tok[i]; }
    n_keep = n_src;
    __syncthreads();
  } else {
    n_keep = compact_kept(src, n_src, sel, min_key, sm, [&](int posn, uint32_t key, int idx) {
      const int tok = from_cand ? sm.cand_tok[idx] : idx;
      if (posn < kLdsKeep) { sm.keys[posn] = key; sm.toks[posn] = tok; }
      if (gk0) { gk0[posn] = key; gt0[posn] = tok; }
    });
  }
  // The pairs are a prefix of the reference's sorted order (descending value, ties by token id) that contains its kept
  // set.  Rank them, then apply sampler.py:577-580 element by element on the sorted list: inclusive prefix sums in
  // double (what torch.cumsum computes on the CPU; a fixed tree here instead of its left-to-right loop: the two agree
  // to ~1e-16 relative, far inside the fl32 rounding the rule applies next), the rule's own fp32 arithmetic, and the
  // gumbel score of every surviving rank.  A zeroed element keeps its rank (the reference zeroes in place).
  int kept = 0;
  if (n_keep <= kLdsKeep) {
    // sorted copies alias the candidate list / the select's histograms (both dead by now)
    uint32_t* s_key = sm.cand_val;
    int* s_tok2 = sm.cand_tok;
    __syncthreads();
    uint32_t my_key[kLdsKeep / kT];
    int my_tok[kLdsKeep / kT], my_rank[kLdsKeep / kT];
#pragma unroll
    for (int u = 0; u < kLdsKeep / kT; ++u) {
      const int e = tid + u * kT;
      my_rank[u] = -1;
      if (e < n_keep) {
        const uint32_t ke = sm.keys[e];
        const int te = sm.toks[e];
        int rank = 0;
        for (int j = 0; j < n_keep; ++j) {
          const uint32_t kj = sm.keys[j];
          // ties by token id (for lists in token order that is `j < e`; sample_from_logits' lists are in no particular order)
          rank += (kj > ke) || (kj == ke && sm.toks[j] < te);
        }
        my_key[u] = ke; my_tok[u] = sm.toks[e]; my_rank[u] = rank;
      }
    }
    __syncthreads();                                  // (cand_tok was read by the compaction's emit until here)
#pragma unroll
    for (int u = 0; u < kLdsKeep / kT; ++u)
      if (my_rank[u] >= 0) { s_key[my_rank[u]] = my_key[u]; s_tok2[my_rank[u]] = my_tok[u]; }
    __syncthreads();
    // thread t owns sorted ranks 2t, 2t + 1
    constexpr int kPer = kLdsKeep / kT;
    float pv[kPer];
    double loc = 0.0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int r = tid * kPer + u;
      pv[u] = r < n_keep ? __uint_as_float(s_key[r]) : 0.f;
      loc += static_cast<double>(pv[u]);
    }
    const double incl = block_scan_incl(loc, sm.s_score);
    double run = incl - loc;
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int r = tid * kPer + u;
      if (fr) {
        // (workgroup-uniform branch) is r the first rank at or below the frontier?  ranks 0 .. n_keep all have an owner (n_keep < kLdsKeep)
        const bool below = r >= n_keep || s_key[r] <= fr->key;
        const bool prev_above = r == 0 || (r - 1 < n_keep && s_key[r - 1] > fr->key);
        if (r <= n_keep && below && prev_above) {
          const bool keep_h = r < top_k && fr->key >= min_key && exact_top_p_keep(run + static_cast<double>(fr->prob), fr->prob, top_p);
          if (keep_h) *fr->unusable = 1;
        }
      }
      run += static_cast<double>(pv[u]);
      const bool in_rules = !direct || (r < top_k && s_key[r < n_keep ? r : 0] >= min_key);     // (the select enforced both otherwise)
      if (r < n_keep && in_rules && exact_top_p_keep(run, pv[u], top_p)) {
        ++kept;
        const double sc = log(static_cast<double>(pv[u])) + gumbel_from_hash(murmur_hash32(hpre, r));
        Best c{sc, r, s_tok2[r]};
        if (best_better(best, c)) best = c;
      }
    }
  } else {
    if (!gk0) {   // no workspace: cannot rank -- refuse the row (-1) instead of mis-sampling it
      if (tid == 0) { p.out_ids[row] = -1; if (p.out_n_keep) p.out_n_keep[row] = n_keep; }
      return;
    }
    __threadfence_block();
    __syncthreads();
    radix_sort_desc(gk0, gt0, gk0 + V, gt0 + V, n_keep, sm);
    // thread t owns the contiguous sorted ranks [t seg, (t + 1) seg)
    const int seg = (n_keep + kT - 1) / kT;
    const int r0 = tid * seg, r1 = r0 + seg < n_keep ? r0 + seg : n_keep;
    double loc = 0.0;
    for (int r = r0; r < r1; ++r) loc += static_cast<double>(__uint_as_float(gk0[r]));
    const double incl = block_scan_incl(loc, sm.s_score);
    double run = incl - loc;
    for (int r = r0; r < r1; ++r) {
      const float pr = __uint_as_float(gk0[r]);
      run += static_cast<double>(pr);
      if (exact_top_p_keep(run, pr, top_p)) {
        ++kept;
        const double sc = log(static_cast<double>(pr)) + gumbel_from_hash(murmur_hash32(hpre, r));
        Best c{sc, r, gt0[r]};
        if (best_better(best, c)) best = c;
      }
    }
  }
  if (p.out_n_keep) {
    const int total = static_cast<int>(block_sum(static_cast<float>(kept), sm.red) + 0.5f);   // (< 2^24: exact in fp32)
    if (tid == 0) p.out_n_keep[row] = total;
  }
  best = block_best(best, sm.s_score, sm.s_rank, sm.s_tok);
  if (tid == 0) {
    int tok = best.token;
    if (best.rank < 0) {
      // empty nucleus (top_k == 0): the reference's all -inf row argmax-es to sorted rank 0
      tok = 0;
    }
    p.out_ids[row] = tok;
  }
}

// rank-0 token of an EMPTY nucleus (top_k == 0: the reference's all -inf row argmax-es to sorted rank 0) = the arg max of the row
__device__ void fixup_empty_nucleus(const SampleParams& p, const int row, RowSmem& sm) {
  if (!p.top_ks || p.top_ks[row] > 0) return;          // (workgroup-uniform)
  const int tid = threadIdx.x;
  const float* x = p.probs + static_cast<int64_t>(row) * p.row_stride;
  Best best{0.0, -1, 0};
  for (int i = tid; i < p.V; i += kT) {
    Best c{static_cast<double>(x[i]), i, i};
    if (best_better(best, c)) best = c;
  }
  __syncthreads();
  best = block_best(best, sm.s_score, sm.s_rank, sm.s_tok);
  if (tid == 0) p.out_ids[row] = best.token;
}

__global__ __launch_bounds__(kT) void sample_kernel(SampleParams p) {
  __shared__ RowSmem sm;
  sample_row(p, blockIdx.x, sm, -1);
}

// ---- the filtered case for decode-sized batches: the two full-row passes cut into column ranges over the whole chip -----
// One workgroup per row keeps 64 of 256 CUs busy and reads its 0.5 MB row twice at one CU's rate (75 us for [64, 128256],
// top-k 50 / top-p 0.9).  Three launches instead:
//   hist     grid (ranges, rows): the first radix level's count + fixed-point sum histogram of every column range -> workspace
//            (integer and 2^-40 fixed-point sums: the merged histogram is the single-workgroup one bit for bit);
//   collect  grid (ranges, rows): every workgroup merges its row's range histograms, takes the level's decision (the same code,
//            select_decide), and writes its range's candidates -- every element of the cut's bin and the bins above it -- into
//            the row's candidate list at the offset the range histograms give it: the list is in token order;
//   finish   grid (rows): the candidates (<= kCand) go to LDS and the row is finished as before: levels 1-3, min-p, compaction,
//            ranking, the exact top-p rule, the gumbel arg-max.  Rows the shortcut does not cover (everything kept at level 0,
//            more than kCand candidates) run the whole single-workgroup routine there.
// Same ids and kept counts as sample_kernel by construction.
struct RangeWs {
  uint32_t* cnt;               // [B, S, 256]
  unsigned long long* sum;     // [B, S, 256]
  uint32_t* cand_val;          // [B, kCand]
  int* cand_tok;               // [B, kCand]
  int* state;                  // [B, 8]: 0 shortcut usable, 1 found bin, 2 c_above, 3 n_cand, 4/5 s_above lo / hi
};

__host__ __device__ inline int64_t range_ws_bytes(int64_t batch, int splits) {
  return batch * splits * 256 * 12 + batch * kCand * 8 + batch * 32;
}
__host__ __device__ inline RangeWs range_ws_view(void* base, int64_t batch, int splits) {
  RangeWs w;
  unsigned char* b = static_cast<unsigned char*>(base);
  w.sum = reinterpret_cast<unsigned long long*>(b);                 b += batch * splits * 256 * 8;
  w.cnt = reinterpret_cast<uint32_t*>(b);                           b += batch * splits * 256 * 4;
  w.cand_val = reinterpret_cast<uint32_t*>(b);                      b += batch * kCand * 4;
  w.cand_tok = reinterpret_cast<int*>(b);                           b += batch * kCand * 4;
  w.state = reinterpret_cast<int*>(b);
  return w;
}

__device__ __forceinline__ void sample_range(int V, int splits, int* begin, int* end) {
  const int per = ((V + splits - 1) / splits + 3) / 4 * 4;
  int b = per * static_cast<int>(blockIdx.x), e = b + per;
  if (b > V) b = V;
  if (e > V) e = V;
  *begin = b; *end = e;
}

__global__ __launch_bounds__(kT) void sample_hist_ranges_kernel(SampleParams p, int splits, RangeWs ws) {
  __shared__ uint32_t part_cnt[kHistCopies * kHistStride];
  __shared__ unsigned long long part_sum[kHistCopies * kHistStride];
  const int row = blockIdx.y, tid = threadIdx.x;
  const float* x = p.probs + static_cast<int64_t>(row) * p.row_stride;
  int b, e;
  sample_range(p.V, splits, &b, &e);
  for (int z = tid; z < kHistCopies * kHistStride; z += kT) { part_cnt[z] = 0; part_sum[z] = 0; }
  __syncthreads();
  const int copy = (tid & (kHistCopies - 1)) * kHistStride;
  auto tally = [&](float v) {
    const int bin = copy + (key_of(v) >> 24);
    atomicAdd(&part_cnt[bin], 1u);
    atomicAdd(&part_sum[bin], static_cast<unsigned long long>(to_fix(v)));
  };
  const int e4 = vec4_len(x, p.V) ? b + (e - b) / 4 * 4 : b;       // (ranges start at multiples of four)
  for (int i = b + 4 * tid; i < e4; i += 4 * kT) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    tally(v.x); tally(v.y); tally(v.z); tally(v.w);
  }
  for (int i = e4 + tid; i < e; i += kT) tally(x[i]);
  __syncthreads();
  if (tid < 256) {
    uint32_t c = 0;
    unsigned long long sfix = 0;
#pragma unroll
    for (int k = 0; k < kHistCopies; ++k) { c += part_cnt[k * kHistStride + tid]; sfix += part_sum[k * kHistStride + tid]; }
    const int64_t o = (static_cast<int64_t>(row) * splits + blockIdx.x) * 256 + tid;
    ws.cnt[o] = c;
    ws.sum[o] = sfix;
  }
}

// what the collect launch needs of RowSmem: the level's histogram, the select state, two scratch rows (3 KiB instead of 87: two
// workgroups per CU)
struct CollectSmem {
  uint32_t hist_cnt[256];
  unsigned long long hist_sum[256];
  int wave_a[kNW];
  float red[16];
  uint32_t prefix;
  int c_above;
  unsigned long long s_above;
  int found_bin;
  int n_eq_keep;
  int done_all;
};

__global__ __launch_bounds__(kT) void sample_collect_ranges_kernel(SampleParams p, int splits, RangeWs ws) {
  __shared__ CollectSmem sm;
  const int row = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* x = p.probs + static_cast<int64_t>(row) * p.row_stride;
  const int V = p.V;
  int64_t top_k = p.top_ks ? p.top_ks[row] : V;
  if (top_k > V) top_k = V;
  if (top_k < 0) top_k = 0;
  const float top_p = p.top_ps ? p.top_ps[row] : 1.0f;
  const unsigned long long p_fix = top_p_fix_superset(top_p);
  const int64_t hrow = static_cast<int64_t>(row) * splits * 256;
  select_begin(sm);
  uint32_t mine_before = 0;           // (tid < 256) elements of bin `tid` in the ranges ahead of this one
  if (tid < 256) {
    uint32_t c = 0;
    unsigned long long sfix = 0;
    for (int s = 0; s < splits; ++s) {
      const uint32_t cs = ws.cnt[hrow + s * 256 + tid];
      if (s < static_cast<int>(blockIdx.x)) mine_before += cs;
      c += cs;
      sfix += ws.sum[hrow + s * 256 + tid];
    }
    sm.hist_cnt[tid] = c;
    sm.hist_sum[tid] = sfix;
  }
  __syncthreads();
  select_decide(0, 24, 0u, top_k, p_fix, sm);
  __syncthreads();
  const int bin = sm.found_bin;
  const long long n_cand = sm.done_all ? 0 : static_cast<long long>(sm.c_above) + sm.hist_cnt[bin];
  const bool usable = !sm.done_all && n_cand <= kCand;
  if (blockIdx.x == 0 && tid == 0) {
    int* st = ws.state + static_cast<int64_t>(row) * 8;
    st[0] = usable ? 1 : 0;
    st[1] = bin;
    st[2] = sm.c_above;
    st[3] = static_cast<int>(n_cand);
    st[4] = static_cast<int>(sm.s_above & 0xffffffffull);
    st[5] = static_cast<int>(sm.s_above >> 32);
  }
  if (!usable) return;
  // where this range's candidates start in the row's list: the candidates of the ranges ahead of it
  float before_f = (tid < 256 && tid >= bin) ? static_cast<float>(mine_before) : 0.f;       // (<= kCand: exact in fp32)
  int base = static_cast<int>(block_sum(before_f, sm.red) + 0.5f);
  uint32_t* out_val = ws.cand_val + static_cast<int64_t>(row) * kCand;
  int* out_tok = ws.cand_tok + static_cast<int64_t>(row) * kCand;
  int b, e;
  sample_range(V, splits, &b, &e);
  const uint32_t lo = static_cast<uint32_t>(bin) << 24;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int e4 = vec4_len(x, V) ? b + (e - b) / 4 * 4 : b;
  for (int i0 = b; i0 < e4; i0 += 4 * kT) {
    const int i = i0 + 4 * tid;
    float val[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < e4) {
      const float4 v = *reinterpret_cast<const float4*>(x + i);
      val[0] = v.x; val[1] = v.y; val[2] = v.z; val[3] = v.w;
    }
    bool in[4];
    int before = 0, wave = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      in[j] = i < e4 && key_of(val[j]) >= lo;
      const unsigned long long bb = __ballot(in[j]);
      before += __popcll(bb & lt_mask);
      wave += __popcll(bb);
    }
    __syncthreads();
    if (lane == 0) sm.wave_a[wid] = wave;
    __syncthreads();
    int o = 0, t = 0;
    for (int w = 0; w < kNW; ++w) {
      const int a = sm.wave_a[w];
      if (w < wid) o += a;
      t += a;
    }
    int pos = base + o + before;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (in[j]) { out_val[pos] = __float_as_uint(val[j]); out_tok[pos] = i + j; ++pos; }
    base += t;
  }
  for (int i0 = e4; i0 < e; i0 += kT) {
    const int i = i0 + tid;
    const float v = i < e ? x[i] : 0.f;
    const bool in = i < e && key_of(v) >= lo;
    const unsigned long long bb = __ballot(in);
    __syncthreads();
    if (lane == 0) sm.wave_a[wid] = __popcll(bb);
    __syncthreads();
    int o = 0, t = 0;
    for (int w = 0; w < kNW; ++w) {
      const int a = sm.wave_a[w];
      if (w < wid) o += a;
      t += a;
    }
    if (in) { const int pos = base + o + __popcll(bb & lt_mask); out_val[pos] = __float_as_uint(v); out_tok[pos] = i; }
    base += t;
  }
}

__global__ __launch_bounds__(kT) void sample_finish_ranges_kernel(SampleParams p, RangeWs ws) {
  __shared__ RowSmem sm;
  const int row = blockIdx.x, tid = threadIdx.x;
  // ONE round trip: the row's state record and this thread's share of the candidate list (whatever its length: entries behind
  // the list's end are never looked at) travel together
  const int4 st0 = *reinterpret_cast<const int4*>(ws.state + static_cast<int64_t>(row) * 8);          // usable, bin, c_above, n_cand
  const int2 st1 = *reinterpret_cast<const int2*>(ws.state + static_cast<int64_t>(row) * 8 + 4);      // s_above lo, hi
  uint32_t cv[kCand / kT];
  int ct[kCand / kT];
#pragma unroll
  for (int u = 0; u < kCand / kT; ++u) {
    cv[u] = ws.cand_val[static_cast<int64_t>(row) * kCand + tid + u * kT];
    ct[u] = ws.cand_tok[static_cast<int64_t>(row) * kCand + tid + u * kT];
  }
  if (!st0.x) {               // (workgroup-uniform) the whole routine, from the row
    sample_row(p, row, sm, -1);
    fixup_empty_nucleus(p, row, sm);
    return;
  }
  const int n_cand = st0.w;
#pragma unroll
  for (int u = 0; u < kCand / kT; ++u)
    if (tid + u * kT < n_cand) { sm.cand_val[tid + u * kT] = cv[u]; sm.cand_tok[tid + u * kT] = ct[u]; }
  if (tid == 0) {
    sm.prefix = static_cast<uint32_t>(st0.y) << 24;
    sm.found_bin = st0.y;
    sm.c_above = st0.z;
    sm.s_above = (static_cast<unsigned long long>(static_cast<uint32_t>(st1.y)) << 32) | static_cast<uint32_t>(st1.x);
    sm.done_all = 0;
    sm.n_eq_keep = 0;
  }
  __syncthreads();
  sample_row(p, row, sm, n_cand);
  fixup_empty_nucleus(p, row, sm);
}

// ---- the filtered case straight from bf16 logits: the probabilities of a row are never written -------------------------------
// Sampler.forward on a decode batch used to be five launches over the [B, V] fp32 probabilities (softmax partials, normalise:
// 33 MB written; histogram + collect: 33 MB read each; finish).  The kept set of sampler.py:574-591 is a prefix of the row's
// descending order, and p(x) = expf(x / t - max) / sum is monotone in the bf16 logit x, so the prefix can be found on the LOGITS:
//   candidates  grid (16 column ranges, rows), 256 threads: the softmax partials of the range (the bits of the two-launch softmax:
//               softmax_ranges.hpp) and an exact two-level radix select on the 16-bit order keys of the range's logits: every
//               element >= the range's K-th largest (K = min(top_k, 64)) goes to the range's candidate list in token order
//               (<= 127 of them: ties at the cut included), with the largest key NOT emitted -- the range's frontier;
//   finish      grid (rows), 1024 threads: merges the partials, turns the <= 16 x 127 candidates into probabilities (the same
//               function of the same inputs as the normalise launch: the same bits), and runs the row routine's direct form on
//               them (rank all, the three rules element by element, fp64 gumbel arg-max) with the frontier test of sample_row;
//   a row the candidates cannot decide (the rules reach the frontier: flat rows, top-p without top-k over a wide nucleus;
//   more than 127 ties; top_k == 0; a non-finite softmax) is redone the long way by the same workgroup: it writes the row's
//   probabilities into a scratch matrix and runs the single-workgroup routine on them (no launch for rows that do not need it).
// [64, 128256], top-k 50 / top-p 0.9: 16 MB read once from HBM instead of ~130 MB of traffic.  Same ids and kept counts as the
// long way by construction (tests/test_sampler_gpu.py compares them on every row).
constexpr int kFastRangesMax = 16;
constexpr int kFastCap = 127;
constexpr int kFastK = 64;
static_assert(kFastRangesMax * kFastCap < kDirect, "the finish launch ranks the whole list in LDS, with one spare rank");

struct FastWs {
  float* partials;      // [B, S, 2]
  uint32_t* cand_val;   // [B, R, kFastCap]: the logit as fp32 bits
  int* cand_tok;        // [B, R, kFastCap]
  int* meta;            // [B, R, 4]: candidates (-1: more than kFastCap), 1 if columns were left out, the largest of them (fp32 bits), 0
  int* fallback;        // [B]: 1 = redo the row from its probabilities
};
__host__ __device__ inline int64_t fast_ws_bytes(int64_t batch, int splits) {
  return batch * splits * 8 + batch * kFastRangesMax * kFastCap * 8 + batch * kFastRangesMax * 16 + batch * 4;
}
__host__ __device__ inline FastWs fast_ws_view(void* base, int64_t batch, int splits) {
  FastWs w;
  unsigned char* b = static_cast<unsigned char*>(base);
  w.partials = reinterpret_cast<float*>(b);         b += batch * splits * 8;
  w.cand_val = reinterpret_cast<uint32_t*>(b);      b += batch * kFastRangesMax * kFastCap * 4;
  w.cand_tok = reinterpret_cast<int*>(b);           b += batch * kFastRangesMax * kFastCap * 4;
  w.meta = reinterpret_cast<int*>(b);               b += batch * kFastRangesMax * 16;
  w.fallback = reinterpret_cast<int*>(b);
  return w;
}

// fp32 bit pattern -> the top 16 bits of a key that orders like the value (-0 < +0; NaN rows never get here: their softmax is
// not finite).  bf16 logits are their fp32 widening: the 16 bits ARE the whole key; for genuine fp32 logits columns that agree in
// them are selected together (a superset still).
__device__ __forceinline__ uint32_t order_hi16(uint32_t fbits) { return (fbits ^ ((fbits & 0x80000000u) ? 0xffffffffu : 0x80000000u)) >> 16; }
// the smallest fp32 value whose key is `key16` (every column with a key >= key16 compares >= it)
__device__ __forceinline__ float floor_of_key16(uint32_t key16) {
  const uint32_t k = key16 << 16;
  return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// four consecutive columns of a row of logits in registers
template <typename IN>
struct Cols4;
template <>
struct Cols4<uint16_t> {
  uint2 q;
  __device__ __forceinline__ void load(const uint16_t* p) { q = *reinterpret_cast<const uint2*>(p); }
  __device__ __forceinline__ void zero() { q = uint2{0u, 0u}; }
  __device__ __forceinline__ float v(int k) const {
    return __uint_as_float(k == 0 ? (q.x << 16) : k == 1 ? (q.x & 0xffff0000u) : k == 2 ? (q.y << 16) : (q.y & 0xffff0000u));
  }
};
template <>
struct Cols4<float> {
  float4 q;
  __device__ __forceinline__ void load(const float* p) { q = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void zero() { q = float4{0.f, 0.f, 0.f, 0.f}; }
  __device__ __forceinline__ float v(int k) const { return k == 0 ? q.x : k == 1 ? q.y : k == 2 ? q.z : q.w; }
};

// which of 256 bins holds the K-th largest element: `c` = this thread's (tid = bin) count.  Returns through LDS `sel`:
// [0] bin, [1] elements in the bins above it, [2] elements in it.  Needs K >= 1 and K <= the total.  256 threads.
__device__ __forceinline__ void select_bin_256(uint32_t c, uint32_t K, uint32_t* wave_tot, int* sel) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  uint32_t v = c;                                   // inclusive suffix sum over the wave's 64 bins
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_down(v, off, 64);
    if (lane + off < 64) v += o;
  }
  if (lane == 0) wave_tot[wid] = v;
  __syncthreads();
  uint32_t above = 0;
  for (int w = wid + 1; w < 4; ++w) above += wave_tot[w];
  const uint32_t incl = v + above, excl = incl - c;
  if (incl >= K && excl < K) { sel[0] = tid; sel[1] = static_cast<int>(excl); sel[2] = static_cast<int>(c); }
  __syncthreads();
}

constexpr int kFastSlots = 16;     // 1024-column steps of a workgroup's columns (4 per thread and step): <= 16384 columns

// A workgroup's columns live in REGISTERS (one round trip to L2 / HBM for the whole range: a pass over them from memory cost
// ~3 us each, and the first version made seven).  Slot s = (softmax range g = s / J of the workgroup's G, step j = s % J): the 4
// columns b_g + 1024 j + 4 tid ..; the row's last range may end in <= 3 more columns (thread tid < 3 holds one).  NS = slots
// compiled in (8 covers [*, 128256] at 16 candidate ranges).
template <typename IN, int NS>
__global__ __launch_bounds__(kSplitThreads) void sample_logit_candidates_kernel(const IN* __restrict__ logits, int64_t row_stride,
                                                                                const float* __restrict__ temperatures, int V, int S, int G, int J,
                                                                                const int32_t* __restrict__ top_ks, FastWs ws) {
  __shared__ float red[4][4];
  __shared__ int redi[4];
  __shared__ alignas(16) uint32_t hist[4][257];
  __shared__ uint32_t wave_tot[4];
  __shared__ int sel[4];
  const int row = blockIdx.y, c = blockIdx.x, R = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const IN* x = logits + static_cast<int64_t>(row) * row_stride;
  const float t = temperatures[row];
  const int per = ((V + S - 1) / S + 3) / 4 * 4;    // (split_range_of's geometry)
  // ---- one burst: every column of the workgroup ----
  Cols4<IN> q[NS];
  uint32_t on_mask = 0;                              // slot s holds four columns of this thread
  auto slot_g = [&](int s) { return s / J; };        // (uniform)
  auto slot_col = [&](int s) {
    const int g = s / J, j = s - g * J;
    int64_t b = static_cast<int64_t>(per) * (c * G + g);
    if (b > V) b = V;
    return static_cast<int>(b) + 1024 * j + 4 * tid;
  };
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int g = s / J;
    int64_t b = static_cast<int64_t>(per) * (c * G + g), e = b + per;
    if (b > V) b = V;
    if (e > V) e = V;
    const int64_t e4 = b + (e - b) / 4 * 4;
    const int i = slot_col(s);
    const bool on = g < G && i < e4;
    on_mask |= on ? (1u << s) : 0u;
    if (on) q[s].load(x + i);
    else q[s].zero();
  }
  // (the <= 3 columns behind the last multiple of four: in the row's last range only, which is this workgroup's last if any)
  const int tail_g = G - 1;
  int tail_i = -1;
  float tail_v = 0.f;
  {
    int64_t b = static_cast<int64_t>(per) * (c * G + tail_g), e = b + per;
    if (b > V) b = V;
    if (e > V) e = V;
    const int64_t e4 = b + (e - b) / 4 * 4;
    if (e4 + tid < e) { tail_i = static_cast<int>(e4 + tid); tail_v = ld1(x + tail_i); }
  }
  auto val_of = [&](int s, int k) -> float { return q[s].v(k); };
  auto is_on = [&](int s) { return (on_mask >> s) & 1u; };

  // ---- the softmax partials of the G ranges: range_partial's arithmetic in range_partial's order (softmax_ranges.hpp), every
  // range's two workgroup reductions done side by side.  The maximum of x / t is (the maximum of x) / t for t > 0 -- a correctly
  // rounded division is monotone -- so the first pass divides once per range, not once per column. ----
  const bool t_pos = t > 0.f;                        // (uniform; other temperatures: column by column, and the row goes the long way)
  const bool t_one = t == 1.0f;                      // (uniform) x / 1 is x: the exponent pass skips 32 divisions
  float mxg[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    float m;
    if (t_pos) m = fmaxf(fmaxf(val_of(s, 0), val_of(s, 1)), fmaxf(val_of(s, 2), val_of(s, 3)));
    else m = fmaxf(fmaxf(val_of(s, 0) / t, val_of(s, 1) / t), fmaxf(val_of(s, 2) / t, val_of(s, 3) / t));
    const int g = is_on(s) ? slot_g(s) : -1;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) mxg[gg] = g == gg ? fmaxf(mxg[gg], m) : mxg[gg];
  }
  if (tail_i >= 0) {
    const float m = t_pos ? tail_v : tail_v / t;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) mxg[gg] = tail_g == gg ? fmaxf(mxg[gg], m) : mxg[gg];
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float w = wave_max(mxg[g]);
    if (lane == 0) red[g][wid] = w;
  }
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float m = fmaxf(fmaxf(red[g][0], red[g][1]), fmaxf(red[g][2], red[g][3]));
    mxg[g] = t_pos ? m / t : m;
  }
  __syncthreads();
  float smg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (is_on(s)) {
      const int g = slot_g(s);
      const float mx = g == 0 ? mxg[0] : g == 1 ? mxg[1] : g == 2 ? mxg[2] : mxg[3];
      if (mx > -INFINITY) {
        float add;
        if (t_one) add = expf(val_of(s, 0) - mx) + expf(val_of(s, 1) - mx) + expf(val_of(s, 2) - mx) + expf(val_of(s, 3) - mx);
        else add = expf(val_of(s, 0) / t - mx) + expf(val_of(s, 1) / t - mx) + expf(val_of(s, 2) / t - mx) + expf(val_of(s, 3) / t - mx);
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) smg[gg] = g == gg ? smg[gg] + add : smg[gg];
      }
    }
  }
  if (tail_i >= 0) {
    const float mx = tail_g == 0 ? mxg[0] : tail_g == 1 ? mxg[1] : tail_g == 2 ? mxg[2] : mxg[3];
    if (mx > -INFINITY) {
      const float add = expf(tail_v / t - mx);
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) smg[gg] = tail_g == gg ? smg[gg] + add : smg[gg];
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float w = wave_sum(smg[g]);
    if (lane == 0) red[g][wid] = w;
  }
  __syncthreads();
  if (tid < G) {
    // block_sum's association over the four wave sums: (w0 + w2) + (w1 + w3)
    const float tot = (red[tid][0] + red[tid][2]) + (red[tid][1] + red[tid][3]);
    const float mx = tid == 0 ? mxg[0] : tid == 1 ? mxg[1] : tid == 2 ? mxg[2] : mxg[3];
    ws.partials[(static_cast<int64_t>(row) * S + c * G + tid) * 2 + 0] = mx;
    ws.partials[(static_cast<int64_t>(row) * S + c * G + tid) * 2 + 1] = tot;
  }

  // ---- candidates ----
  int* meta = ws.meta + (static_cast<int64_t>(row) * R + c) * 4;
  int64_t top_k = top_ks ? top_ks[row] : V;
  if (top_k > V) top_k = V;
  int n;
  {
    int64_t cb = static_cast<int64_t>(per) * (c * G), ce = static_cast<int64_t>(per) * (c * G + G);
    if (cb > V) cb = V;
    if (ce > V) ce = V;
    n = static_cast<int>(ce - cb);
  }
  if (n <= 0 || top_k <= 0) {                       // (top_k <= 0: the finish launch sends the row the long way)
    if (tid == 0) { meta[0] = 0; meta[1] = 0; meta[2] = 0; }
    return;
  }
  const int Kc = static_cast<int>(top_k < kFastK ? top_k : kFastK);
  const uint32_t K = static_cast<uint32_t>(Kc < n ? Kc : n);
  auto for_each_val = [&](auto&& f) {               // every column this thread holds (registers): f(fp32 value of the logit, column)
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (is_on(s)) {
        const int col = slot_col(s);
        f(q[s].v(0), col); f(q[s].v(1), col + 1); f(q[s].v(2), col + 2); f(q[s].v(3), col + 3);
      }
    if (tail_i >= 0) f(tail_v, tail_i);
  };
  auto for_each = [&](auto&& f) {                   // the same as (16-bit order key, column)
    for_each_val([&](float v, int col) { f(order_hi16(__float_as_uint(v)), col); });
  };
  const int n_mine = 4 * __popc(on_mask) + (tail_i >= 0 ? 1 : 0);
  // A lower bound of the K-th largest logit without a histogram: every wave takes the ceil(K / 4)-th largest of its 64 per-lane
  // maxima (a 16-step bit search on the 16-bit order keys with ballots), the bound is the smallest of the four: K distinct columns
  // are >= it.  For rows in any order but an adversarial one ~K .. 2 K columns pass it: they ARE the candidates, a superset of
  // everything >= the K-th largest.  (Histogramming the range cost 13 us of LDS atomics on the handful of hot bins of a softmax
  // row.)  Columns are compared as fp32 values (-0 == +0 there: a superset still; a NaN passes no test -- its row's softmax is
  // not finite and the finish launch sends it the long way).
  float my_maxf = -INFINITY;
  for_each_val([&](float v, int) { my_maxf = fmaxf(my_maxf, v); });
  const int my_max = n_mine > 0 ? static_cast<int>(order_hi16(__float_as_uint(my_maxf))) : -1;
  {
    const int kw = (static_cast<int>(K) + 3) / 4;
    uint32_t pre = 0;                               // largest v with #(lane maxima >= v) >= kw  (0 if fewer than kw lanes hold columns)
#pragma unroll
    for (int bit = 15; bit >= 0; --bit) {
      const uint32_t cand = pre | (1u << bit);
      const int cnt = __popcll(__ballot(my_max >= static_cast<int>(cand)));
      pre = cnt >= kw ? cand : pre;
    }
    if (lane == 0) redi[wid] = static_cast<int>(pre);
  }
  if (tid == 0) sel[3] = 0;                          // the emit counter
  __syncthreads();
  int bound = redi[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) bound = redi[w] < bound ? redi[w] : bound;
  uint32_t kth = static_cast<uint32_t>(bound);
  uint32_t* out_val = ws.cand_val + (static_cast<int64_t>(row) * R + c) * kFastCap;
  int* out_tok = ws.cand_tok + (static_cast<int64_t>(row) * R + c) * kFastCap;
  // ---- emit (in no particular order: the finish launch ranks by (value, token)); the largest value left out is the frontier.
  // Positions: hits per lane, an exclusive scan over the wave, ONE LDS atomic per wave ----
  float front_f = -INFINITY;
  bool any_left = false;
  auto emit = [&]() {
    const float L = kth == 0u ? -INFINITY : floor_of_key16(kth);
    front_f = -INFINITY;
    int hits = 0;
    for_each_val([&](float v, int) {
      hits += v >= L;
      front_f = fmaxf(front_f, v < L ? v : -INFINITY);
    });
    any_left = hits < n_mine;
    int incl = hits;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    const int total = __builtin_amdgcn_readlane(incl, 63);
    int base = 0;
    if (total > 0) {                                  // (wave-uniform)
      if (lane == 0) base = atomicAdd(&sel[3], total);
      base = __builtin_amdgcn_readfirstlane(base);
    }
    int pos = base + incl - hits;
    if (hits > 0) {
      for_each_val([&](float v, int col) {
        if (v >= L) {
          if (pos < kFastCap) { out_val[pos] = __float_as_uint(v); out_tok[pos] = col; }
          ++pos;
        }
      });
    }
  };
  emit();
  __syncthreads();
  if (sel[3] > kFastCap) {                          // (workgroup-uniform) too many pass the bound: the exact two-level select
    __syncthreads();
    for (int z = tid; z < 4 * 257; z += kSplitThreads) (&hist[0][0])[z] = 0;
    __syncthreads();
    for_each([&](uint32_t k16, int) { atomicAdd(&hist[wid][k16 >> 8], 1u); });
    __syncthreads();
    select_bin_256(hist[0][tid] + hist[1][tid] + hist[2][tid] + hist[3][tid], K, wave_tot, sel);
    const uint32_t b1 = static_cast<uint32_t>(sel[0]), above1 = static_cast<uint32_t>(sel[1]);
    __syncthreads();
    for (int z = tid; z < 4 * 257; z += kSplitThreads) (&hist[0][0])[z] = 0;
    __syncthreads();
    for_each([&](uint32_t k16, int) {
      if ((k16 >> 8) == b1) atomicAdd(&hist[wid][k16 & 255u], 1u);
    });
    __syncthreads();
    select_bin_256(hist[0][tid] + hist[1][tid] + hist[2][tid] + hist[3][tid], K - above1, wave_tot, sel);
    kth = (b1 << 8) | static_cast<uint32_t>(sel[0]);
    const int n_ge = static_cast<int>(above1) + sel[1] + sel[2];
    __syncthreads();
    if (n_ge > kFastCap) {                            // (workgroup-uniform) a tie wider than the list
      if (tid == 0) { meta[0] = -1; meta[1] = 0; meta[2] = 0; }
      return;
    }
    if (tid == 0) sel[3] = 0;
    __syncthreads();
    emit();
    __syncthreads();
  }
  // the largest value left out, over the workgroup (any_left: this thread left something out)
  float fr_w = any_left ? front_f : -INFINITY;
  int left_w = any_left ? 1 : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    fr_w = fmaxf(fr_w, __shfl_xor(fr_w, off, 64));
    left_w |= __shfl_xor(left_w, off, 64);
  }
  if (lane == 0) { red[0][wid] = fr_w; redi[wid] = left_w; }
  __syncthreads();
  if (tid == 0) {
    const float f = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    int4 m;
    m.x = sel[3] > kFastCap ? -1 : sel[3];            // (fp32 compares emit -0 with +0: the exact select's count is a key count)
    m.y = redi[0] | redi[1] | redi[2] | redi[3];
    m.z = static_cast<int>(__float_as_uint(f));
    m.w = 0;
    *reinterpret_cast<int4*>(meta) = m;
  }
}

// The row routine's direct form for a list of <= 64 survivors, by ONE wave (lane = list element): the same rules, the same
// arithmetic (exact_top_p_keep on fp64 inclusive prefix sums, fp64 gumbel scores, best_better) without a workgroup barrier --
// sample_row spends ~15 us on its dozen barriers and serial LDS loops for the ~50 elements a pruned list holds.
__device__ void finish_small_list(const SampleParams& p, const int row, const uint32_t* cand_val, const int* cand_tok, uint32_t* s_key, int* s_tok,
                                  const int n, int64_t top_k, const float top_p, const bool has_min_p, const float min_p, const uint64_t seed,
                                  const uint32_t pos, const Frontier* fr) {
  const int lane = threadIdx.x & 63;
  const uint32_t key = lane < n ? key_of(__uint_as_float(cand_val[lane])) : 0u;
  const int tok = lane < n ? cand_tok[lane] : 0x7fffffff;
  uint32_t min_key = 0;
  if (has_min_p) {
    float mxp = lane < n ? __uint_as_float(cand_val[lane]) : 0.f;
    mxp = wave_max(fmaxf(mxp, 0.f));
    min_key = key_of(mxp * min_p);
  }
  int rank = 0;                                       // descending value, ties by token id
#pragma unroll 8
  for (int j = 0; j < 64; ++j) {
    const uint32_t kj = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(key), j));
    const int tj = __builtin_amdgcn_readlane(tok, j);
    rank += (j < n) && ((kj > key) || (kj == key && tj < tok));
  }
  if (lane < n) { s_key[rank] = key; s_tok[rank] = tok; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int r = lane;                                 // lane r owns sorted rank r
  const uint32_t kr = r < n ? s_key[r] : 0u;
  const int tr = r < n ? s_tok[r] : 0;
  const float pv = r < n ? __uint_as_float(kr) : 0.f;
  double run = static_cast<double>(pv);               // inclusive prefix sums in lane order
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double o = __shfl_up(run, off, 64);
    if (lane >= off) run += o;
  }
  if (fr) {
    // the first rank at or below the frontier: the elements above it are a prefix of the sorted order
    const int r0 = __popcll(__ballot(r < n && kr > fr->key));
    const double incl_prev = __shfl(run, r0 > 0 ? r0 - 1 : 0, 64);
    const double excl = r0 > 0 ? incl_prev : 0.0;
    const bool keep_h = r0 < top_k && fr->key >= min_key && exact_top_p_keep(excl + static_cast<double>(fr->prob), fr->prob, top_p);
    if (keep_h && lane == 0) *fr->unusable = 1;
  }
  const uint32_t hpre = murmur_prefix(seed, pos);
  Best best{0.0, -1, 0};
  const bool keep = r < n && r < top_k && kr >= min_key && exact_top_p_keep(run, pv, top_p);
  if (keep) {
    best.score = log(static_cast<double>(pv)) + gumbel_from_hash(murmur_hash32(hpre, r));
    best.rank = r;
    best.token = tr;
  }
  const int kept = __popcll(__ballot(keep));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Best o;
    o.score = __shfl_xor(best.score, off, 64);
    o.rank = __shfl_xor(best.rank, off, 64);
    o.token = __shfl_xor(best.token, off, 64);
    if (best_better(best, o)) best = o;
  }
  if (lane == 0) {
    p.out_ids[row] = best.rank < 0 ? 0 : best.token;   // (empty nucleus: the reference's all -inf row argmax-es to sorted rank 0)
    if (p.out_n_keep) p.out_n_keep[row] = kept;
  }
}

template <typename IN>
__global__ __launch_bounds__(kT) void sample_finish_fast_kernel(SampleParams p, const IN* __restrict__ logits, int64_t logits_row_stride,
                                                                 const float* __restrict__ temperatures, int S, int R, FastWs ws) {
  __shared__ RowSmem sm;
  __shared__ int pre[kFastRangesMax + 1];
  __shared__ int m_cnt[kFastRangesMax], m_left[kFastRangesMax];
  __shared__ float m_front[kFastRangesMax];
  __shared__ int s_has_front, s_bad, s_unusable;
  __shared__ float s_front;
  __shared__ float s_mx, s_sum;
  const int row = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  // ---- ONE round trip to memory for everything the row needs: its parameters, the R range records, the 2 S softmax partials and
  // this thread's <= 2 slots of the candidate lists (read whether filled or not: validity comes from the records) ----
  const float t = temperatures[row];
  int64_t top_k = p.top_ks ? p.top_ks[row] : p.V;
  const float top_p = p.top_ps ? p.top_ps[row] : 1.0f;
  const float min_p = p.min_ps ? p.min_ps[row] : 0.f;
  const uint64_t seed = static_cast<uint64_t>(p.seeds[row]);
  const uint32_t spos = p.positions ? static_cast<uint32_t>(p.positions[row] & 0xffffffffll) : 0u;
  uint32_t lb[2] = {0u, 0u};
  int tok[2] = {0, 0};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int idx = tid + u * kT;
    if (idx < R * kFastCap) {
      lb[u] = ws.cand_val[static_cast<int64_t>(row) * R * kFastCap + idx];
      tok[u] = ws.cand_tok[static_cast<int64_t>(row) * R * kFastCap + idx];
    }
  }
  if (tid >= 256 && tid < 512) { sm.hist_cnt[tid - 256] = 0; sm.hist_sum[tid - 256] = 0; sm.part_cnt[tid - 256] = 0; sm.part_sum[16 + tid - 256] = 0; }
  if (tid == 512) { sm.found_bin = -1; sm.n_eq_keep = -1; }
  if (tid >= 64 && tid < 64 + R) {
    const int4 m = *reinterpret_cast<const int4*>(ws.meta + (static_cast<int64_t>(row) * R + tid - 64) * 4);
    m_cnt[tid - 64] = m.x; m_left[tid - 64] = m.y; m_front[tid - 64] = __uint_as_float(static_cast<uint32_t>(m.z));
  }
  if (wid == 0) {
    // the row's maximum and sum from its S partials: merge_partials' arithmetic (softmax_ranges.hpp) with the S exponentials
    // side by side; the sum itself stays a left-to-right chain over the ranges (the bits of the normalise launch)
    const float pm = lane < S ? ws.partials[(static_cast<int64_t>(row) * S + lane) * 2] : -INFINITY;
    const float ps = lane < S ? ws.partials[(static_cast<int64_t>(row) * S + lane) * 2 + 1] : 0.f;
    const float mx = wave_max(pm);
    const float ex = pm > -INFINITY ? expf(pm - mx) : 0.f;
    float sum = 0.f;
    for (int s2 = 0; s2 < S; ++s2) {
      const float es = __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(ex)), s2)));
      const float ss = __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(ps)), s2)));
      const float ms = __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(pm)), s2)));
      if (ms > -INFINITY) sum = __fmaf_rn(ss, es, sum);
    }
    if (lane == 0) { s_mx = mx; s_sum = sum; }
  }
  __syncthreads();
  const float mx = s_mx, sum = s_sum;
  if (tid == 0) {
    int acc = 0, has = 0, bad = 0;
    float front = -INFINITY;
    for (int c = 0; c < R; ++c) {
      pre[c] = acc;
      if (m_cnt[c] < 0) bad = 1; else acc += m_cnt[c];
      if (m_left[c]) { has = 1; front = fmaxf(front, m_front[c]); }
    }
    pre[R] = acc;
    if (top_k <= 0 || acc == 0 || acc >= kDirect) bad = 1;
    if (!(t > 0.f) || !(sum > 0.f) || !(sum < INFINITY) || !(mx > -INFINITY) || !(mx < INFINITY)) bad = 1;
    s_front = front; s_has_front = has; s_bad = bad; s_unusable = 0;
  }
  __syncthreads();
  if (!s_bad) {                                         // (workgroup-uniform)
    if (top_k > p.V) top_k = p.V;
    const unsigned long long p_fix = top_p_fix_superset(top_p);
    bool have[2];
    uint32_t key16[2], pbits[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + u * kT;
      const int c = idx / kFastCap, j = idx - c * kFastCap;
      have[u] = c < R && j < m_cnt[c < R ? c : 0];
      key16[u] = have[u] ? order_hi16(lb[u]) : 0u;
      pbits[u] = have[u] ? __float_as_uint(softmax_prob(__uint_as_float(lb[u]), t, mx, sum)) : 0u;
    }
    // ---- prune to a prefix of the descending order that holds top_k elements or top_p (+ 2^-22) of the mass, whichever comes
    // first: a two-level select on the 16-bit logit keys with count and fixed-point mass histograms (order-independent).  The
    // candidates dropped here join the elements the ranges left out: the frontier rises to the largest of them. ----
    uint32_t cutkey = 0;
    bool found_any = true;
    {
      // (both levels' histograms were cleared before the kernel's first barrier: level 0 in hist_cnt / hist_sum, level 1 in
      // part_cnt[0..255] / part_sum[16..271]; three barriers per level)
      uint32_t base_c = 0;
      unsigned long long base_s = 0;
#pragma unroll
      for (int level = 0; level < 2; ++level) {
        uint32_t* h_cnt = level == 0 ? sm.hist_cnt : sm.part_cnt;
        unsigned long long* h_sum = level == 0 ? sm.hist_sum : sm.part_sum + 16;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bool in = have[u] && (level == 0 || (key16[u] >> 8) == (cutkey >> 8));
          if (in) {
            const int bin = level == 0 ? (key16[u] >> 8) : (key16[u] & 255u);
            atomicAdd(&h_cnt[bin], 1u);
            atomicAdd(&h_sum[bin], static_cast<unsigned long long>(to_fix(__uint_as_float(pbits[u]))));
          }
        }
        __syncthreads();
        uint32_t vc = 0;
        unsigned long long vs = 0;
        uint32_t my_c = 0;
        unsigned long long my_s = 0;
        if (tid < 256) {                                // (waves 0 .. 3 whole) inclusive suffix sums over the wave's 64 bins
          my_c = vc = h_cnt[tid];
          my_s = vs = h_sum[tid];
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const uint32_t oc = __shfl_down(vc, off, 64);
            const unsigned long long os = __shfl_down(vs, off, 64);
            if (lane + off < 64) { vc += oc; vs += os; }
          }
          if (lane == 0) { sm.wave_a[wid] = static_cast<int>(vc); sm.part_sum[wid] = vs; }
        }
        __syncthreads();
        if (tid < 256) {
          uint32_t ac = base_c;
          unsigned long long as = base_s;
          for (int w = wid + 1; w < 4; ++w) { ac += static_cast<uint32_t>(sm.wave_a[w]); as += sm.part_sum[w]; }
          const uint32_t incl_c = vc + ac, excl_c = incl_c - my_c;
          const unsigned long long incl_s = vs + as, excl_s = incl_s - my_s;
          const bool cond_incl = static_cast<int64_t>(incl_c) >= top_k || incl_s >= p_fix;
          const bool cond_excl = static_cast<int64_t>(excl_c) >= top_k || excl_s >= p_fix;
          if (cond_incl && !cond_excl) {
            if (level == 0) { sm.found_bin = tid; sm.c_above = static_cast<int>(excl_c); sm.s_above = excl_s; }
            else { sm.n_eq_keep = tid; }
          }
        }
        __syncthreads();
        const int fb = level == 0 ? sm.found_bin : sm.n_eq_keep;
        if (fb < 0) { found_any = false; break; }       // (uniform) level 0 only: the whole list is inside both limits
        cutkey = level == 0 ? (static_cast<uint32_t>(fb) << 8) : (cutkey | static_cast<uint32_t>(fb));
        base_c = static_cast<uint32_t>(sm.c_above);
        base_s = sm.s_above;
      }
      if (!found_any) cutkey = 0;
    }
    // ---- compact the survivors, in list order, into the row routine's candidate arrays ----
    bool keep[2];
    float dropped = s_has_front ? s_front : -INFINITY;   // the largest logit NOT in the list handed on: left out by a range, or dropped here
    int any_out = s_has_front;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      keep[u] = have[u] && key16[u] >= cutkey;
      if (have[u] && !keep[u]) { dropped = fmaxf(dropped, __uint_as_float(lb[u])); any_out = 1; }
    }
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long b0 = __ballot(keep[0]), b1 = __ballot(keep[1]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      dropped = fmaxf(dropped, __shfl_xor(dropped, off, 64));
      any_out |= __shfl_xor(any_out, off, 64);
    }
    __syncthreads();
    if (lane == 0) { sm.wave_a[wid] = __popcll(b0); sm.wave_b[wid] = __popcll(b1); sm.red[wid] = dropped; sm.s_rank[wid] = any_out; }
    __syncthreads();
    int off0 = 0, tot0 = 0, off1 = 0, n = 0, has_front_i = 0;
    float front = -INFINITY;
    for (int w = 0; w < kNW; ++w) {
      if (w < wid) { off0 += sm.wave_a[w]; off1 += sm.wave_b[w]; }
      tot0 += sm.wave_a[w];
      n += sm.wave_b[w];
      front = fmaxf(front, sm.red[w]);
      has_front_i |= sm.s_rank[w];
    }
    n += tot0;
    if (keep[0]) { const int pos = off0 + __popcll(b0 & lt_mask); sm.cand_val[pos] = pbits[0]; sm.cand_tok[pos] = tok[0]; }
    if (keep[1]) { const int pos = tot0 + off1 + __popcll(b1 & lt_mask); sm.cand_val[pos] = pbits[1]; sm.cand_tok[pos] = tok[1]; }
    Frontier fr;
    fr.unusable = &s_unusable;
    fr.prob = 0.f;
    fr.key = 0u;
    const bool has_front = has_front_i != 0;
    if (has_front) {
      fr.prob = softmax_prob(front, t, mx, sum);
      fr.key = key_of(fr.prob);
    }
    __syncthreads();
    // A long list that reaches neither limit (top-p over a wide nucleus without top-k) and has elements behind it: the frontier
    // test below would send the row the long way after an O(n^2) ranking of the whole list -- go there at once.
    if (!found_any && has_front && n > 256 && !p.min_ps) {
      if (tid == 0) s_unusable = 1;
    } else if (n <= 64) {                               // (workgroup-uniform) what a pruned list almost always is
      if (wid == 0)
        finish_small_list(p, row, sm.cand_val, sm.cand_tok, sm.keys, sm.toks, n, top_k, top_p, p.min_ps != nullptr, min_p, seed, spos,
                          has_front ? &fr : nullptr);
    } else {
      sample_row(p, row, sm, n, has_front ? &fr : nullptr);
    }
    __syncthreads();
  }
  const bool redo = s_bad || s_unusable;
  if (tid == 0) ws.fallback[row] = redo ? 1 : 0;
  if (!redo) return;
  // ---- the long way, for this row only: its probabilities into the scratch matrix (this workgroup writes the whole row and is the
  // only one to read it), then the row routine on them ----
  {
    const IN* x = logits + static_cast<int64_t>(row) * logits_row_stride;
    float* y = const_cast<float*>(p.probs) + static_cast<int64_t>(row) * p.row_stride;
    const int V4 = p.V & ~3;
    for (int i = 4 * tid; i < V4; i += 4 * kT) {
      float v[4];
      ld4<IN>(x + i, v);
      float4 o;
      o.x = softmax_prob(v[0], t, mx, sum); o.y = softmax_prob(v[1], t, mx, sum); o.z = softmax_prob(v[2], t, mx, sum); o.w = softmax_prob(v[3], t, mx, sum);
      *reinterpret_cast<float4*>(y + i) = o;
    }
    for (int i = V4 + tid; i < p.V; i += kT) y[i] = softmax_prob(ld1(x + i), t, mx, sum);
    __threadfence_block();
    __syncthreads();
  }
  sample_row(p, row, sm, -1);
  fixup_empty_nucleus(p, row, sm);
}

// ---- the unfiltered case for decode-sized batches: a row cut into column ranges over the whole chip -------------
// sampling_from_probs scores EVERY token (fp32 log + an fp64 gumbel each): one workgroup per row keeps 64 of 256 CUs busy
// (209 us for [64, 128256]).  Pass 1: grid (ranges, rows), every workgroup's best (score, token) of its range goes to the
// caller's workspace; pass 2: one wave per row merges them -- the same comparison (NaN maximal, first index on ties), so
// the token is the one the single-workgroup kernel returns.
struct PartialBest {
  double score;
  int rank;
  int token;
};

__global__ __launch_bounds__(kT) void sample_unfiltered_ranges_kernel(SampleParams p, int splits, PartialBest* __restrict__ partials) {
  __shared__ double s_score[kNW];
  __shared__ int s_rank[kNW];
  __shared__ int s_tok[kNW];
  const int row = blockIdx.y, tid = threadIdx.x;
  const float* x = p.probs + static_cast<int64_t>(row) * p.row_stride;
  const int V = p.V;
  const int per = ((V + splits - 1) / splits + 3) / 4 * 4;
  int b = per * static_cast<int>(blockIdx.x), e = b + per;
  if (b > V) b = V;
  if (e > V) e = V;
  const uint64_t seed = static_cast<uint64_t>(p.seeds[row]);
  const uint32_t pos = p.positions ? static_cast<uint32_t>(p.positions[row] & 0xffffffffll) : 0u;
  const uint32_t hpre = murmur_prefix(seed, pos);
  Best best{0.0, -1, 0};
  for (int i = b + tid; i < e; i += kT) {
    const double sc = static_cast<double>(logf(x[i])) + gumbel_from_hash(murmur_hash32(hpre, i));
    Best c{sc, i, i};
    if (best_better(best, c)) best = c;
  }
  best = block_best(best, s_score, s_rank, s_tok);
  if (tid == 0) {
    PartialBest o;
    o.score = best.score; o.rank = best.rank; o.token = best.token;
    partials[static_cast<int64_t>(row) * splits + blockIdx.x] = o;
  }
}

__global__ __launch_bounds__(64) void sample_unfiltered_merge_kernel(SampleParams p, int splits, const PartialBest* __restrict__ partials) {
  const int row = blockIdx.x, lane = threadIdx.x;
  Best r{0.0, -1, 0};
  if (lane < splits) {
    const PartialBest o = partials[static_cast<int64_t>(row) * splits + lane];
    r.score = o.score; r.rank = o.rank; r.token = o.token;
  }
  for (int off = 32; off > 0; off >>= 1) {
    Best o;
    o.score = __shfl_xor(r.score, off, 64);
    o.rank = __shfl_xor(r.rank, off, 64);
    o.token = __shfl_xor(r.token, off, 64);
    if (best_better(r, o)) r = o;
  }
  if (lane == 0) {
    p.out_ids[row] = r.rank < 0 ? 0 : r.token;
    if (p.out_n_keep) p.out_n_keep[row] = p.V;
  }
}

// rank-0 token of an empty nucleus must be the arg max of the row; handled by a tiny fix-up
// kernel so that the common path stays branch-free.
__global__ __launch_bounds__(kT) void empty_nucleus_fixup_kernel(SampleParams p) {
  __shared__ RowSmem sm;
  fixup_empty_nucleus(p, blockIdx.x, sm);
}

// ---- renormalisation (sgl_kernel.top_k_renorm_prob / top_p_renorm_prob,
//      sampler.py:753-762 top_p_normalize_probs_torch) -----------------------------
__global__ __launch_bounds__(kT) void renorm_kernel(const float* __restrict__ probs, float* __restrict__ out,
                                                     int64_t in_stride, int64_t out_stride, int V,
                                                     const int32_t* __restrict__ top_ks, int top_k_val,
                                                     const float* __restrict__ top_ps, float top_p_val) {
  __shared__ RowSmem sm;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* x = probs + static_cast<int64_t>(row) * in_stride;
  float* y = out + static_cast<int64_t>(row) * out_stride;
  int64_t top_k = top_ks ? top_ks[row] : top_k_val;
  if (top_k > V || top_k < 0) top_k = V;
  const float top_p = top_ps ? top_ps[row] : top_p_val;
  const Select sel = radix_select(x, V, top_k, top_p, sm);
  const uint32_t thr = sel.thr_key;
  // sum of the kept values (fixed thread mapping -> deterministic), ties kept by token order
  float part = 0.f;
  int base_eq = 0;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // pass A: mark + sum
  for (int i0 = 0; i0 < V; i0 += kT) {
    const int i = i0 + tid;
    const float v = i < V ? x[i] : 0.f;
    const uint32_t key = i < V ? key_of(v) : 0u;
    const bool gt = i < V && key > thr, eq = i < V && key == thr;
    const unsigned long long be = __ballot(eq);
    __syncthreads();
    if (lane == 0) sm.wave_b[wid] = __popcll(be);
    __syncthreads();
    int oe = 0, te = 0;
    for (int w = 0; w < kNW; ++w) { const int b = sm.wave_b[w]; if (w < wid) oe += b; te += b; }
    const bool keep = gt || (eq && (base_eq + oe + __popcll(be & lt_mask)) < sel.n_eq_keep);
    if (i < V) { y[i] = keep ? v : 0.f; part += keep ? v : 0.f; }
    base_eq += te;
  }
  const float tot = block_sum(part, sm.red);
  for (int i = tid; i < V; i += kT) y[i] = y[i] / tot;   // each thread re-reads its own writes
}

}  // namespace

extern "C" {

int sgl_amd_top_k_top_p_min_p_sample(const float* probs, int64_t row_stride, int64_t batch, int64_t vocab,
                                     const int32_t* top_ks, const float* top_ps, const float* min_ps,
                                     const int64_t* seeds, const int64_t* positions, int32_t* out_ids,
                                     void* ws_keys, void* ws_toks, int32_t* out_n_keep, int filtered,
                                     void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(vocab > 0 && vocab <= 0x7fffffffLL, "top_k_top_p_min_p_sample: bad vocab");
  SGL_CHECK_ARG(batch <= 0x7fffffffLL, "top_k_top_p_min_p_sample: batch too large");
  SGL_CHECK_ARG(seeds != nullptr, "top_k_top_p_min_p_sample: seeds are required (the caller draws them when sampling is not seeded)");
  SGL_CHECK_ARG((ws_keys == nullptr) == (ws_toks == nullptr), "top_k_top_p_min_p_sample: pass both workspaces or neither");
  if (batch == 0) return 0;
  SampleParams p;
  p.probs = probs; p.row_stride = row_stride; p.V = static_cast<int>(vocab);
  p.top_ks = top_ks; p.top_ps = top_ps; p.min_ps = min_ps; p.seeds = seeds; p.positions = positions;
  p.out_ids = out_ids; p.ws_keys = static_cast<uint32_t*>(ws_keys); p.ws_toks = static_cast<int32_t*>(ws_toks);
  p.out_n_keep = out_n_keep; p.filtered = filtered;
  // the unfiltered case of a decode-sized batch: column ranges over the whole chip (the workspace holds the partials)
  int splits = static_cast<int>(512 / batch);
  if (splits > 16) splits = 16;
  if (!filtered && ws_keys && splits >= 2 && vocab >= 4096 * splits && batch <= 65535) {
    PartialBest* partials = static_cast<PartialBest*>(ws_keys);
    hipLaunchKernelGGL(sample_unfiltered_ranges_kernel, dim3(splits, static_cast<unsigned>(batch)), dim3(kT), 0, as_stream(stream), p, splits,
                       partials);
    hipLaunchKernelGGL(sample_unfiltered_merge_kernel, dim3(batch), dim3(64), 0, as_stream(stream), p, splits, partials);
    SGL_CHECK_LAUNCH("top_k_top_p_min_p_sample(unfiltered ranges)");
    return 0;
  }
  hipLaunchKernelGGL(sample_kernel, dim3(batch), dim3(kT), 0, as_stream(stream), p);
  SGL_CHECK_LAUNCH("top_k_top_p_min_p_sample");
  if (filtered && top_ks) {
    hipLaunchKernelGGL(empty_nucleus_fixup_kernel, dim3(batch), dim3(kT), 0, as_stream(stream), p);
    SGL_CHECK_LAUNCH("top_k_top_p_min_p_sample(fixup)");
  }
  return 0;
}

int sgl_amd_sampling_lds_keep(void) { return kLdsKeep; }

int64_t sgl_amd_sample_ranges_workspace_bytes(int64_t batch, int num_ranges) { return range_ws_bytes(batch, num_ranges); }

int sgl_amd_top_k_top_p_min_p_sample_ranges(const float* probs, int64_t row_stride, int64_t batch, int64_t vocab,
                                            const int32_t* top_ks, const float* top_ps, const float* min_ps,
                                            const int64_t* seeds, const int64_t* positions, int32_t* out_ids,
                                            void* ws_keys, void* ws_toks, int32_t* out_n_keep, int num_ranges,
                                            void* ws_ranges, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(vocab > 0 && vocab <= 0x7fffffffLL, "top_k_top_p_min_p_sample_ranges: bad vocab");
  SGL_CHECK_ARG(batch <= 65535, "top_k_top_p_min_p_sample_ranges: batch <= 65535");
  SGL_CHECK_ARG(seeds != nullptr, "top_k_top_p_min_p_sample_ranges: seeds are required");
  SGL_CHECK_ARG((ws_keys == nullptr) == (ws_toks == nullptr), "top_k_top_p_min_p_sample_ranges: pass both ranking workspaces or neither");
  SGL_CHECK_ARG(num_ranges >= 2 && num_ranges <= 64 && ws_ranges, "top_k_top_p_min_p_sample_ranges: 2..64 column ranges and their workspace");
  SGL_CHECK_ARG((reinterpret_cast<uintptr_t>(ws_ranges) & 15) == 0, "top_k_top_p_min_p_sample_ranges: the range workspace must be 16-byte aligned");
  if (batch == 0) return 0;
  SampleParams p;
  p.probs = probs; p.row_stride = row_stride; p.V = static_cast<int>(vocab);
  p.top_ks = top_ks; p.top_ps = top_ps; p.min_ps = min_ps; p.seeds = seeds; p.positions = positions;
  p.out_ids = out_ids; p.ws_keys = static_cast<uint32_t*>(ws_keys); p.ws_toks = static_cast<int32_t*>(ws_toks);
  p.out_n_keep = out_n_keep; p.filtered = 1;
  const RangeWs ws = range_ws_view(ws_ranges, batch, num_ranges);
  const dim3 grid(num_ranges, static_cast<unsigned>(batch));
  hipLaunchKernelGGL(sample_hist_ranges_kernel, grid, dim3(kT), 0, as_stream(stream), p, num_ranges, ws);
  hipLaunchKernelGGL(sample_collect_ranges_kernel, grid, dim3(kT), 0, as_stream(stream), p, num_ranges, ws);
  hipLaunchKernelGGL(sample_finish_ranges_kernel, dim3(batch), dim3(kT), 0, as_stream(stream), p, ws);
  SGL_CHECK_LAUNCH("top_k_top_p_min_p_sample_ranges");        // (the empty-nucleus fix-up is part of the finish launch)
  return 0;
}

int64_t sgl_amd_sample_from_logits_workspace_bytes(int64_t batch, int num_splits) { return fast_ws_bytes(batch, num_splits); }

int sgl_amd_top_k_top_p_min_p_sample_from_logits(const void* logits, int logits_is_bf16, int64_t logits_row_stride, const float* temperatures,
                                                 float* probs_scratch, int64_t probs_row_stride, int64_t batch, int64_t vocab,
                                                 const int32_t* top_ks, const float* top_ps, const float* min_ps,
                                                 const int64_t* seeds, const int64_t* positions, int32_t* out_ids,
                                                 void* ws_keys, void* ws_toks, int32_t* out_n_keep, int num_splits,
                                                 void* ws_fast, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(vocab > 0 && vocab <= 0x7fffffffLL, "sample_from_logits: bad vocab");
  SGL_CHECK_ARG(batch <= 65535, "sample_from_logits: batch <= 65535");
  SGL_CHECK_ARG(seeds != nullptr && temperatures != nullptr, "sample_from_logits: seeds and temperatures are required");
  SGL_CHECK_ARG((ws_keys == nullptr) == (ws_toks == nullptr), "sample_from_logits: pass both ranking workspaces or neither");
  SGL_CHECK_ARG(num_splits >= 2 && num_splits <= 64 && ws_fast && (reinterpret_cast<uintptr_t>(ws_fast) & 15) == 0,
                "sample_from_logits: 2..64 softmax ranges and a 16-byte aligned workspace");
  SGL_CHECK_ARG(logits && (reinterpret_cast<uintptr_t>(logits) & (logits_is_bf16 ? 7 : 15)) == 0 && logits_row_stride % 4 == 0 && probs_scratch &&
                    (reinterpret_cast<uintptr_t>(probs_scratch) & 15) == 0 && probs_row_stride % 4 == 0,
                "sample_from_logits: needs 8-byte aligned bf16 (16-byte aligned fp32) rows and a 16-byte aligned fp32 scratch matrix for the rows "
                "redone the long way");
  if (batch == 0) return 0;
  SampleParams p;
  p.probs = probs_scratch; p.row_stride = probs_row_stride; p.V = static_cast<int>(vocab);
  p.top_ks = top_ks; p.top_ps = top_ps; p.min_ps = min_ps; p.seeds = seeds; p.positions = positions;
  p.out_ids = out_ids; p.ws_keys = static_cast<uint32_t*>(ws_keys); p.ws_toks = static_cast<int32_t*>(ws_toks);
  p.out_n_keep = out_n_keep; p.filtered = 1;
  const FastWs ws = fast_ws_view(ws_fast, batch, num_splits);
  const int G = (num_splits >= kFastRangesMax && num_splits % kFastRangesMax == 0) ? num_splits / kFastRangesMax : 1;
  const int R = num_splits / G;
  SGL_CHECK_ARG(R <= kFastRangesMax, "sample_from_logits: num_splits must be <= 16 or a multiple of 16");
  const int64_t per = ((vocab + num_splits - 1) / num_splits + 3) / 4 * 4;
  const int J = static_cast<int>((per + 1023) / 1024);
  SGL_CHECK_ARG(G <= 4 && G * J <= kFastSlots, "sample_from_logits: a workgroup holds at most 16384 columns of a row (vocab / ranges too large)");
  const dim3 cgrid(R, static_cast<unsigned>(batch));
#define SGL_LAUNCH_FAST(IN_)                                                                                                               \
  do {                                                                                                                                     \
    const IN_* x = static_cast<const IN_*>(logits);                                                                                        \
    if (G * J <= 8)                                                                                                                        \
      hipLaunchKernelGGL((sample_logit_candidates_kernel<IN_, 8>), cgrid, dim3(kSplitThreads), 0, as_stream(stream), x, logits_row_stride, \
                         temperatures, static_cast<int>(vocab), num_splits, G, J, top_ks, ws);                                             \
    else                                                                                                                                   \
      hipLaunchKernelGGL((sample_logit_candidates_kernel<IN_, kFastSlots>), cgrid, dim3(kSplitThreads), 0, as_stream(stream), x,           \
                         logits_row_stride, temperatures, static_cast<int>(vocab), num_splits, G, J, top_ks, ws);                          \
    hipLaunchKernelGGL(sample_finish_fast_kernel<IN_>, dim3(batch), dim3(kT), 0, as_stream(stream), p, x, logits_row_stride, temperatures,  \
                       num_splits, R, ws);                                                                                                 \
  } while (0)
  if (logits_is_bf16) SGL_LAUNCH_FAST(uint16_t);
  else SGL_LAUNCH_FAST(float);
#undef SGL_LAUNCH_FAST
  SGL_CHECK_LAUNCH("top_k_top_p_min_p_sample_from_logits");
  return 0;
}

int sgl_amd_top_k_top_p_renorm_probs(const float* probs, float* out, int64_t in_row_stride,
                                     int64_t out_row_stride, int64_t batch, int64_t vocab,
                                     const int32_t* top_ks, int top_k_val, const float* top_ps,
                                     float top_p_val, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(vocab > 0 && vocab <= 0x7fffffffLL, "renorm_probs: bad vocab");
  SGL_CHECK_ARG(batch <= 0x7fffffffLL, "renorm_probs: batch too large");
  if (batch == 0) return 0;
  hipLaunchKernelGGL(renorm_kernel, dim3(batch), dim3(kT), 0, as_stream(stream), probs, out, in_row_stride,
                     out_row_stride, static_cast<int>(vocab), top_ks, top_k_val, top_ps, top_p_val);
  SGL_CHECK_LAUNCH("renorm_probs");
  return 0;
}

}  // extern "C"

// HBM-bound row kernels of the decoder layer: RMSNorm / fused-add RMSNorm,
// SiLU-and-mul, rotary embedding, KV-row scatter into the token->KV pool.
//
// Replaces (reference, /root/reference/python/sglang):
//   srt/layers/layernorm.py:777-826            RMSNorm.forward_native
//   kernels/ops/layernorm/__init__.py:75-92    rmsnorm forward_native
//   srt/layers/activation.py:141-143           SiluAndMul.forward_native
//   srt/layers/rotary_embedding/base.py:236-276 + utils.py:36-63
//   srt/mem_cache/memory_pool.py:141-193       _set_kv_buffer_impl / store_cache
//
// Design (gfx950): every kernel moves 16 B per lane per access (8 bf16), one
// workgroup per row, wave64 shuffle reductions, fp32 math with the reference's
// bf16 rounding points reproduced so results are bit-comparable with the
// torch-native path.
#include "common.hpp"
#include "kv_format.hpp"
#include "sglang_amd.h"

using namespace sgl_amd;

namespace {

constexpr int kMaxVecPerThread = 8;  // 8 x (8 bf16) per thread -> hidden <= 64 * threads

__device__ __forceinline__ void unpack8(const U4& v, float* f) {
  f[0] = bf_lo(v.x); f[1] = bf_hi(v.x);
  f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
  f[4] = bf_lo(v.z); f[5] = bf_hi(v.z);
  f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ U4 pack8(const float* f) {
  U4 v;
  v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]);
  v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
  return v;
}

// ---------------------------------------------------------------------------
// RMSNorm.  FUSED: x += residual (fp32), residual <- bf16(x), norm on fp32 x.
// layernorm.py:786-820: variance = mean(x^2) over fp32 x; x*=rsqrt(var+eps);
// out = bf16((x * w))  [cast after the weight multiply].
// ---------------------------------------------------------------------------
template <bool FUSED>
__global__ void rmsnorm_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ residual,
                               const uint16_t* __restrict__ w, uint16_t* __restrict__ out,
                               int hidden, int64_t x_stride, int64_t res_stride,
                               int64_t out_stride, float eps) {
  __shared__ float scratch[16];
  const int64_t row = blockIdx.x;
  const int nvec = hidden >> 3;
  const uint16_t* xr = x + row * x_stride;
  uint16_t* rr = FUSED ? residual + row * res_stride : nullptr;
  uint16_t* orow = out + row * out_stride;

  float xs[kMaxVecPerThread][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxVecPerThread; ++i) {
    const int v = i * blockDim.x + threadIdx.x;
    if (v < nvec) {
      unpack8(ld16(xr + (v << 3)), xs[i]);
      if (FUSED) {
        float rs[8];
        unpack8(ld16(rr + (v << 3)), rs);
#pragma unroll
        for (int j = 0; j < 8; ++j) xs[i][j] += rs[j];
        st16(rr + (v << 3), pack8(xs[i]));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += xs[i][j] * xs[i][j];
    }
  }
  ss = block_sum(ss, scratch);
  const float var = ss / static_cast<float>(hidden);
  const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < kMaxVecPerThread; ++i) {
    const int v = i * blockDim.x + threadIdx.x;
    if (v < nvec) {
      float ws[8], o[8];
      unpack8(ld16(w + (v << 3)), ws);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (xs[i][j] * rs) * ws[j];
      st16(orow + (v << 3), pack8(o));
    }
  }
}

// hidden[row] = table[ids[row]] (the embedding lookup of the decode step, llama.py:433-437) and out[row] = RMSNorm(hidden[row])
// (the first layer's input_layernorm, llama.py:349-353) in ONE launch: the rows are in registers anyway.  ids outside
// [0, vocab) read row 0 (the reference's F.embedding would fault: the caller never passes them).
__global__ void embedding_rmsnorm_kernel(const int64_t* __restrict__ ids, const uint16_t* __restrict__ table,
                                         const uint16_t* __restrict__ w, uint16_t* __restrict__ hidden_out,
                                         uint16_t* __restrict__ out, int hidden, int64_t vocab, int64_t table_stride,
                                         int64_t hid_stride, int64_t out_stride, float eps) {
  __shared__ float scratch[16];
  const int64_t row = blockIdx.x;
  const int nvec = hidden >> 3;
  int64_t id = ids[row];
  if (id < 0 || id >= vocab) id = 0;
  const uint16_t* xr = table + id * table_stride;
  uint16_t* hr = hidden_out + row * hid_stride;
  uint16_t* orow = out + row * out_stride;
  float xs[kMaxVecPerThread][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxVecPerThread; ++i) {
    const int v = i * blockDim.x + threadIdx.x;
    if (v < nvec) {
      const U4 raw = ld16(xr + (v << 3));
      st16(hr + (v << 3), raw);
      unpack8(raw, xs[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += xs[i][j] * xs[i][j];
    }
  }
  ss = block_sum(ss, scratch);
  const float rs = 1.0f / sqrtf(ss / static_cast<float>(hidden) + eps);
#pragma unroll
  for (int i = 0; i < kMaxVecPerThread; ++i) {
    const int v = i * blockDim.x + threadIdx.x;
    if (v < nvec) {
      float ws[8], o[8];
      unpack8(ld16(w + (v << 3)), ws);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (xs[i][j] * rs) * ws[j];
      st16(orow + (v << 3), pack8(o));
    }
  }
}

// ---------------------------------------------------------------------------
// SiLU-and-mul.  in [T, 2d] -> out [T, d].  torch-native rounding
// (activation.py:141-143): s = bf16(silu(g)); out = bf16(s * u).
// With ROUND_MID=false the product is formed in fp32 (the sgl_kernel form).
// ---------------------------------------------------------------------------
constexpr int kSiluVecPerThread = 8;   // 16-byte gate + up loads in flight per thread: 16

// The projection's output is read exactly once, here: non-temporal loads keep it from displacing what the next GEMM
// wants in L2 (T = 4096: 67 -> 56 us, T = 7680: 127 -> 102 us = 6.3-6.5 TB/s; benchmarks/r03_exp2_elementwise.sh).
__device__ __forceinline__ U4 ld16_once(const uint16_t* p) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(U4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)));
}

template <bool ROUND_MID>
__global__ __launch_bounds__(256) void silu_and_mul_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                           int d, int64_t in_stride, int64_t out_stride) {
  const int64_t row = blockIdx.y;
  const uint16_t* g = in + row * in_stride;
  const uint16_t* u = g + d;
  uint16_t* o = out + row * out_stride;
  const int nvec = d >> 3;
  // a workgroup owns kSiluVecPerThread consecutive 256-vector slabs of the row: all loads go out before the first
  // exp, so a thread keeps 256 B of reads in flight instead of 32
  const int v0 = blockIdx.x * (256 * kSiluVecPerThread) + threadIdx.x;
  U4 gv[kSiluVecPerThread], uv[kSiluVecPerThread];
#pragma unroll
  for (int i = 0; i < kSiluVecPerThread; ++i) {
    const int v = v0 + i * 256;
    if (v < nvec) {
      gv[i] = ld16_once(g + (static_cast<int64_t>(v) << 3));
      uv[i] = ld16_once(u + (static_cast<int64_t>(v) << 3));
    }
  }
#pragma unroll
  for (int i = 0; i < kSiluVecPerThread; ++i) {
    const int v = v0 + i * 256;
    if (v >= nvec) break;
    float gs[8], us[8], r[8];
    unpack8(gv[i], gs);
    unpack8(uv[i], us);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // hardware exp2 / rcp (1 ulp each): the result is rounded to bf16 right after, 2^-15 of the outputs can land on
      // the other side of a rounding boundary; the IEEE expf + division made this kernel VALU-bound at 4.5 TB/s
      float sl = gs[j] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gs[j] * -1.4426950408889634f));
      if (ROUND_MID) sl = rbf(sl);
      r[j] = sl * us[j];
    }
    st16(o + (static_cast<int64_t>(v) << 3), pack8(r));
  }
}

// ---------------------------------------------------------------------------
// Rotary embedding, in place on q [T, Hq, D] and k [T, Hk, D].
// cos_sin_cache [max_pos, rot] = cos || sin halves (base.py:173-182), bf16 or
// fp32.  torch-native rounding (utils.py:49-57): cos/sin cast to bf16, then
//   o1 = bf16(bf16(x1*cos) - bf16(x2*sin)),  o2 = bf16(bf16(x2*cos) + bf16(x1*sin)).
// One workgroup per token; thread t handles (head, pair-chunk).
// Optionally scatters the rotated K row and the V row to the KV pool
// (the fused rope+store of base.py:385-417).
// ---------------------------------------------------------------------------
template <bool NEOX, bool CACHE_F32>
__global__ void rope_kernel(const int64_t* __restrict__ positions, uint16_t* __restrict__ q,
                            uint16_t* __restrict__ k, const void* __restrict__ cos_sin_cache,
                            int num_q_heads, int num_k_heads, int head_dim, int rot_dim,
                            int64_t q_stride, int64_t k_stride,
                            // optional fused KV store
                            const uint16_t* __restrict__ v, int64_t v_stride,
                            uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
                            const int64_t* __restrict__ loc, int64_t cache_row_stride) {
  const int64_t tok = blockIdx.x;
  const int64_t pos = positions[tok];
  const int half = rot_dim >> 1;
  const int total_heads = num_q_heads + num_k_heads;
  const int work = total_heads * half;
  // neox halves that are whole 16-byte vectors (every model of the bench): 8 pairs per thread with 16-byte loads and
  // stores -- the per-pair form below moves 2 bytes per lane and ran a prefill-sized call at 2.4 TB/s
  const bool vec = NEOX && (half & 7) == 0 && (head_dim & 7) == 0 && (q_stride & 7) == 0 && (k_stride & 7) == 0 &&
                   (reinterpret_cast<uintptr_t>(q) & 15) == 0 && (reinterpret_cast<uintptr_t>(k) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(cos_sin_cache) & 15) == 0 && (rot_dim & 7) == 0;
  if (vec) {
    const int hv = half >> 3;                      // 8-pair items per head
    for (int idx = threadIdx.x; idx < total_heads * hv; idx += blockDim.x) {
      const int h = idx / hv;
      const int i = (idx - h * hv) << 3;
      float c[8], sn[8];
      if (CACHE_F32) {
        const float* cs = reinterpret_cast<const float*>(cos_sin_cache) + pos * rot_dim;
#pragma unroll
        for (int j = 0; j < 8; ++j) { c[j] = rbf(cs[i + j]); sn[j] = rbf(cs[half + i + j]); }
      } else {
        const uint16_t* cs = reinterpret_cast<const uint16_t*>(cos_sin_cache) + pos * rot_dim;
        unpack8(ld16(cs + i), c);
        unpack8(ld16(cs + half + i), sn);
      }
      uint16_t* base = (h < num_q_heads) ? q + tok * q_stride + static_cast<int64_t>(h) * head_dim
                                         : k + tok * k_stride + static_cast<int64_t>(h - num_q_heads) * head_dim;
      float x1[8], x2[8], o1[8], o2[8];
      unpack8(ld16(base + i), x1);
      unpack8(ld16(base + half + i), x2);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o1[j] = rbf(x1[j] * c[j]) - rbf(x2[j] * sn[j]);
        o2[j] = rbf(x2[j] * c[j]) + rbf(x1[j] * sn[j]);
      }
      st16(base + i, pack8(o1));
      st16(base + half + i, pack8(o2));
    }
  } else
  for (int idx = threadIdx.x; idx < work; idx += blockDim.x) {
    const int h = idx / half;
    const int i = idx - h * half;
    float c, s;
    if (CACHE_F32) {
      const float* cs = reinterpret_cast<const float*>(cos_sin_cache) + pos * rot_dim;
      c = rbf(cs[i]);
      s = rbf(cs[half + i]);
    } else {
      const uint16_t* cs = reinterpret_cast<const uint16_t*>(cos_sin_cache) + pos * rot_dim;
      c = bf2f(cs[i]);
      s = bf2f(cs[half + i]);
    }
    uint16_t* base = (h < num_q_heads) ? q + tok * q_stride + static_cast<int64_t>(h) * head_dim
                                       : k + tok * k_stride +
                                             static_cast<int64_t>(h - num_q_heads) * head_dim;
    const int i1 = NEOX ? i : 2 * i;
    const int i2 = NEOX ? i + half : 2 * i + 1;
    const float x1 = bf2f(base[i1]);
    const float x2 = bf2f(base[i2]);
    const float o1 = rbf(x1 * c) - rbf(x2 * s);
    const float o2 = rbf(x2 * c) + rbf(x1 * s);
    base[i1] = f2bf(o1);
    base[i2] = f2bf(o2);
  }
  if (k_cache != nullptr) {
    __syncthreads();  // this block's rotated K row is complete (block-local writes)
    const int64_t slot = loc[tok];
    const int row = num_k_heads * head_dim;
    const uint16_t* ks = k + tok * k_stride;
    const uint16_t* vs = v + tok * v_stride;
    uint16_t* kd = k_cache + slot * cache_row_stride;
    uint16_t* vd = v_cache + slot * cache_row_stride;
    for (int e = threadIdx.x * 8; e < row; e += blockDim.x * 8) {
      st16(kd + e, ld16(ks + e));
      st16(vd + e, ld16(vs + e));
    }
  }
}

// ---------------------------------------------------------------------------
// KV-row scatter: k_cache[loc[t], :] = k[t, :], same for v.  16 B per lane.
// ---------------------------------------------------------------------------
__global__ void store_kv_kernel(const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
                                const int64_t* __restrict__ loc, int row_elems, int v_row_elems,
                                int64_t k_stride, int64_t v_stride, int64_t kc_stride,
                                int64_t vc_stride) {
  const int64_t tok = blockIdx.x;
  const int64_t slot = loc[tok];
  const uint16_t* ks = k + tok * k_stride;
  const uint16_t* vs = v + tok * v_stride;
  uint16_t* kd = k_cache + slot * kc_stride;
  uint16_t* vd = v_cache + slot * vc_stride;
  for (int e = threadIdx.x * 8; e < row_elems; e += blockDim.x * 8) st16(kd + e, ld16(ks + e));
  for (int e = threadIdx.x * 8; e < v_row_elems; e += blockDim.x * 8) st16(vd + e, ld16(vs + e));
}

// The same scatter for any pool layout / element format: one workgroup per token, a thread per 8 elements of the
// token's [H_kv, D] row; fp8 rows hold x / scale rounded to OCP e4m3 (memory_pool.py:2364-2374).
template <bool FP8>
__global__ void store_kv_fmt_kernel(const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                    void* __restrict__ k_cache, void* __restrict__ v_cache,
                                    const int64_t* __restrict__ loc, int num_kv_heads, int head_dim,
                                    int64_t k_stride, int64_t v_stride, KvFormat fmt, float inv_k_scale, float inv_v_scale) {
  const int64_t tok = blockIdx.x;
  const int slot = static_cast<int>(loc[tok]);
  const int per_head = head_dim >> 3;
  for (int e = threadIdx.x; e < num_kv_heads * per_head; e += blockDim.x) {
    const int h = e / per_head, c = e - h * per_head;
    const U4 kv8 = ld16(k + tok * k_stride + static_cast<int64_t>(h) * head_dim + c * 8);
    const U4 vv8 = ld16(v + tok * v_stride + static_cast<int64_t>(h) * head_dim + c * 8);
    unsigned char* kd = const_cast<unsigned char*>(kv_row(k_cache, fmt, slot, h));
    unsigned char* vd = const_cast<unsigned char*>(kv_row(v_cache, fmt, slot, h));
    if constexpr (FP8) {
      const uint32_t kw[4] = {kv8.x, kv8.y, kv8.z, kv8.w}, vw[4] = {vv8.x, vv8.y, vv8.z, vv8.w};
      float kf[8], vf[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // (K / k_scale) is a bf16 division in the reference (cache_k.div_(k_scale) on a bf16 tensor) before the cast
        kf[2 * j] = rbf(bf_lo(kw[j]) * inv_k_scale); kf[2 * j + 1] = rbf(bf_hi(kw[j]) * inv_k_scale);
        vf[2 * j] = rbf(bf_lo(vw[j]) * inv_v_scale); vf[2 * j + 1] = rbf(bf_hi(vw[j]) * inv_v_scale);
      }
      *reinterpret_cast<uint2*>(kd + c * 8) = f32x8_to_fp8x8(kf, 1.0f);
      *reinterpret_cast<uint2*>(vd + c * 8) = f32x8_to_fp8x8(vf, 1.0f);
    } else {
      st16(kd + c * 16, kv8);
      st16(vd + c * 16, vv8);
    }
  }
}

inline int norm_threads(int hidden) {
  const int nvec = hidden >> 3;
  int t = (nvec + 1) / 2;
  t = ((t + 63) / 64) * 64;
  if (t < 64) t = 64;
  if (t > 1024) t = 1024;
  return t;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" {

int sgl_amd_rmsnorm(const void* x, const void* weight, void* out, int64_t num_rows, int hidden,
                    int64_t x_row_stride, int64_t out_row_stride, float eps, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(hidden > 0 && hidden % 8 == 0, "rmsnorm: hidden=%d must be a positive multiple of 8", hidden);
  SGL_CHECK_ARG(x_row_stride % 8 == 0 && out_row_stride % 8 == 0, "rmsnorm: row strides must be multiples of 8 elements");
  SGL_CHECK_ARG(aligned16(x) && aligned16(weight) && aligned16(out), "rmsnorm: pointers must be 16-byte aligned");
  if (num_rows == 0) return 0;
  const int threads = norm_threads(hidden);
  SGL_CHECK_ARG((hidden >> 3) <= threads * kMaxVecPerThread, "rmsnorm: hidden=%d too large", hidden);
  hipLaunchKernelGGL(rmsnorm_kernel<false>, dim3(num_rows), dim3(threads), 0, as_stream(stream),
                     static_cast<const uint16_t*>(x), nullptr, static_cast<const uint16_t*>(weight),
                     static_cast<uint16_t*>(out), hidden, x_row_stride, 0, out_row_stride, eps);
  SGL_CHECK_LAUNCH("rmsnorm");
  return 0;
}

int sgl_amd_embedding_rmsnorm(const int64_t* ids, const void* table, const void* weight, void* hidden_out, void* out,
                              int64_t num_rows, int hidden, int64_t vocab, int64_t table_row_stride, int64_t hidden_row_stride,
                              int64_t out_row_stride, float eps, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(hidden > 0 && hidden % 8 == 0 && vocab > 0, "embedding_rmsnorm: hidden=%d must be a positive multiple of 8", hidden);
  SGL_CHECK_ARG(table_row_stride % 8 == 0 && hidden_row_stride % 8 == 0 && out_row_stride % 8 == 0,
                "embedding_rmsnorm: row strides must be multiples of 8 elements");
  SGL_CHECK_ARG(ids && aligned16(table) && aligned16(weight) && aligned16(hidden_out) && aligned16(out),
                "embedding_rmsnorm: pointers must be 16-byte aligned");
  if (num_rows == 0) return 0;
  const int threads = norm_threads(hidden);
  SGL_CHECK_ARG((hidden >> 3) <= threads * kMaxVecPerThread, "embedding_rmsnorm: hidden=%d too large", hidden);
  hipLaunchKernelGGL(embedding_rmsnorm_kernel, dim3(num_rows), dim3(threads), 0, as_stream(stream), ids,
                     static_cast<const uint16_t*>(table), static_cast<const uint16_t*>(weight), static_cast<uint16_t*>(hidden_out),
                     static_cast<uint16_t*>(out), hidden, vocab, table_row_stride, hidden_row_stride, out_row_stride, eps);
  SGL_CHECK_LAUNCH("embedding_rmsnorm");
  return 0;
}

int sgl_amd_fused_add_rmsnorm(void* x, void* residual, const void* weight, int64_t num_rows,
                              int hidden, int64_t x_row_stride, int64_t res_row_stride, float eps,
                              void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(hidden > 0 && hidden % 8 == 0, "fused_add_rmsnorm: hidden=%d must be a positive multiple of 8", hidden);
  SGL_CHECK_ARG(x_row_stride % 8 == 0 && res_row_stride % 8 == 0, "fused_add_rmsnorm: row strides must be multiples of 8 elements");
  SGL_CHECK_ARG(aligned16(x) && aligned16(weight) && aligned16(residual), "fused_add_rmsnorm: pointers must be 16-byte aligned");
  if (num_rows == 0) return 0;
  const int threads = norm_threads(hidden);
  SGL_CHECK_ARG((hidden >> 3) <= threads * kMaxVecPerThread, "fused_add_rmsnorm: hidden=%d too large", hidden);
  hipLaunchKernelGGL(rmsnorm_kernel<true>, dim3(num_rows), dim3(threads), 0, as_stream(stream),
                     static_cast<const uint16_t*>(x), static_cast<uint16_t*>(residual),
                     static_cast<const uint16_t*>(weight), static_cast<uint16_t*>(x), hidden,
                     x_row_stride, res_row_stride, x_row_stride, eps);
  SGL_CHECK_LAUNCH("fused_add_rmsnorm");
  return 0;
}

int sgl_amd_silu_and_mul(const void* in, void* out, int64_t num_rows, int d, int64_t in_row_stride,
                         int64_t out_row_stride, int round_intermediate, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(d > 0 && d % 8 == 0, "silu_and_mul: d=%d must be a positive multiple of 8", d);
  SGL_CHECK_ARG(in_row_stride % 8 == 0 && out_row_stride % 8 == 0, "silu_and_mul: row strides must be multiples of 8 elements");
  SGL_CHECK_ARG(aligned16(in) && aligned16(out), "silu_and_mul: pointers must be 16-byte aligned");
  SGL_CHECK_ARG(num_rows <= 65535LL * 65535LL, "silu_and_mul: too many rows");
  if (num_rows == 0) return 0;
  const int nvec = d >> 3;
  const int threads = 256;
  const int bx = (nvec + threads * kSiluVecPerThread - 1) / (threads * kSiluVecPerThread);
  // grid.y is limited to 65535: fold the rows in chunks.
  for (int64_t r0 = 0; r0 < num_rows; r0 += 65535) {
    const int64_t nr = (num_rows - r0 < 65535) ? (num_rows - r0) : 65535;
    const uint16_t* ip = static_cast<const uint16_t*>(in) + r0 * in_row_stride;
    uint16_t* op = static_cast<uint16_t*>(out) + r0 * out_row_stride;
    if (round_intermediate)
      hipLaunchKernelGGL(silu_and_mul_kernel<true>, dim3(bx, nr), dim3(threads), 0, as_stream(stream),
                         ip, op, d, in_row_stride, out_row_stride);
    else
      hipLaunchKernelGGL(silu_and_mul_kernel<false>, dim3(bx, nr), dim3(threads), 0, as_stream(stream),
                         ip, op, d, in_row_stride, out_row_stride);
  }
  SGL_CHECK_LAUNCH("silu_and_mul");
  return 0;
}

int sgl_amd_rotary_embedding(const int64_t* positions, void* q, void* k, const void* cos_sin_cache,
                             int cache_is_f32, int64_t num_tokens, int num_q_heads, int num_k_heads,
                             int head_dim, int rot_dim, int64_t q_token_stride,
                             int64_t k_token_stride, int is_neox, const void* v,
                             int64_t v_token_stride, void* k_cache, void* v_cache,
                             const int64_t* cache_loc, int64_t cache_row_stride, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_dim, "rotary_embedding: bad rot_dim=%d head_dim=%d", rot_dim, head_dim);
  const bool fused = k_cache != nullptr;
  if (fused) {
    SGL_CHECK_ARG(v != nullptr && v_cache != nullptr && cache_loc != nullptr, "rotary_embedding: fused store needs v, v_cache, cache_loc");
    SGL_CHECK_ARG((num_k_heads * head_dim) % 8 == 0 && k_token_stride % 8 == 0 && v_token_stride % 8 == 0 && cache_row_stride % 8 == 0,
                  "rotary_embedding: fused store needs 16-byte aligned rows");
    SGL_CHECK_ARG(aligned16(k) && aligned16(v) && aligned16(k_cache) && aligned16(v_cache), "rotary_embedding: fused store needs 16-byte aligned pointers");
  }
  if (num_tokens == 0) return 0;
  const int work = (num_q_heads + num_k_heads) * (rot_dim / 2);
  int threads = ((work + 63) / 64) * 64;
  if (threads > 512) threads = 512;
  if (threads < 64) threads = 64;
#define SGL_ROPE_LAUNCH(NEOX, F32)                                                               \
  hipLaunchKernelGGL((rope_kernel<NEOX, F32>), dim3(num_tokens), dim3(threads), 0,               \
                     as_stream(stream), positions, static_cast<uint16_t*>(q),                    \
                     static_cast<uint16_t*>(k), cos_sin_cache, num_q_heads, num_k_heads,         \
                     head_dim, rot_dim, q_token_stride, k_token_stride,                          \
                     static_cast<const uint16_t*>(v), v_token_stride,                            \
                     static_cast<uint16_t*>(k_cache), static_cast<uint16_t*>(v_cache), cache_loc, \
                     cache_row_stride)
  if (is_neox) {
    if (cache_is_f32) SGL_ROPE_LAUNCH(true, true); else SGL_ROPE_LAUNCH(true, false);
  } else {
    if (cache_is_f32) SGL_ROPE_LAUNCH(false, true); else SGL_ROPE_LAUNCH(false, false);
  }
#undef SGL_ROPE_LAUNCH
  SGL_CHECK_LAUNCH("rotary_embedding");
  return 0;
}

int sgl_amd_store_kv_cache(const void* k, const void* v, void* k_cache, void* v_cache,
                           const int64_t* loc, int64_t num_tokens, int k_row_elems,
                           int v_row_elems, int64_t k_token_stride, int64_t v_token_stride,
                           int64_t k_cache_row_stride, int64_t v_cache_row_stride, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(k_row_elems % 8 == 0 && v_row_elems % 8 == 0, "store_kv_cache: row sizes must be multiples of 8 elements (16 B)");
  SGL_CHECK_ARG(k_token_stride % 8 == 0 && v_token_stride % 8 == 0 && k_cache_row_stride % 8 == 0 && v_cache_row_stride % 8 == 0,
                "store_kv_cache: strides must be multiples of 8 elements");
  SGL_CHECK_ARG(aligned16(k) && aligned16(v) && aligned16(k_cache) && aligned16(v_cache), "store_kv_cache: pointers must be 16-byte aligned");
  if (num_tokens == 0) return 0;
  const int maxrow = k_row_elems > v_row_elems ? k_row_elems : v_row_elems;
  int threads = ((maxrow / 8 + 63) / 64) * 64;
  if (threads > 256) threads = 256;
  if (threads < 64) threads = 64;
  hipLaunchKernelGGL(store_kv_kernel, dim3(num_tokens), dim3(threads), 0, as_stream(stream),
                     static_cast<const uint16_t*>(k), static_cast<const uint16_t*>(v),
                     static_cast<uint16_t*>(k_cache), static_cast<uint16_t*>(v_cache), loc,
                     k_row_elems, v_row_elems, k_token_stride, v_token_stride, k_cache_row_stride,
                     v_cache_row_stride);
  SGL_CHECK_LAUNCH("store_kv_cache");
  return 0;
}

int sgl_amd_store_kv_cache_ex(const void* k, const void* v, void* k_cache, void* v_cache, const int64_t* loc,
                              int64_t num_tokens, int num_kv_heads, int head_dim, int64_t k_token_stride,
                              int64_t v_token_stride, int64_t cache_row_stride, int kv_fp8, float k_scale, float v_scale,
                              int page_size, int kv_layout_hnd, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(head_dim % 8 == 0 && num_kv_heads > 0, "store_kv_cache: head_dim must be a multiple of 8");
  SGL_CHECK_ARG(k_token_stride % 8 == 0 && v_token_stride % 8 == 0 && cache_row_stride % 8 == 0,
                "store_kv_cache: strides must be multiples of 8 elements");
  SGL_CHECK_ARG(!kv_fp8 || (k_scale > 0.f && v_scale > 0.f), "store_kv_cache: fp8 KV needs positive k_scale / v_scale");
  if (num_tokens == 0) return 0;
  KvFormat fmt;
  SGL_CHECK_ARG(make_kv_format(&fmt, cache_row_stride, num_kv_heads, head_dim, page_size, kv_layout_hnd, kv_fp8),
                "store_kv_cache: HND pools need a power-of-two page_size (got %d)", page_size);
  int threads = ((num_kv_heads * head_dim / 8 + 63) / 64) * 64;
  if (threads > 256) threads = 256;
  if (kv_fp8)
    hipLaunchKernelGGL(store_kv_fmt_kernel<true>, dim3(num_tokens), dim3(threads), 0, as_stream(stream),
                       static_cast<const uint16_t*>(k), static_cast<const uint16_t*>(v), k_cache, v_cache, loc, num_kv_heads,
                       head_dim, k_token_stride, v_token_stride, fmt, 1.0f / k_scale, 1.0f / v_scale);
  else
    hipLaunchKernelGGL(store_kv_fmt_kernel<false>, dim3(num_tokens), dim3(threads), 0, as_stream(stream),
                       static_cast<const uint16_t*>(k), static_cast<const uint16_t*>(v), k_cache, v_cache, loc, num_kv_heads,
                       head_dim, k_token_stride, v_token_stride, fmt, 1.0f, 1.0f);
  SGL_CHECK_LAUNCH("store_kv_cache_ex");
  return 0;
}

}  // extern "C"

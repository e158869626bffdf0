// Paged decode attention for gfx950: one query token per request over the
// request's cached KV rows in the token->KV pool (page_size = 1 row gather).
//
// Replaces (reference, /root/reference/python/sglang):
//   kernels/ops/attention/decode_attention.py:540 _fwd_grouped_kernel_stage1,
//   :911 _fwd_kernel_stage2, :1163 decode_attention_fwd            (Triton)
// Oracle: srt/layers/attention/torch_native_backend.py:176-277.
//
// This kernel is HBM-bound (len * 2*H_kv*D*2 bytes per
// request and layer), so it is built around the memory system, not MFMA:
//   * a KV row (one token, one kv head, D bf16) is D*2 contiguous bytes; D/8
//     adjacent lanes read it with one 16-byte load each, so every wave load
//     instruction fetches 64/(D/8) complete rows, fully coalesced;
//   * each wave keeps 4 K + 4 V such loads per lane in flight and software-
//     pipelines the next tile's loads (and the slot indices two tiles ahead)
//     under the current tile's math;
//   * all G = H_q/H_kv query heads of a kv head are processed from the same
//     registers (GQA reuse), scores reduced across the D/8 lanes of a row with
//     DPP row operations, online softmax state kept per 16-lane group and
//     merged once at the end (wave shuffles, then LDS across the 4 waves);
//   * long sequences are split over gridDim.z; partial (m, l, acc) go to a
//     caller-owned fp32 workspace and a small second kernel merges them.
#include "common.hpp"
#include "kv_format.hpp"
#include "sglang_amd.h"

using namespace sgl_amd;

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kLoadsPerTile = 4;   // K (and V) 16-byte loads per lane per tile
constexpr float kNegBig = -1.0e30f;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// Sum over the LPR adjacent lanes that hold one KV row.
template <int LPR, bool USE_DPP>
__device__ __forceinline__ float row_sum(float v) {
  if (USE_DPP && LPR <= 16) {
    v += dpp_mov<0xB1>(v);                  // quad_perm [1,0,3,2]  (xor 1)
    v += dpp_mov<0x4E>(v);                  // quad_perm [2,3,0,1]  (xor 2)
    if (LPR >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror      (other quad of 8)
    if (LPR >= 16) v += dpp_mov<0x140>(v);  // row_mirror           (other half of 16)
    return v;
  }
#pragma unroll
  for (int off = 1; off < LPR; off <<= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ void cvt8(const U4& v, float* f) {
  f[0] = bf_lo(v.x); f[1] = bf_hi(v.x);
  f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
  f[4] = bf_lo(v.z); f[5] = bf_hi(v.z);
  f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}

struct DecodeParams {
  const uint16_t* q;          // [B, Hq, D]
  const uint16_t* k_cache;    // [slots, Hkv, D]
  const uint16_t* v_cache;
  const int32_t* req_to_token;  // [reqs, max_ctx]   (or flat kv_indices when kv_indptr != null)
  const int64_t* req_pool_indices;  // [B]
  const int32_t* seq_lens;    // [B]   (kv length per request)
  const int32_t* kv_indptr;   // optional [B+1]: token t of request b is req_to_token[kv_indptr[b]+t]
  uint16_t* out;              // [B, Hq, D]
  float* ws_acc;              // [B, Hq, S, D] unnormalised partial outputs
  float* ws_ml;               // [B, Hq, S, 2] (running max in log2 units, sum)
  int64_t q_stride;           // elements between tokens of q
  int64_t out_stride;
  int64_t kc_stride;          // elements between slots
  int64_t vc_stride;
  int64_t r2t_stride;
  int num_q_heads;
  int num_kv_heads;
  int num_splits;
  int min_chunk;              // split chunk granularity (tokens)
  float scale_log2;           // sm_scale * log2(e)
  KvFormat fmt;               // layout + element format of the pool rows
  float v_scale;              // fp8 rows: multiplied into the output (k_scale is folded into scale_log2)
  int window;                 // sliding window: kv positions >= len - 1 - window (torch_native_backend.py:36-48); < 0: off
  float cap_log2, inv_cap_log2;   // logit soft cap (cap * log2 e, 1 / (cap * log2 e)); cap_log2 == 0: off
  // optional [B] permutation that puts requests sharing a KV prefix next to each other: with it the
  // (request, kv head) workgroups are laid out so that a group's members run on ONE XCD (workgroup
  // i runs on XCD i % 8) and re-read the shared rows from that XCD's L2 instead of HBM
  const int32_t* batch_order;
};

__device__ __forceinline__ void split_range(int len, int num_splits, int min_chunk, int split,
                                            int& c0, int& c1) {
  int chunk = (len + num_splits - 1) / num_splits;
  chunk = ((chunk + min_chunk - 1) / min_chunk) * min_chunk;
  c0 = split * chunk;
  c1 = c0 + chunk;
  if (c1 > len) c1 = len;
  if (c0 > len) c0 = len;
}

// G = query heads per kv head handled by this block (padded to a power of two),
// D = head dim.  Grid: (B, Hkv * head_blocks, splits).
template <int G, int D, bool USE_DPP, bool FP8>
__global__ __launch_bounds__(kThreads) void decode_stage1_kernel(DecodeParams p, int group_size,
                                                                 int head_blocks) {
  constexpr int LPR = D / 8;          // lanes per KV row
  constexpr int TPI = 64 / LPR;       // tokens (rows) per wave load instruction
  constexpr int TILE = TPI * kLoadsPerTile;  // tokens per wave iteration

  __shared__ float sm_o[kWaves][G][D];
  __shared__ float sm_m[kWaves][G];
  __shared__ float sm_l[kWaves][G];

  int b = blockIdx.x;
  int by = blockIdx.y;
  if (p.batch_order != nullptr) {
    // linear work index L over (kv-head block, sorted request); XCD x owns the contiguous range
    // [x * per, (x + 1) * per) of it
    const int nb = gridDim.x, total = gridDim.x * gridDim.y;
    const int i = blockIdx.y * nb + blockIdx.x;          // dispatch order: x fastest
    const int qd = total / 8, rm = total % 8, xcd = i % 8;   // bijective for any total
    const int L = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + i / 8;
    by = L / nb;
    b = p.batch_order[L - by * nb];
  }
  const int kvh = by / head_blocks;
  const int hb = by - kvh * head_blocks;
  const int split = blockIdx.z;
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int sub = lane / LPR;         // which row of the load instruction
  const int col = lane - sub * LPR;   // 16-byte column inside the row

  const int len = p.seq_lens[b];
  // sliding window: the query (position len - 1) sees kv positions [len - 1 - window, len - 1]
  const int kv_start = (p.window >= 0 && len - 1 - p.window > 0) ? len - 1 - p.window : 0;
  int c0, c1;
  split_range(len - kv_start, p.num_splits, p.min_chunk, split, c0, c1);
  c0 += kv_start; c1 += kv_start;
  const int h0 = kvh * group_size + hb * G;                 // first q head of this block
  int g_valid = group_size - hb * G;
  if (g_valid > G) g_valid = G;

  const bool direct = p.num_splits == 1;
  if (c0 >= c1) {
    // Empty split: stage 2 skips it (same split_range there).  A zero-length
    // request writes zeros so the output is defined.
    if (len == 0 && split == 0) {
      for (int idx = threadIdx.x; idx < g_valid * D; idx += kThreads) {
        const int h = idx / D, d = idx - h * D;
        p.out[static_cast<int64_t>(b) * p.out_stride + static_cast<int64_t>(h0 + h) * D + d] = 0;
      }
    }
    return;
  }

  const int32_t* idx_base =
      p.kv_indptr ? p.req_to_token + p.kv_indptr[b]
                  : p.req_to_token + p.req_pool_indices[b] * p.r2t_stride;

  // ---- query heads -> registers (fp32, pre-multiplied by scale*log2e) -----
  float qf[G][8];
#pragma unroll
  for (int h = 0; h < G; ++h) {
    if (h < g_valid) {
      const uint16_t* qp = p.q + static_cast<int64_t>(b) * p.q_stride +
                           static_cast<int64_t>(h0 + h) * D + col * 8;
      cvt8(ld16(qp), qf[h]);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[h][j] *= p.scale_log2;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[h][j] = 0.f;
    }
  }

  float m_run[G], l_run[G], acc[G][8];
#pragma unroll
  for (int h = 0; h < G; ++h) {
    m_run[h] = kNegBig;
    l_run[h] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[h][j] = 0.f;
  }

  const int last_tok = c1 - 1;
  const int tile_stride = kWaves * TILE;
  int t_cur = c0 + wid * TILE;  // first token of this wave's current tile

  // slot indices of a tile (clamped to a valid token so the loads stay in bounds)
  auto load_slots = [&](int tbase, int32_t* s) {
#pragma unroll
    for (int u = 0; u < kLoadsPerTile; ++u) {
      int t = tbase + u * TPI + sub;
      if (t > last_tok) t = last_tok;
      s[u] = idx_base[t];
    }
  };
  auto load_rows = [&](const int32_t* s, U4* kr, U4* vr) {
#pragma unroll
    for (int u = 0; u < kLoadsPerTile; ++u) {
      kr[u] = ld_kv8<FP8>(kv_row(p.k_cache, p.fmt, s[u], kvh), col);
      vr[u] = ld_kv8<FP8>(kv_row(p.v_cache, p.fmt, s[u], kvh), col);
    }
  };

  int32_t slot_a[kLoadsPerTile], slot_b[kLoadsPerTile];
  U4 k_cur[kLoadsPerTile], v_cur[kLoadsPerTile], k_nxt[kLoadsPerTile], v_nxt[kLoadsPerTile];

  if (t_cur < c1) {
    load_slots(t_cur, slot_a);
    load_rows(slot_a, k_cur, v_cur);
    if (t_cur + tile_stride < c1) load_slots(t_cur + tile_stride, slot_a);
  }

  while (t_cur < c1) {
    const int t_next = t_cur + tile_stride;
    const bool has_next = t_next < c1;
    if (has_next) {
      load_rows(slot_a, k_nxt, v_nxt);                       // tile i+1 rows
      if (t_next + tile_stride < c1) load_slots(t_next + tile_stride, slot_b);  // tile i+2 slots
    }

    // ---- scores for the kLoadsPerTile rows this lane group owns ----------
    float s[G][kLoadsPerTile];
#pragma unroll
    for (int u = 0; u < kLoadsPerTile; ++u) {
      float kf[8];
      cvt8(k_cur[u], kf);
      const bool valid = (t_cur + u * TPI + sub) <= last_tok;
#pragma unroll
      for (int h = 0; h < G; ++h) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) d = fmaf(qf[h][j], kf[j], d);
        d = row_sum<LPR, USE_DPP>(d);
        if (p.cap_log2 != 0.f) d = soft_cap_log2(d, p.cap_log2, p.inv_cap_log2);
        s[h][u] = valid ? d : kNegBig;
      }
    }
    // ---- online softmax + PV over the same rows --------------------------
    float pw[G][kLoadsPerTile];
#pragma unroll
    for (int h = 0; h < G; ++h) {
      float mx = s[h][0];
#pragma unroll
      for (int u = 1; u < kLoadsPerTile; ++u) mx = fmaxf(mx, s[h][u]);
      const float m_new = fmaxf(m_run[h], mx);
      const float alpha = exp2f(m_run[h] - m_new);
      float psum = 0.f;
#pragma unroll
      for (int u = 0; u < kLoadsPerTile; ++u) {
        const float e = (s[h][u] > 0.5f * kNegBig) ? exp2f(s[h][u] - m_new) : 0.f;
        pw[h][u] = e;
        psum += e;
      }
      l_run[h] = l_run[h] * alpha + psum;
      m_run[h] = m_new;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[h][j] *= alpha;
    }
#pragma unroll
    for (int u = 0; u < kLoadsPerTile; ++u) {
      float vf[8];
      cvt8(v_cur[u], vf);
#pragma unroll
      for (int h = 0; h < G; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[h][j] = fmaf(pw[h][u], vf[j], acc[h][j]);
    }

    if (has_next) {
#pragma unroll
      for (int u = 0; u < kLoadsPerTile; ++u) {
        k_cur[u] = k_nxt[u];
        v_cur[u] = v_nxt[u];
        slot_a[u] = slot_b[u];
      }
    }
    t_cur = t_next;
  }

  // ---- merge the TPI lane groups of the wave (lanes with equal `col`) -----
#pragma unroll
  for (int h = 0; h < G; ++h) {
    float m_all = m_run[h];
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) m_all = fmaxf(m_all, __shfl_xor(m_all, off, 64));
    const float sc = exp2f(m_run[h] - m_all);
    float l = l_run[h] * sc;
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) l += __shfl_xor(l, off, 64);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = acc[h][j] * sc;
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) a += __shfl_xor(a, off, 64);
      acc[h][j] = a;
    }
    m_run[h] = m_all;
    l_run[h] = l;
  }
  if (sub == 0) {
#pragma unroll
    for (int h = 0; h < G; ++h) {
#pragma unroll
      for (int j = 0; j < 8; ++j) sm_o[wid][h][col * 8 + j] = acc[h][j];
      if (col == 0) {
        sm_m[wid][h] = m_run[h];
        sm_l[wid][h] = l_run[h];
      }
    }
  }
  __syncthreads();

  // ---- merge the 4 waves, write final output or the split partial ---------
  for (int idx = threadIdx.x; idx < g_valid * D; idx += kThreads) {
    const int h = idx / D, d = idx - h * D;
    float m_all = sm_m[0][h];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) m_all = fmaxf(m_all, sm_m[w][h]);
    float l = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const float sc = exp2f(sm_m[w][h] - m_all);
      l += sm_l[w][h] * sc;
      o += sm_o[w][h][d] * sc;
    }
    const int hq = h0 + h;
    if (direct) {
      const float r = (l > 0.f) ? o / l * p.v_scale : 0.f;
      p.out[static_cast<int64_t>(b) * p.out_stride + static_cast<int64_t>(hq) * D + d] = f2bf(r);
    } else {
      const int64_t base = (static_cast<int64_t>(b) * p.num_q_heads + hq) * p.num_splits + split;
      p.ws_acc[base * D + d] = o;
      if (d == 0) {
        p.ws_ml[base * 2 + 0] = m_all;
        p.ws_ml[base * 2 + 1] = l;
      }
    }
  }
}

// Stage 2: merge split partials.  Grid (B, Hq), D threads.
__global__ void decode_stage2_kernel(const float* __restrict__ ws_acc, const float* __restrict__ ws_ml,
                                     const int32_t* __restrict__ seq_lens, uint16_t* __restrict__ out,
                                     int64_t out_stride, int num_q_heads, int head_dim,
                                     int num_splits, int min_chunk, int window, float v_scale) {
  const int b = blockIdx.x, hq = blockIdx.y, d = threadIdx.x;
  if (d >= head_dim) return;
  const int len = seq_lens[b];
  if (len == 0) return;  // stage 1 already wrote zeros
  const int64_t base = (static_cast<int64_t>(b) * num_q_heads + hq) * num_splits;
  const int kv_start = (window >= 0 && len - 1 - window > 0) ? len - 1 - window : 0;
  float m_all = kNegBig;
  int n_valid = 0;
  for (int s = 0; s < num_splits; ++s) {
    int c0, c1;
    split_range(len - kv_start, num_splits, min_chunk, s, c0, c1);
    if (c0 >= c1) break;
    ++n_valid;
    m_all = fmaxf(m_all, ws_ml[(base + s) * 2]);
  }
  float l = 0.f, o = 0.f;
  for (int s = 0; s < n_valid; ++s) {
    const float sc = exp2f(ws_ml[(base + s) * 2] - m_all);
    l += ws_ml[(base + s) * 2 + 1] * sc;
    o += ws_acc[(base + s) * head_dim + d] * sc;
  }
  const float r = (l > 0.f) ? o / l * v_scale : 0.f;
  out[static_cast<int64_t>(b) * out_stride + static_cast<int64_t>(hq) * head_dim + d] = f2bf(r);
}


template <int G, int D>
int launch_stage1(const DecodeParams& p, int batch, int group_size, int head_blocks, bool use_dpp,
                  hipStream_t stream) {
  dim3 grid(batch, p.num_kv_heads * head_blocks, p.num_splits);
  if (p.fmt.fp8) {
    hipLaunchKernelGGL((decode_stage1_kernel<G, D, true, true>), grid, dim3(kThreads), 0, stream, p, group_size, head_blocks);
  } else if (use_dpp) {
    hipLaunchKernelGGL((decode_stage1_kernel<G, D, true, false>), grid, dim3(kThreads), 0, stream, p, group_size, head_blocks);
  } else {
    hipLaunchKernelGGL((decode_stage1_kernel<G, D, false, false>), grid, dim3(kThreads), 0, stream, p, group_size, head_blocks);
  }
  return 0;
}

template <int D>
int dispatch_group(const DecodeParams& p, int batch, int group_size, bool use_dpp,
                   hipStream_t stream) {
  // Pad the GQA group to a power of two <= 8; larger groups are processed as
  // several 8-head blocks that re-read the KV rows.
  if (group_size == 1) return launch_stage1<1, D>(p, batch, group_size, 1, use_dpp, stream);
  if (group_size == 2) return launch_stage1<2, D>(p, batch, group_size, 1, use_dpp, stream);
  if (group_size <= 4) return launch_stage1<4, D>(p, batch, group_size, 1, use_dpp, stream);
  const int head_blocks = (group_size + 7) / 8;
  return launch_stage1<8, D>(p, batch, group_size, head_blocks, use_dpp, stream);
}

}  // namespace

extern "C" {

int sgl_amd_decode_attention_min_chunk(void) {
  SGL_CLEAR_STALE_ERROR(); return 128; }

int sgl_amd_decode_attention(const void* q, const void* k_cache, const void* v_cache, void* out,
                             const int32_t* req_to_token, int64_t req_to_token_stride,
                             const int64_t* req_pool_indices, const int32_t* seq_lens,
                             const int32_t* kv_indptr, int64_t batch, int num_q_heads,
                             int num_kv_heads, int head_dim, int64_t q_token_stride,
                             int64_t out_token_stride, int64_t k_cache_row_stride,
                             int64_t v_cache_row_stride, float sm_scale, int num_splits,
                             void* ws_acc, void* ws_ml, const int32_t* batch_order, int flags, void* stream) {
  return sgl_amd_decode_attention_ex(q, k_cache, v_cache, out, req_to_token, req_to_token_stride, req_pool_indices, seq_lens,
                                     kv_indptr, batch, num_q_heads, num_kv_heads, head_dim, q_token_stride, out_token_stride,
                                     k_cache_row_stride, v_cache_row_stride, sm_scale, num_splits, ws_acc, ws_ml, batch_order,
                                     flags, 0, 1.0f, 1.0f, 1, 0, -1, 0.0f, stream);
}

int sgl_amd_decode_attention_ex(const void* q, const void* k_cache, const void* v_cache, void* out,
                                const int32_t* req_to_token, int64_t req_to_token_stride,
                                const int64_t* req_pool_indices, const int32_t* seq_lens,
                                const int32_t* kv_indptr, int64_t batch, int num_q_heads,
                                int num_kv_heads, int head_dim, int64_t q_token_stride,
                                int64_t out_token_stride, int64_t k_cache_row_stride,
                                int64_t v_cache_row_stride, float sm_scale, int num_splits,
                                void* ws_acc, void* ws_ml, const int32_t* batch_order, int flags,
                                int kv_fp8, float k_scale, float v_scale, int page_size, int kv_layout_hnd,
                                int sliding_window, float logit_cap, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(k_cache_row_stride == v_cache_row_stride, "decode_attention: K and V pools must share a row stride");
  SGL_CHECK_ARG(!kv_fp8 || (k_scale > 0.f && v_scale > 0.f), "decode_attention: fp8 KV needs positive k_scale / v_scale");
  SGL_CHECK_ARG(logit_cap >= 0.f, "decode_attention: logit_cap must be >= 0 (0 = off)");
  SGL_CHECK_ARG(head_dim == 64 || head_dim == 128 || head_dim == 256,
                "decode_attention: head_dim=%d not supported (64/128/256)", head_dim);
  SGL_CHECK_ARG(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0,
                "decode_attention: num_q_heads=%d not a multiple of num_kv_heads=%d", num_q_heads, num_kv_heads);
  SGL_CHECK_ARG(num_splits >= 1 && num_splits <= 65535, "decode_attention: bad num_splits=%d", num_splits);
  SGL_CHECK_ARG(num_splits == 1 || (ws_acc && ws_ml), "decode_attention: split-KV needs the fp32 workspaces");
  SGL_CHECK_ARG(kv_indptr != nullptr || req_pool_indices != nullptr, "decode_attention: need req_pool_indices or kv_indptr");
  SGL_CHECK_ARG(q_token_stride % 8 == 0 && k_cache_row_stride % 8 == 0 && v_cache_row_stride % 8 == 0,
                "decode_attention: strides must be multiples of 8 elements");
  SGL_CHECK_ARG(batch <= 0x7fffffffLL, "decode_attention: batch too large");
  if (batch == 0) return 0;
  DecodeParams p;
  p.q = static_cast<const uint16_t*>(q);
  p.k_cache = static_cast<const uint16_t*>(k_cache);
  p.v_cache = static_cast<const uint16_t*>(v_cache);
  p.req_to_token = req_to_token;
  p.req_pool_indices = req_pool_indices;
  p.seq_lens = seq_lens;
  p.kv_indptr = kv_indptr;
  p.out = static_cast<uint16_t*>(out);
  p.ws_acc = static_cast<float*>(ws_acc);
  p.ws_ml = static_cast<float*>(ws_ml);
  p.q_stride = q_token_stride;
  p.out_stride = out_token_stride;
  p.kc_stride = k_cache_row_stride;
  p.vc_stride = v_cache_row_stride;
  p.r2t_stride = req_to_token_stride;
  p.num_q_heads = num_q_heads;
  p.num_kv_heads = num_kv_heads;
  p.num_splits = num_splits;
  p.min_chunk = sgl_amd_decode_attention_min_chunk();
  p.scale_log2 = sm_scale * 1.4426950408889634f * (kv_fp8 ? k_scale : 1.0f);
  p.v_scale = kv_fp8 ? v_scale : 1.0f;
  p.window = sliding_window;
  p.cap_log2 = logit_cap > 0.f ? logit_cap * 1.4426950408889634f : 0.f;
  p.inv_cap_log2 = logit_cap > 0.f ? 1.0f / p.cap_log2 : 0.f;
  SGL_CHECK_ARG(make_kv_format(&p.fmt, k_cache_row_stride, num_kv_heads, head_dim, page_size, kv_layout_hnd, kv_fp8),
                "decode_attention: HND pools need a power-of-two page_size (got %d)", page_size);
  p.batch_order = batch_order;
  const int group = num_q_heads / num_kv_heads;
  const bool use_dpp = (flags & SGL_AMD_ATTN_FLAG_NO_DPP) == 0;
  hipStream_t st = as_stream(stream);
  if (head_dim == 64) dispatch_group<64>(p, static_cast<int>(batch), group, use_dpp, st);
  else if (head_dim == 128) dispatch_group<128>(p, static_cast<int>(batch), group, use_dpp, st);
  else dispatch_group<256>(p, static_cast<int>(batch), group, use_dpp, st);
  SGL_CHECK_LAUNCH("decode_attention(stage1)");
  if (num_splits > 1) {
    hipLaunchKernelGGL(decode_stage2_kernel, dim3(batch, num_q_heads), dim3(head_dim), 0, st,
                       p.ws_acc, p.ws_ml, seq_lens, p.out, out_token_stride, num_q_heads, head_dim,
                       num_splits, p.min_chunk, p.window, p.v_scale);
    SGL_CHECK_LAUNCH("decode_attention(stage2)");
  }
  return 0;
}

}  // extern "C"

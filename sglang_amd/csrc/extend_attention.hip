// Ragged extend (prefill) attention over the paged token->KV pool, gfx950 MFMA.
//
// Replaces (reference, /root/reference/python/sglang):
//   kernels/ops/attention/extend_attention.py:304 _fwd_kernel, :753 extend_attention_fwd
// Oracle: srt/layers/attention/torch_native_backend.py:61-174, 279-336 -- the
// new K/V rows are written to the pool first, then every query token attends
// causally over req_to_token[req, 0:prefix+i+1].
//
// Structure (one workgroup = 4 waves = 128 "rows"):
//   * a row is a (query token, q head inside the GQA group) pair, so all
//     H_q/H_kv heads that share a kv head share ONE staged K/V tile (GQA reuse
//     in LDS instead of L2); each wave owns 32 rows = 2 MFMA M-tiles;
//   * a KV tile is 64 cached tokens.  Rows of the pool are gathered through
//     req_to_token with 16-byte loads, D/8 adjacent lanes per row, so every
//     gathered row is one fully coalesced D*2-byte read;
//   * K goes to LDS row-major with an XOR swizzle of the 16-byte chunks;
//     V is transposed on the way in (8x8 bf16 blocks transposed in registers,
//     written as 16-byte chunks of V^T) with the token order inside a chunk
//     chosen so the PV operand is a single ds_read_b128;
//   * both contractions run transposed on v_mfma_f32_16x16x32_bf16:
//         S^T = K . Q^T      O^T = V^T . P^T
//     which leaves every lane owning exactly one row per M-tile: running max,
//     running sum and the rescale factor never cross lanes, and the exp'd
//     S^T registers ARE the P^T operand of the second MFMA (no LDS round trip);
//   * fp32 accumulation, exp2 with the softmax scale folded in, bf16 output.
#include <type_traits>

#include "common.hpp"
#include "kv_format.hpp"
#include "sglang_amd.h"

using namespace sgl_amd;

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kStagers = 256;     // threads that gather K/V tiles (waves 0-1: V, waves 2-3: K)
constexpr int kRows = 128;      // rows per workgroup with two 16-row M-tiles per wave (64 with one)
constexpr int kKvTile = 64;     // cached tokens per tile
constexpr float kNegBig = -1.0e30f;

__device__ __forceinline__ bf16x8_t as_frag(const U4& v) { return __builtin_bit_cast(bf16x8_t, v); }

struct ExtendParams {
  const uint16_t* q;            // [T, Hq, D]
  uint16_t* out;                // [T, Hq, D]
  const uint16_t* k_cache;      // [slots, Hkv, D]
  const uint16_t* v_cache;
  const int32_t* req_to_token;  // [reqs, max_ctx]
  const int64_t* req_pool_indices;  // [B]
  const int32_t* seq_lens;      // [B] prefix + extend
  const int32_t* prefix_lens;   // [B]
  const int32_t* qo_indptr;     // [B+1] start of each request's query tokens
  int64_t q_stride, out_stride, kc_stride, vc_stride, r2t_stride;
  int num_kv_heads;
  int group;                    // q heads per kv head
  int tokens_per_tile;          // kRows / group
  int causal;
  float scale_log2;
  KvFormat fmt;                 // layout + element format of the pool rows
  float v_scale;                // fp8 rows: multiplied into the output (k_scale is folded into scale_log2)
  int window;                   // sliding window: kv position >= q position - window (extend_attention.py:480-485); < 0: off
  float cap_log2, inv_cap_log2; // logit soft cap (extend_attention.py:546-547), 0: off
  // custom mask (speculative-decoding verify, triton_backend.py:860-919; extend_attention.py USE_CUSTOM_MASK):
  // request b's [extend_len, kv_len] row-major uint8 mask starts at custom_mask + mask_indptr[b]; it REPLACES the
  // causal rule on the extend part and is AND-ed with "inside the prefix" on the prefix part
  const uint8_t* custom_mask;
  const int64_t* mask_indptr;
  int mask_skip_prefix;         // the prefix part of the mask is not consulted (extend_attention.py:774 skip_prefix_custom_mask)
  int num_tiles, batch;         // the double-buffered kernel's 1-D grid: query tiles per request, requests
};

template <int D>
struct Smem {
  U4 k[kKvTile * D / 8];        // [token][chunk ^ swz]
  U4 vt[D * kKvTile / 8];       // [d][token-chunk ^ swz]
};

// MTW = 16-row M-tiles per wave: 2 shares every staged K/V tile between 128 rows (fewest gathers per flop) at
// ~256 VGPRs = one workgroup per CU; 1 halves the rows per workgroup at ~130 VGPRs = three workgroups per CU
// (head dim 64; at 128 the 170-register budget of three workgroups spilled 16 registers inside the tile loop: two),
// whose waves cover each other's barrier, softmax and gather stalls.
// NWV = waves per workgroup: 8 (x 2 M-tiles = 256 rows) halves the K/V gathers per flop once more and puts two
// waves on every SIMD (the whole register file: 512 threads x 256 VGPRs) -- the form for long prefixes, where the
// kernel is bound by the gather traffic rather than by the matrix cores.
// (The bf16 8-wave case has its own kernel below, extend_attention_dbuf_kernel; this one serves fp8 pools and the
// 4-wave shapes.)
template <int D, int MTW, int NWV, bool FP8>
__global__ __launch_bounds__(64 * NWV, (MTW == 1 && NWV == 4) ? (D > 64 ? 2 : 3) : (NWV == 8 ? 2 : 1)) void extend_attention_kernel(ExtendParams p) {
  constexpr bool DEEP = false;
  __shared__ Smem<D> sm;
  const int block_x = blockIdx.x, block_z = blockIdx.z;
  constexpr int CPR = D / 8;            // 16-byte chunks per KV row
  constexpr int KC = D / 32;            // MFMA k-steps over the head dim
  constexpr int ND = D / 16;            // 16-wide output tiles over the head dim
  constexpr int NK_LOADS = kKvTile * CPR / 128;   // K 16-byte loads per K-thread
  constexpr int V_THREADS = 8 * CPR;    // threads that each transpose one 8x8 block

  int tile = block_x;
  const int kvh = blockIdx.y;
  int b = block_z;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int l15 = lane & 15;
  const int g = lane >> 4;

  const int q_begin = p.qo_indptr[b];
  const int ext_len = p.qo_indptr[b + 1] - q_begin;
  const int kv_len = p.seq_lens[b];
  const int prefix = p.prefix_lens[b];
  const int kv_begin = 0;
  const int32_t* idx_base = p.req_to_token + p.req_pool_indices[b] * p.r2t_stride;
  const int q0 = tile * p.tokens_per_tile;
  if (q0 >= ext_len) return;
  int q1 = q0 + p.tokens_per_tile;
  if (q1 > ext_len) q1 = ext_len;
  const uint8_t* cmask = p.custom_mask ? p.custom_mask + p.mask_indptr[b] : nullptr;
  const bool causal = p.causal && cmask == nullptr;
  const bool plain = cmask == nullptr && p.window < 0 && p.cap_log2 == 0.f;   // workgroup-uniform: the unmasked fast path is legal
  const int kv_end = causal ? (prefix + q1 < kv_len ? prefix + q1 : kv_len) : kv_len;
  // sliding window: the first query row of this tile bounds the earliest kv tile any row can see
  const int kv_first = (p.window >= 0 && prefix + q0 - p.window > 0) ? prefix + q0 - p.window : 0;
  const int t_first = (kv_begin > kv_first ? kv_begin : kv_first) / kKvTile;
  const int n_tiles = (kv_end + kKvTile - 1) / kKvTile;

  // ---- this lane's two rows (one per M-tile) ------------------------------
  int row_tok[MTW], row_limit[MTW], row_start[MTW];
  bool row_ok[MTW];
  int64_t row_off[MTW];   // element offset of (token, head) inside q / out
  U4 qfrag[MTW][KC];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    const int r = wid * (16 * MTW) + mt * 16 + l15;
    const int t = r / p.group;
    const int hg = r - t * p.group;
    row_tok[mt] = q0 + t;
    row_ok[mt] = (t < p.tokens_per_tile) && (q0 + t < q1);
    row_limit[mt] = row_ok[mt] ? (causal ? prefix + q0 + t + 1 : kv_len) : 0;
    if (row_limit[mt] > kv_len) row_limit[mt] = kv_len;
    row_start[mt] = (p.window >= 0 && prefix + q0 + t - p.window > 0) ? prefix + q0 + t - p.window : 0;
    row_off[mt] = static_cast<int64_t>(kvh * p.group + hg) * D;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      if (row_ok[mt]) {
        const int64_t qrow = q_begin + row_tok[mt];
        qfrag[mt][kc] = ld16(p.q + qrow * p.q_stride + row_off[mt] + kc * 32 + g * 8);
      } else {
        qfrag[mt][kc] = U4{0u, 0u, 0u, 0u};
      }
    }
  }

  f32x4_t ot[MTW][ND];
  float m_run[MTW], l_run[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    m_run[mt] = kNegBig;
    l_run[mt] = 0.f;
#pragma unroll
    for (int n = 0; n < ND; ++n) ot[mt][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging roles -------------------------------------------------------
  // waves 0-1: V (8x8 transposing blocks), waves 2-3: K (row-major).
  const int stid = DEEP ? (tid & 255) : tid;  // role inside the stager half
  const int half = DEEP ? (tid >> 8) : 0;     // wave-uniform: which tiles of the walk this half stages
  const bool is_v = stid < 128;
  const bool is_k = stid >= 128 && stid < kStagers;
  const int st = is_v ? stid : (stid - 128) & 127;
  const int st_c = st % CPR;                 // 16-byte column of the KV row
  const int st_r = st / CPR;
  const bool v_active = is_v && st < V_THREADS;
  const int v_blk32 = st_r >> 2, v_g = st_r & 3;
  constexpr int kStage = (NK_LOADS > 8) ? NK_LOADS : 8;
  U4 stage[kStage];

  // Slot ids run one tile ahead of the rows: a row load that had to wait for its own slot-id load would
  // park the wave for a full memory round trip in front of every tile's MFMAs.
  int32_t sidx[kStage];
  auto prefetch_idx = [&](int t) {
    const int kv0 = t * kKvTile;
    const int last = kv_end - 1;
    if (is_v) {
      if (v_active) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          // token order inside the chunk: i = 4a + r  <->  token 32*blk + 16a + 4g + r
          int tok = kv0 + 32 * v_blk32 + 16 * (i >> 2) + 4 * v_g + (i & 3);
          if (tok > last) tok = last;
          sidx[i] = idx_base[tok];
        }
      }
    } else if (is_k) {
#pragma unroll
      for (int i = 0; i < NK_LOADS; ++i) {
        int tok = kv0 + st_r + (128 / CPR) * i;
        if (tok > last) tok = last;
        sidx[i] = idx_base[tok];
      }
    }
  };
  auto prefetch_rows = [&]() {
    if (is_v) {
      if (v_active) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          stage[i] = ld_kv8<FP8>(kv_row(p.v_cache, p.fmt, sidx[i], kvh), st_c);
      }
    } else if (is_k) {
#pragma unroll
      for (int i = 0; i < NK_LOADS; ++i)
        stage[i] = ld_kv8<FP8>(kv_row(p.k_cache, p.fmt, sidx[i], kvh), st_c);
    }
  };

  auto commit = [&]() {
    if (is_v) {
      if (v_active) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(stage);  // stage[i] dword q -> w[4i+q]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int d = st_c * 8 + j;
          U4 o;
          uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const uint32_t a = w[4 * (2 * qd) + (j >> 1)];
            const uint32_t c = w[4 * (2 * qd + 1) + (j >> 1)];
            ow[qd] = (j & 1) ? ((a >> 16) | (c & 0xffff0000u)) : ((a & 0xffffu) | (c << 16));
          }
          const int chunk = (4 * v_blk32 + v_g) ^ ((d ^ (d >> 3)) & 7);
          sm.vt[d * 8 + chunk] = o;
        }
      }
    } else if (is_k) {
#pragma unroll
      for (int i = 0; i < NK_LOADS; ++i) {
        const int row = st_r + (128 / CPR) * i;
        sm.k[row * CPR + (st_c ^ (row & (CPR - 1)))] = stage[i];
      }
    }
  };

  constexpr int STEP = DEEP ? 2 : 1;          // tiles between two stagings of the same half
  if (t_first + half < n_tiles) {
    prefetch_idx(t_first + half);
    prefetch_rows();
    if (t_first + half + STEP < n_tiles) prefetch_idx(t_first + half + STEP);
  }

  for (int t = t_first; t < n_tiles; ++t) {
    const bool mine = !DEEP || ((t - t_first) & 1) == half;      // wave-uniform
    __syncthreads();   // every wave is done reading the previous tile
    if (mine) commit();
    __syncthreads();
    if (mine && t + STEP < n_tiles) {
      prefetch_rows();                                   // this half's next tile, slot ids loaded one staging ago
      if (t + 2 * STEP < n_tiles) prefetch_idx(t + 2 * STEP);
    }

    const int kv0 = t * kKvTile;

    // ---- S^T = K . Q^T ---------------------------------------------------
    f32x4_t st_acc[MTW][4];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) st_acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int row = nt * 16 + l15;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const U4 kf = sm.k[row * CPR + ((kc * 4 + g) ^ (row & (CPR - 1)))];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
          st_acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              as_frag(kf), as_frag(qfrag[mt][kc]), st_acc[mt][nt], 0, 0, 0);
      }
    }

    // ---- online softmax; lane owns row (l15 + 16 mt), tokens 16nt + 4g + r --
    // The VALU work here, not the MFMAs, sets the pace of this kernel, so: hardware bf16 packing and
    // v_exp_f32, the scale folded into one fma per score, no masking on tiles that lie inside every row's
    // causal limit, and no rescale of O when no row of the wave raised its maximum.
    U4 pfrag[MTW][2];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const bool full = plain && __ballot(kv0 + kKvTile > row_limit[mt]) == 0ull;     // wave-uniform
      float pv[4][4];
      float m_new, psum = 0.f;
      if (full) {
        float mx = st_acc[mt][0][0];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st_acc[mt][nt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        m_new = fmaxf(m_run[mt], mx * p.scale_log2);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = fast_exp2(fmaf(st_acc[mt][nt][r], p.scale_log2, -m_new));
            pv[nt][r] = e;
            psum += e;
          }
      } else {
        float sv[4][4];
        float mx = kNegBig;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kvpos = kv0 + nt * 16 + g * 4 + r;
            bool on = kvpos < row_limit[mt] && kvpos >= row_start[mt];
            if (cmask != nullptr && on && !(p.mask_skip_prefix && kvpos < prefix))   // the mask row of this query token over all kv_len positions
              on = cmask[static_cast<int64_t>(row_tok[mt]) * kv_len + kvpos] != 0;
            float s = st_acc[mt][nt][r] * p.scale_log2;
            if (p.cap_log2 != 0.f) s = soft_cap_log2(s, p.cap_log2, p.inv_cap_log2);
            s = on ? s : kNegBig;
            sv[nt][r] = s;
            mx = fmaxf(mx, s);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        m_new = fmaxf(m_run[mt], mx);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = (sv[nt][r] > 0.5f * kNegBig) ? fast_exp2(sv[nt][r] - m_new) : 0.f;   // masked scores add nothing, even while the row's maximum is still the sentinel
            pv[nt][r] = e;
            psum += e;
          }
      }
      const float alpha = fast_exp2(m_run[mt] - m_new);
      m_run[mt] = m_new;
      l_run[mt] = l_run[mt] * alpha + psum;
      if (__ballot(alpha != 1.0f) != 0ull) {
#pragma unroll
        for (int n = 0; n < ND; ++n) ot[mt][n] *= alpha;
      }
      // P^T operand for k-step kk: [P(nt=2kk, r=0..3), P(nt=2kk+1, r=0..3)]
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        pfrag[mt][kk].x = pack_bf2(pv[2 * kk][0], pv[2 * kk][1]);
        pfrag[mt][kk].y = pack_bf2(pv[2 * kk][2], pv[2 * kk][3]);
        pfrag[mt][kk].z = pack_bf2(pv[2 * kk + 1][0], pv[2 * kk + 1][1]);
        pfrag[mt][kk].w = pack_bf2(pv[2 * kk + 1][2], pv[2 * kk + 1][3]);
      }
    }

    // ---- O^T += V^T . P^T ------------------------------------------------
#pragma unroll
    for (int n = 0; n < ND; ++n) {
      const int d = n * 16 + l15;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const U4 vf = sm.vt[d * 8 + ((kk * 4 + g) ^ ((d ^ (d >> 3)) & 7))];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
          ot[mt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(vf), as_frag(pfrag[mt][kk]),
                                                               ot[mt][n], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane holds O^T[d = 16n + 4g + r][row] ----------------------
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    float l = l_run[mt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (!row_ok[mt]) continue;
    const float inv = (l > 0.f) ? p.v_scale / l : 0.f;
    uint16_t* op = p.out + static_cast<int64_t>(q_begin + row_tok[mt]) * p.out_stride + row_off[mt];
#pragma unroll
    for (int n = 0; n < ND; ++n) {
      uint2 w;
      w.x = pack_bf2(ot[mt][n][0] * inv, ot[mt][n][1] * inv);
      w.y = pack_bf2(ot[mt][n][2] * inv, ot[mt][n][3] * inv);
      *reinterpret_cast<uint2*>(op + n * 16 + g * 4) = w;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The bf16 8-wave form (256 rows per workgroup, two M-tiles per wave): the shape every prefill of the bench runs.
// Same products, lane maps and softmax as above; what differs is how the K/V tiles travel:
//   * two LDS images per operand and ONE barrier per tile: all 512 threads ask for the rows of tile t + 1 (four 16-byte
//     loads each) before they multiply tile t, and write them to the other image once tile t is done -- the gather has
//     the whole tile's matrix work to land, and no wave waits for a commit between two barriers;
//   * V stays row-major: per 16 head dims a [64 tokens][16] sub-image (rows of 32 B, sub-images 2080 B apart so that the
//     16 pieces of a row land on 16 distinct bank slots); the V^T operand of O^T = V^T . P^T comes out of
//     ds_read_b64_tr_b16 (lane i of a 16-lane group, element j <- row j, column i of the 4 x 16 block the group points
//     at), two reads per operand: tokens 32 kk + 4 g + j and 32 kk + 16 + 4 g + j -- the token order of the P^T
//     fragments built from the S^T accumulators.  No transposition through the VALU;
//   * the running maximum follows every tile (EXT_DEFER_MAX = 0).  Round 3 raised it only when a row outgrew it by 2^8
//     (the rescale of the 64 output accumulators is the largest VALU item of a tile: 2.6-5.5 % of the kernel), but then
//     the row's largest P is exp2(s - m_stale) instead of exp2(0) = 1: the dominant term of a peaked row picks up a bf16
//     rounding error it does not have in the reference's kernels.  Against the fp64 result the deferred form measured
//     rms 0.378 / 0.389 / 0.205 bf16 ulp (flat / unit / peaked scores) where torch's bf16 SDPA has 0.370 / 0.368 / 0.145;
//     with the exact maximum this kernel's figures are the SDPA's to the last digit (benchmarks/r04_exp2_extend_error.py,
//     profiles/r04_exp2_extend_error.json).
#ifndef EXT_DEFER_MAX
#define EXT_DEFER_MAX 0.0f
#endif
constexpr float kDeferMax = EXT_DEFER_MAX;   // log2 units

#ifdef EXT_TRACE
// phase probe (benchmarks/r02_exp9_ext_trace.py): per wave, shader clocks spent in [row loads issued, S^T, softmax,
// PV, commit, barrier] summed over the tiles of the walk, and the tile count
__device__ uint64_t* g_ext_trace = nullptr;
#define EXT_T(i)                                   \
  do {                                             \
    const uint64_t now_ = __builtin_readcyclecounter(); \
    tacc[i] += now_ - tprev;                       \
    tprev = now_;                                  \
  } while (0)
#else
#define EXT_T(i)
#endif

// Maximum over the four 16-lane rows of a wave (the lanes l, l + 16, l + 32, l + 48 hold one query row's scores of
// different tokens): v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second,
// v_permlane32_swap the upper half of the first with the lower half of the second; on two copies of x each leaves
// (own, partner) in the two registers.  Inline asm on purpose: with the builtin on two copies of ONE value hipcc kept
// only the first result (max(r0, r0)), which turned the reduction into "everybody takes row 0's maximum" -- harmless
// until a row's largest score sits in another lane row and exceeds the others by 2^128 (tests/test_kernels_gpu.py,
// every_workgroup_shape).  The s_nops cover the VALU-write -> permlane-read hazard the assembler does not pad.
__device__ __forceinline__ float row_groups_max(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  a = fmaxf(a, b);
  b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

template <int D>
struct SmemDbuf {
  static constexpr int kVSub = 64 * 32 + 32;                   // bytes of one 16-dim V sub-image + bank skew
  U4 k[2][kKvTile * D / 8];                                    // [token][chunk ^ swz]
  U4 v[2][(D / 16) * kVSub / 16];
};

typedef short v4s16_t __attribute__((ext_vector_type(4)));

// The Q^T fragments stay in 32 registers per lane (round 2 kept them in a lane-private 64 KB LDS image: 5-8 % slower on
// the bench's prefill shapes, profiles/r03_exp0_qreg.txt).  Token-major bf16 pools only (the layout the bench and the
// reference's default pool use): paged head-major pools take the general kernel, so that no gathered row costs a branch.
template <int D>
__global__ __launch_bounds__(512, 2) void extend_attention_dbuf_kernel(ExtendParams p) {
  __shared__ SmemDbuf<D> sm;
  constexpr int MTW = 2;
  constexpr int CPR = D / 8;            // 16-byte chunks per KV row
  constexpr int KC = D / 32;            // MFMA k-steps over the head dim
  constexpr int ND = D / 16;            // 16-wide output tiles over the head dim
  constexpr int ROWS_PER_PASS = 512 / CPR;
  constexpr int LOADS = kKvTile / ROWS_PER_PASS;     // 16-byte loads per thread per operand per tile (2 at D = 128)
  constexpr int VSUB16 = SmemDbuf<D>::kVSub / 16;    // sub-image stride in 16-byte units

  // 1-D grid -> (query tile, kv head, request).  Two things ride on the order: (1) under a causal mask the last query
  // tile of a request walks the most KV tiles, so ALL pairs' heaviest tiles are handed out first and the light ones
  // fill the tail of the launch (4 x 1024 tokens: 115 -> 76 us; pair after pair measured 100); (2) workgroup L runs
  // on XCD L % 8 and every query tile of a (request, kv head) pair reads the same K/V rows, so a pair always lands on
  // the same XCD: what one of its tiles pulled from HBM the others find in that XCD's L2.
  int tile, kvh, b;
  {
    const int L = blockIdx.x;
    const int pairs = p.batch * p.num_kv_heads;
    int pair, rank;
    if (pairs % 8 == 0) {
      const int per_xcd = pairs / 8;
      const int j = L >> 3;
      pair = (j % per_xcd) * 8 + (L & 7);
      rank = j / per_xcd;
    } else {
      pair = L % pairs;
      rank = L / pairs;
    }
    tile = p.num_tiles - 1 - rank;
    kvh = pair % p.num_kv_heads;
    b = pair / p.num_kv_heads;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int l15 = lane & 15;
  const int g = lane >> 4;

  const int q_begin = p.qo_indptr[b];
  const int ext_len = p.qo_indptr[b + 1] - q_begin;
  const int kv_len = p.seq_lens[b];
  const int prefix = p.prefix_lens[b];
  const int32_t* idx_base = p.req_to_token + p.req_pool_indices[b] * p.r2t_stride;
  const int q0 = tile * p.tokens_per_tile;
  if (q0 >= ext_len) return;
  int q1 = q0 + p.tokens_per_tile;
  if (q1 > ext_len) q1 = ext_len;
  const bool causal = p.causal != 0;                // (sliding window / soft cap / custom mask: the general kernel)
  const int kv_end = causal ? (prefix + q1 < kv_len ? prefix + q1 : kv_len) : kv_len;
  const int t_first = 0;
  const int n_tiles = (kv_end + kKvTile - 1) / kKvTile;

  // ---- this lane's two rows (one per M-tile) ------------------------------
  int row_tok[MTW], row_limit[MTW];
  bool row_ok[MTW];
  int row_off[MTW];       // element offset of the row's head inside a q / out token
  U4 qreg[MTW][KC];                                 // this lane's Q^T fragments
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    const int r = wid * (16 * MTW) + mt * 16 + l15;
    const int t = r / p.group;
    const int hg = r - t * p.group;
    row_tok[mt] = q0 + t;
    row_ok[mt] = (t < p.tokens_per_tile) && (q0 + t < q1);
    row_limit[mt] = row_ok[mt] ? (causal ? prefix + q0 + t + 1 : kv_len) : 0;
    if (row_limit[mt] > kv_len) row_limit[mt] = kv_len;
    row_off[mt] = (kvh * p.group + hg) * D;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      U4 qf = U4{0u, 0u, 0u, 0u};
      if (row_ok[mt]) {
        const int64_t qrow = q_begin + row_tok[mt];
        qf = ld16(p.q + qrow * p.q_stride + row_off[mt] + kc * 32 + g * 8);
      }
      qreg[mt][kc] = qf;
    }
  }

  f32x4_t ot[MTW][ND];
  float m_run[MTW], l_run[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    m_run[mt] = kNegBig;
    l_run[mt] = 0.f;
#pragma unroll
    for (int n = 0; n < ND; ++n) ot[mt][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging: thread = (row st_r + ROWS_PER_PASS i, 16-byte chunk st_c) of a K tile and of a V tile ----
  const int st_c = tid % CPR, st_r = tid / CPR;
  int32_t idx_k[LOADS], idx_v[LOADS];       // slot ids of the K tile / V tile this thread fetches next
  U4 kst[LOADS], vst[LOADS];
  auto load_idx = [&](int t, int32_t (&dst)[LOADS]) {
    const int last = kv_end - 1;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      int tok = t * kKvTile + st_r + ROWS_PER_PASS * i;
      if (tok > last) tok = last;
      dst[i] = idx_base[tok];
    }
  };
  // token-major pool: row = base + slot * row bytes, one v_mad_u64_u32 per gathered row
  const unsigned char* k_rows = reinterpret_cast<const unsigned char*>(p.k_cache) + static_cast<uint64_t>(kvh) * p.fmt.head_stride + st_c * 16;
  const unsigned char* v_rows = reinterpret_cast<const unsigned char*>(p.v_cache) + static_cast<uint64_t>(kvh) * p.fmt.head_stride + st_c * 16;
  auto load_k = [&]() {
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      kst[i] = ld16(k_rows + static_cast<uint64_t>(static_cast<uint32_t>(idx_k[i])) * p.fmt.page_stride);
    }
  };
  auto load_v = [&]() {
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      vst[i] = ld16(v_rows + static_cast<uint64_t>(static_cast<uint32_t>(idx_v[i])) * p.fmt.page_stride);
    }
  };
  auto commit_k = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int row = st_r + ROWS_PER_PASS * i;
      sm.k[buf][row * CPR + (st_c ^ ((row * CPR / 16) & (CPR - 1)))] = kst[i];
    }
  };
  auto commit_v = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int row = st_r + ROWS_PER_PASS * i;
      sm.v[buf][(st_c >> 1) * VSUB16 + row * 2 + (st_c & 1)] = vst[i];
    }
  };
  // V^T fragment reads: byte offset of this lane inside a sub-image, token rows 4 g + (l15 >> 2), 8-byte piece l15 & 3
  const int v_lane = (4 * g + (l15 >> 2)) * 32 + (l15 & 3) * 8;

  // ---- the two halves of the workgroup run half a tile apart ---------------------------------------------------
  // Waves w and w + 4 share a SIMD.  Time is cut into slots, one barrier each; in slot s a wave is either in its
  // matrix block  [ O^T += V^T P^T of tile u - 1,  S^T = K Q^T of tile u ]  or in its vector block  [ softmax of
  // tile u, staging ]: waves 0-3 multiply in the even slots (u = s / 2), waves 4-7 in the odd ones, so on every SIMD
  // one wave feeds the matrix core while the other one runs the exp / max / pack work on the VALU (in lock-step both
  // queue for the same unit: 7100 clocks per tile, 2780 of them softmax, benchmarks/r02_exp9_ext_trace.py).
  // Staging rides at the end of the matrix blocks: the waves multiplying in slots 2 v and 2 v + 1 write their
  // share of K tile v + 1 (its image was last read in slot 2 v - 1, is first read in slot 2 v + 2) and of V tile v
  // (last read 2 v - 1, first read 2 v + 2), then ask for K tile v + 2 and V tile v + 1: a tile-time in flight.
  const int grp = __builtin_amdgcn_readfirstlane(wid >> 2);
  const int t0 = t_first;
  if (t0 < n_tiles) {
    load_idx(t0, idx_k);
    load_idx(t0, idx_v);
    load_k();
    if (t0 + 1 < n_tiles) load_idx(t0 + 1, idx_k);
    commit_k(0);
    load_v();                                        // V tile t0, written in slot 2 t0 / 2 t0 + 1
    if (t0 + 1 < n_tiles) {
      load_idx(t0 + 1, idx_v);
      load_k();                                      // K tile t0 + 1, same
    }
    if (t0 + 2 < n_tiles) load_idx(t0 + 2, idx_k);
  }
  __syncthreads();

  f32x4_t st_acc[MTW][4];
  U4 pfrag[MTW][2];
#ifdef EXT_TRACE
  uint64_t tacc[6] = {0, 0, 0, 0, 0, 0};
  uint64_t tprev = __builtin_readcyclecounter();
  const uint64_t tstart = tprev;
#endif
  // matrix block of tile u: O^T += V^T P^T of tile u - 1, then S^T of tile u.  The LDS fragment reads run ONE MFMA
  // GROUP AHEAD of the products that use them (two register sets per operand): in program order hipcc waited for every
  // group's own ds_reads with the matrix core drained (64 MFMAs took 1720 clocks instead of 1024).
  auto read_v = [&](const unsigned char* vbase, int n, U4 (&dst)[2]) {
    typedef __attribute__((address_space(3))) v4s16_t* lds_v4_t;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const v4s16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(vbase + n * SmemDbuf<D>::kVSub + kk * 1024));
      const v4s16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(vbase + n * SmemDbuf<D>::kVSub + kk * 1024 + 512));
      dst[kk].x = __builtin_bit_cast(uint2, lo).x; dst[kk].y = __builtin_bit_cast(uint2, lo).y;
      dst[kk].z = __builtin_bit_cast(uint2, hi).x; dst[kk].w = __builtin_bit_cast(uint2, hi).y;
    }
  };
  auto read_k = [&](const U4* kimg, int nt, U4 (&dst)[KC]) {
    const int row = nt * 16 + l15;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) dst[kc] = kimg[row * CPR + ((kc * 4 + g) ^ ((row * CPR / 16) & (CPR - 1)))];
  };
  auto matrix_block = [&](int u) {
    EXT_T(5);
    const bool have_pv = u - 1 >= t0 && u - 1 < n_tiles, have_s = u < n_tiles;
    const U4* kimg = sm.k[(u - t0) & 1];
    U4 kf[2][KC];
    if (have_s) {                                       // the first K fragments travel under the PV products
      read_k(kimg, 0, kf[0]);
      __builtin_amdgcn_sched_group_barrier(0x100, KC, 0);
    }
    if (have_pv) {
      // ---- O^T += V^T . P^T (tile u - 1) ----------------------------------
      const unsigned char* vbase = reinterpret_cast<const unsigned char*>(sm.v[(u - 1 - t0) & 1]) + v_lane;
      U4 vf[2][2];
      read_v(vbase, 0, vf[0]);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int n = 0; n < ND; ++n) {
        if (n + 1 < ND) read_v(vbase, n + 1, vf[(n + 1) & 1]);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt)
            ot[mt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(vf[n & 1][kk]), as_frag(pfrag[mt][kk]), ot[mt][n], 0, 0, 0);
        // pin the order the source states: the NEXT group's four reads, then this group's four products
        if (n + 1 < ND) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
    }
    EXT_T(3);
    if (have_s) {
      // ---- S^T = K . Q^T (tile u) -----------------------------------------
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) st_acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (nt + 1 < 4) read_k(kimg, nt + 1, kf[(nt + 1) & 1]);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt)
            st_acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(kf[nt & 1][kc]), as_frag(qreg[mt][kc]), st_acc[mt][nt], 0, 0, 0);
        if (nt + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, KC, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * KC, 0);
      }
    }
    EXT_T(1);
    // ---- staging of this slot pair (v = u): K tile v + 1 and V tile v go to LDS, K tile v + 2 and V tile v + 1 are
    // asked for.  Behind the MFMAs: the writes wait for rows requested a tile ago while the matrix core drains.
    const int v = u;
    if (v < n_tiles) commit_v((v - t0) & 1);
    if (v + 1 < n_tiles) commit_k((v + 1 - t0) & 1);
    EXT_T(4);
    EXT_T(0);
  };
  // vector block: softmax of tile u (the scores the wave holds)
  auto vector_block = [&](int u) {
    EXT_T(5);
    if (u >= t0 && u < n_tiles) {
      const int kv0 = u * kKvTile;
      // ---- online softmax; lane owns row (l15 + 16 mt), tokens 16 nt + 4 g + r.  Both M-tiles go through each step
      // together (two independent chains in flight), the row maximum crosses the four 16-lane rows of the wave with
      // v_permlane16_swap / v_permlane32_swap (no LDS round trip), the scale rides in the exp's fma.
      const bool full = __ballot(kv0 + kKvTile > row_limit[0] || kv0 + kKvTile > row_limit[1]) == 0ull;   // wave-uniform
      float mx[MTW];
      if (full) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          float m = st_acc[mt][0][0];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, st_acc[mt][nt][r]);
          mx[mt] = m * p.scale_log2;
        }
      } else {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          float m = kNegBig;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kvpos = kv0 + nt * 16 + g * 4 + r;
              // a masked score becomes the sentinel BEFORE the scale: exp2(fma(sentinel, scale, -m)) = 0
              const float sc = kvpos < row_limit[mt] ? st_acc[mt][nt][r] : kNegBig;
              st_acc[mt][nt][r] = sc;
              m = fmaxf(m, sc);
            }
          mx[mt] = m > 0.5f * kNegBig ? m * p.scale_log2 : kNegBig;
        }
      }
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) mx[mt] = row_groups_max(mx[mt]);
      if (__ballot(mx[0] > m_run[0] + kDeferMax || mx[1] > m_run[1] + kDeferMax) != 0ull) {
        // some row outgrew its maximum: raise them all
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const float m_new = fmaxf(m_run[mt], mx[mt]);
          const float alpha = fast_exp2(m_run[mt] - m_new);
          m_run[mt] = m_new;
          l_run[mt] *= alpha;
#pragma unroll
          for (int n = 0; n < ND; ++n) ot[mt][n] *= alpha;
        }
      }
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        float psum = 0.f;
        float pv[4][4];
        // rows that have seen no key yet keep the sentinel as maximum: their (all masked) scores must give 0, not 1
        const float neg_m = m_run[mt] > 0.5f * kNegBig ? -m_run[mt] : 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = fast_exp2(fmaf(st_acc[mt][nt][r], p.scale_log2, neg_m));
            pv[nt][r] = e;
            psum += e;
          }
        l_run[mt] += psum;
        // P^T operand for k-step kk: [P(nt=2kk, r=0..3), P(nt=2kk+1, r=0..3)]
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          pfrag[mt][kk].x = pack_bf2(pv[2 * kk][0], pv[2 * kk][1]);
          pfrag[mt][kk].y = pack_bf2(pv[2 * kk][2], pv[2 * kk][3]);
          pfrag[mt][kk].z = pack_bf2(pv[2 * kk + 1][0], pv[2 * kk + 1][1]);
          pfrag[mt][kk].w = pack_bf2(pv[2 * kk + 1][2], pv[2 * kk + 1][3]);
        }
      }
    }
    EXT_T(2);
    // ---- requests for the next commit (matrix block u + 1 writes K tile u + 2 and V tile u + 1): the registers are
    // free since this wave's matrix block u wrote their previous contents to LDS
    const int v = u;
    // (each id array is loaded straight into its own registers: handing idx_k over to idx_v by a copy made hipcc wait
    // for EVERY load in flight at the end of this block -- 900-1400 clocks per tile)
    if (v + 1 < n_tiles) {
      load_v();                                     // V tile v + 1 (idx_v holds its slot ids)
      if (v + 2 < n_tiles) load_idx(v + 2, idx_v);
    }
    if (v + 2 < n_tiles) {
      load_k();                                     // K tile v + 2
      if (v + 3 < n_tiles) load_idx(v + 3, idx_k);
    }
    EXT_T(0);
  };
  // two straight-line loops instead of one loop with a role switch per slot (which hipcc spilled around)
  if (grp == 0) {
    for (int u = t0; u <= n_tiles; ++u) {
      matrix_block(u);
      __syncthreads();
      EXT_T(5);
      vector_block(u);
      __syncthreads();
      EXT_T(5);
    }
  } else {
    __syncthreads();                                  // the half-tile offset: an empty first slot
    for (int u = t0; u < n_tiles; ++u) {
      matrix_block(u);
      __syncthreads();
      EXT_T(5);
      vector_block(u);
      __syncthreads();
      EXT_T(5);
    }
    matrix_block(n_tiles);
    __syncthreads();
  }
#ifdef EXT_TRACE
  if (g_ext_trace && lane == 0) {
    uint64_t* o = g_ext_trace + static_cast<int64_t>(blockIdx.x) * 64 + wid * 8;
    for (int i = 0; i < 6; ++i) o[i] = tacc[i];
    o[6] = n_tiles - t_first;
    o[7] = tstart;
  }
#endif

  // ---- epilogue: lane holds O^T[d = 16n + 4g + r][row] ----------------------
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    float l = l_run[mt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (!row_ok[mt]) continue;
    const float inv = (l > 0.f) ? 1.0f / l : 0.f;
    uint16_t* op = p.out + static_cast<int64_t>(q_begin + row_tok[mt]) * p.out_stride + row_off[mt];
#pragma unroll
    for (int n = 0; n < ND; ++n) {
      uint2 w;
      w.x = pack_bf2(ot[mt][n][0] * inv, ot[mt][n][1] * inv);
      w.y = pack_bf2(ot[mt][n][2] * inv, ot[mt][n][3] * inv);
      *reinterpret_cast<uint2*>(op + n * 16 + g * 4) = w;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The 32x32 form of the bf16 8-wave kernel: what every prefill of the bench runs (token-major bf16 pool, causal or not,
// no window / cap / mask; the ping-pong kernel above stays selectable for A/B runs).  What changes against it:
//   * v_mfma_f32_32x32x16_bf16: one wave = 32 rows (query token x head of the group), a 64-key tile = two 32-key blocks.
//     S^T = K . Q^T leaves lane (q = lane & 31, h = lane >> 5) with 32 scores of ONE row (keys 32 blk + (r & 3) +
//     8 (r >> 2) + 4 h): the row maximum is an in-lane chain + ONE v_permlane32_swap, the row sum stays a per-lane partial
//     until the epilogue, and bf16(P) of registers 8 ks .. 8 ks + 7 IS the B operand of O^T += V^T . P^T for the 16 keys
//     {32 blk + 16 ks + (j & 3) + 8 (j >> 2) + 4 h}; the V^T operand is read in that key order (two ds_read_b64_tr_b16:
//     keys kb + 4 h + 0..3 and kb + 8 + 4 h + 0..3).  Half the LDS fragment reads per flop of the 16x16x32 shape;
//   * no role split and ONE barrier per tile; the vector work of a tile rides between the matrix instructions of its
//     NEIGHBOURS (two score sets, see the kernel's own header);
//   * V image per 32 head dims: [64 tokens][64 B] (+ a bank skew between the sub-images), so that the four 16-lane groups
//     of a transposing read cover 512 contiguous bytes;
//   * a raised maximum is applied to O and l when everything exponentiated against the old one is
//     inside them (the order the softmax-rescale hazard asks for);
//   * the epilogue swaps 4-dim pieces between the lane halves and stores 16 bytes per lane.
// Measured on MI355X (benchmarks/r03_exp4_extend_32x32.py, profiles/r03_exp4_extend_32x32.json): 4 x 1024 cold 62 -> 52 us, 60 x 128
// over 896 warm 197 -> 167 us, 2 x 4096 304 -> 279 us, 8 x 2952 676 -> 629 us.  A one-score-set version of the same
// form (S^T(t) -> max(t) -> [products of t - 1 || exponentials of t]; commit 21c58e7..: 2-6 % slower) and its 4-wave /
// two-workgroups-per-CU variant (slower again: twice the row requests per flop) were measured and dropped.
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int D>
struct SmemLadder {
  static constexpr int kVSub = 64 * 64 + (D == 128 ? 64 : 128);   // bytes of one 32-dim V sub-image + bank skew
  U4 k[2][kKvTile * D / 8];                                       // [token][chunk ^ swz]
  U4 v[2][(D / 32) * kVSub / 16];
};

__device__ __forceinline__ float lane_pair_max(float x) {         // max over lanes l and l ^ 32
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

// A step of the walk is two phases of products with the vector work of the NEIGHBOURING tiles spread between them:
//     phase 1:  S^T(t + 1) = K(t + 1) . Q^T   ||  P(t) = exp2(S(t) scale - m)  and the row requests of the tiles ahead
//     phase 2:  O^T += V^T(t) . P^T(t)        ||  mask + row maxima of S(t + 1)
//     then the rescale decision for tile t + 1 (everything exponentiated against the old maximum -- P(t) -- is inside
//     O and l by then), and ONE barrier.
// Every product carries <= 7 vector instructions, none of them waiting for the product next to it; what is left outside
// the products' shadow is the decision (a permlane, two ballots), four LDS writes and the barrier.
// FP8: the pool holds e4m3 rows (8 bytes per lane and chunk, held raw in the staging registers, widened to bf16 -- exact --
// on their way into the LDS images; k_scale rides in the softmax scale, v_scale in the epilogue).
template <int D, int NWV, bool FP8>
__global__ __launch_bounds__(64 * NWV, 2) void extend_attention_pipe_kernel(ExtendParams p) {
  __shared__ SmemLadder<D> sm;
  constexpr int CPR = D / 8;            // 16-byte chunks per KV row
  constexpr int KS = D / 16;            // MFMA k-steps over the head dim
  constexpr int ND = D / 32;            // 32-wide output blocks over the head dim
  constexpr int ROWS_PER_PASS = 64 * NWV / CPR;
  constexpr int LOADS = kKvTile / ROWS_PER_PASS;     // 16-byte loads per thread per operand per tile (2 at D = 128, 8 waves)
  constexpr int VSUB = SmemLadder<D>::kVSub;
  constexpr int NPV = ND * 4;           // PV products per tile

  // grid -> (query tile, kv head, request): as in the ping-pong kernel (heaviest tiles first, a pair stays on one XCD)
  int tile, kvh, b;
  {
    const int L = blockIdx.x;
    const int pairs = p.batch * p.num_kv_heads;
    int pair, rank;
    if (pairs % 8 == 0) {
      const int per_xcd = pairs / 8;
      const int j = L >> 3;
      pair = (j % per_xcd) * 8 + (L & 7);
      rank = j / per_xcd;
    } else {
      pair = L % pairs;
      rank = L / pairs;
    }
    tile = p.num_tiles - 1 - rank;
    kvh = pair % p.num_kv_heads;
    b = pair / p.num_kv_heads;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int l31 = lane & 31;
  const int hi = lane >> 5;

  const int q_begin = p.qo_indptr[b];
  const int ext_len = p.qo_indptr[b + 1] - q_begin;
  const int kv_len = p.seq_lens[b];
  const int prefix = p.prefix_lens[b];
  const int32_t* idx_base = p.req_to_token + p.req_pool_indices[b] * p.r2t_stride;
  const int q0 = tile * p.tokens_per_tile;
  if (q0 >= ext_len) return;
  int q1 = q0 + p.tokens_per_tile;
  if (q1 > ext_len) q1 = ext_len;
  const bool causal = p.causal != 0;
  const int kv_end = causal ? (prefix + q1 < kv_len ? prefix + q1 : kv_len) : kv_len;
  const int n_tiles = (kv_end + kKvTile - 1) / kKvTile;

  // ---- this lane's row (both lane halves of a row hold the same query, other head dims / keys) ----
  const int r_row = wid * 32 + l31;
  const int r_tok = r_row / p.group;
  const int r_hg = r_row - r_tok * p.group;
  const bool row_ok = (r_tok < p.tokens_per_tile) && (q0 + r_tok < q1);
  int row_limit = row_ok ? (causal ? prefix + q0 + r_tok + 1 : kv_len) : 0;
  if (row_limit > kv_len) row_limit = kv_len;
  const int row_off = (kvh * p.group + r_hg) * D;
  U4 qreg[KS];                                      // Q^T operand of k-step s: dims 16 s + 8 h .. + 7 of the row
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    U4 qf = U4{0u, 0u, 0u, 0u};
    if (row_ok) qf = ld16(p.q + static_cast<int64_t>(q_begin + q0 + r_tok) * p.q_stride + row_off + s * 16 + hi * 8);
    qreg[s] = qf;
  }

  f32x16_t ot[ND];
#pragma unroll
  for (int n = 0; n < ND; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[n][r] = 0.f;
  float m_run = kNegBig, l_run = 0.f;               // l: this lane's share of the row sum

  // ---- staging: thread = (row st_r + ROWS_PER_PASS i, 16-byte chunk st_c) of a K tile and of a V tile ----
  const int st_c = tid % CPR, st_r = tid / CPR;
  int32_t idx_k[LOADS], idx_v[LOADS];
  typedef typename std::conditional<FP8, uint2, U4>::type Stage;   // a gathered 8-dim piece as it travels: raw bytes
  Stage kst[LOADS], vst[LOADS];
  auto ld_stage = [&](const unsigned char* ptr) __attribute__((always_inline)) -> Stage {
    if constexpr (FP8) return *reinterpret_cast<const uint2*>(ptr);
    else return ld16(ptr);
  };
  auto widen = [&](const Stage& v) __attribute__((always_inline)) -> U4 {
    if constexpr (FP8) return fp8x8_to_bf16x8(v);
    else return v;
  };
  auto load_idx = [&](int t, int32_t (&dst)[LOADS]) {
    const int last = kv_end - 1;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      int tok = t * kKvTile + st_r + ROWS_PER_PASS * i;
      if (tok > last) tok = last;
      dst[i] = idx_base[tok];
    }
  };
  const unsigned char* k_rows = reinterpret_cast<const unsigned char*>(p.k_cache) + static_cast<uint64_t>(kvh) * p.fmt.head_stride + st_c * (FP8 ? 8 : 16);
  const unsigned char* v_rows = reinterpret_cast<const unsigned char*>(p.v_cache) + static_cast<uint64_t>(kvh) * p.fmt.head_stride + st_c * (FP8 ? 8 : 16);
  // V^T operand reads: lane i of 16-lane group g points at token row 4 (g >> 1) + (i >> 2), dims 16 (g & 1) + 4 (i & 3)
  const int v_lane = (4 * hi + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

  auto load_rows = [&](const unsigned char* base, const int32_t (&ids)[LOADS], Stage (&dst)[LOADS]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < LOADS; ++i) dst[i] = ld_stage(base + static_cast<uint64_t>(static_cast<uint32_t>(ids[i])) * p.fmt.page_stride);
  };
  auto put_k = [&](int buf, const Stage (&src)[LOADS]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int row = st_r + ROWS_PER_PASS * i;
      sm.k[buf][row * CPR + (st_c ^ ((row * CPR / 16) & (CPR - 1)))] = widen(src[i]);
    }
  };
  auto put_v = [&](int buf, const Stage (&src)[LOADS]) __attribute__((always_inline)) {
    unsigned char* base = reinterpret_cast<unsigned char*>(sm.v[buf]) + (st_c >> 2) * VSUB + (st_c & 3) * 16;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) *reinterpret_cast<U4*>(base + (st_r + ROWS_PER_PASS * i) * 64) = widen(src[i]);
  };
  // ---- prologue: K(0), K(1), V(0) go to LDS in one round trip; K(2) and V(1) are asked for; ids of tiles 2 / 3 held.
  // (Tiles behind the end of the walk: load_idx clamps to the walk's last row, nobody reads those images.) ----------
  {
    int32_t i0[LOADS], i1[LOADS];
    Stage k0[LOADS], k1[LOADS], v0[LOADS];
    load_idx(0, i0);
    load_idx(1, i1);
    load_rows(k_rows, i0, k0);
    load_rows(k_rows, i1, k1);
    load_rows(v_rows, i0, v0);
    load_idx(2, idx_v);
    load_idx(3, idx_k);
    put_k(0, k0);
    put_k(1, k1);
    put_v(0, v0);
    load_rows(k_rows, idx_v, kst);                   // K(2): written at the top of step 0
    load_rows(v_rows, i1, vst);                      // V(1): same
  }
  __syncthreads();

  f32x16_t s_a[2], s_b[2];                           // the two score sets: tile t and tile t + 1 trade places every step
  U4 pp[4];                                          // bf16(P(t)): [2 blk + ks] -> the 16 keys of one PV k-step
#pragma unroll
  for (int i = 0; i < 4; ++i) pp[i] = U4{0u, 0u, 0u, 0u};
#ifdef EXT_TRACE
  uint64_t tacc[6] = {0, 0, 0, 0, 0, 0};            // [commit, phase 1, mask, phase 2, decision, barrier] clocks over the walk
  uint64_t tprev = __builtin_readcyclecounter();
  const uint64_t tstart = tprev;
#endif

  auto read_k = [&](const U4* kimg, int i) __attribute__((always_inline)) {   // product i = (block i & 1, k-step i >> 1)
    const int row = (i & 1) * 32 + l31;
    return kimg[row * CPR + ((2 * (i >> 1) + hi) ^ ((row * CPR / 16) & (CPR - 1)))];
  };
  // mask (when the tile needs it), the row maximum of the lane pair, and the rescale decision for tile t: m_run, l_run
  // and O move to the new maximum together.  `links_done`: the in-lane maxima of registers 4..15 are in m4 already.
  auto mask_tile = [&](f32x16_t (&sc)[2], int t) __attribute__((always_inline)) {
    const int kv0 = t * kKvTile;
    if (__ballot(kv0 + kKvTile > row_limit) != 0ull) {                      // wave-uniform
      // the sentinel goes in BEFORE the scale: exp2(sentinel * scale - m) = 0 whatever m is
#pragma unroll
      for (int bk = 0; bk < 2; ++bk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kvpos = kv0 + 32 * bk + (r & 3) + 8 * (r >> 2) + 4 * hi;
          sc[bk][r] = kvpos < row_limit ? sc[bk][r] : kNegBig;
        }
    }
  };
  auto decide = [&](const float (&m4)[4], float psum) __attribute__((always_inline)) {
    const float mraw = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    float mx = mraw > 0.5f * kNegBig ? mraw * p.scale_log2 : kNegBig;      // a row with every key masked keeps the sentinel
    mx = lane_pair_max(mx);
    l_run += psum;
    if (__ballot(mx > m_run + kDeferMax) != 0ull) {                         // wave-uniform
      const float m_new = fmaxf(m_run, mx);
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int n = 0; n < ND; ++n) ot[n] *= alpha;
    }
  };
  constexpr int LINKS = 24;                          // in-lane maxima behind the first four: registers 4..15 of both blocks
  auto link = [&](const f32x16_t (&sc)[2], float (&m4)[4], int j) __attribute__((always_inline)) {
    const int bk = j / 12, r = 4 + j % 12;
    m4[r & 3] = fmaxf(m4[r & 3], sc[bk][r]);
  };

  // ---- S(0) and its decision stand alone; the barrier keeps step 0's write of K(2) off the image they read --------
  {
    U4 kf[3];
    kf[0] = read_k(sm.k[0], 0);
    kf[1] = read_k(sm.k[0], 1);
#pragma unroll
    for (int i = 0; i < 2 * KS; ++i) {
      if (i + 2 < 2 * KS) kf[(i + 2) % 3] = read_k(sm.k[0], i + 2);
      f32x16_t acc = s_a[i & 1];
      if (i < 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      }
      s_a[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(kf[i % 3]), as_frag(qreg[i >> 1]), acc, 0, 0, 0);
    }
    mask_tile(s_a, 0);
    float m4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) m4[c] = fmaxf(s_a[0][c], s_a[1][c]);
#pragma unroll
    for (int j = 0; j < LINKS; ++j) link(s_a, m4, j);
    decide(m4, 0.f);
  }
  __syncthreads();

  auto step = [&](int t, f32x16_t (&sc)[2], f32x16_t (&sn)[2], auto have_next_tag) __attribute__((always_inline)) {
    constexpr bool HAVE_NEXT = decltype(have_next_tag)::value;
    // ---- what was asked for a step ago goes to LDS: K(t + 2), V(t + 1) -- images nobody reads before the barrier below.
    // (Between the phases instead, with the requests beside the phase-2 products: same time per tile, the wait moves
    // with the writes; and the registers no longer fit.)
    put_k(t & 1, kst);
    put_v((t + 1) & 1, vst);
    EXT_T(0);
    // ---- phase 1 ------------------------------------------------------------------------------------------
    constexpr int KF = 3;                            // K fragment sets: reads run two products ahead (1 or 3 ahead: same time)
    constexpr int G1 = 2 * KS;                       // products (16 at D = 128)
    constexpr int SPP = 32 / G1;                     // scores exponentiated beside each (2 at D = 128, 4 at D = 64)
    const float neg_m = m_run > 0.5f * kNegBig ? -m_run : 0.f;   // rows without a key so far: exp2(sentinel * scale) = 0
    float psum = 0.f;
    {
      // the staging registers are free again: V(t + 2) rows, K(t + 3) rows, the ids of tile t + 4 -- one request per
      // RSTRIDE products (in one burst the eight waves' requests queue for the CU's one address unit)
      constexpr int NREQ = 3 * LOADS;
      constexpr int RSTRIDE = G1 / NREQ > 0 ? G1 / NREQ : 1;
      static_assert(NREQ * RSTRIDE <= G1, "requests fit the products of phase 1");
      auto request = [&](int j) __attribute__((always_inline)) {
        if (j < LOADS) {
          vst[j] = ld_stage(v_rows + static_cast<uint64_t>(static_cast<uint32_t>(idx_v[j])) * p.fmt.page_stride);
        } else if (j < 2 * LOADS) {
          if (j == LOADS) {
#pragma unroll
            for (int i = 0; i < LOADS; ++i) idx_v[i] = idx_k[i];          // the ids of K(t + 3) are those of V(t + 3) a step on
          }
          kst[j - LOADS] = ld_stage(k_rows + static_cast<uint64_t>(static_cast<uint32_t>(idx_k[j - LOADS])) * p.fmt.page_stride);
        } else {
          const int last = kv_end - 1;
          const int tok = (t + 4) * kKvTile + st_r + ROWS_PER_PASS * (j - 2 * LOADS);
          idx_k[j - 2 * LOADS] = idx_base[tok > last ? last : tok];
        }
      };
      // the exponentials as a three-stage software pipeline:  A(i) scale + subtract,  B(i) exp2,  C(i) sum + pack
      float ta[SPP], eb[SPP];
      auto stage_a = [&](int i) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < SPP; ++j) {
          const int x = SPP * i + j;                 // flat score index: block x >> 4, register x & 15
          ta[j] = fmaf(sc[x >> 4][x & 15], p.scale_log2, neg_m);
        }
      };
      auto stage_b = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < SPP; ++j) eb[j] = fast_exp2(ta[j]);
      };
      auto stage_c = [&](int i) __attribute__((always_inline)) {
        uint32_t* pw = reinterpret_cast<uint32_t*>(pp);
#pragma unroll
        for (int j = 0; j < SPP; j += 2) {
          psum += eb[j] + eb[j + 1];
          pw[(SPP * i + j) >> 1] = pack_bf2(eb[j], eb[j + 1]);   // word (x & 7) >> 1 of pp[x >> 3]
        }
      };
      const U4* kimg = sm.k[(t + 1) & 1];
      U4 kf[KF];
      if constexpr (HAVE_NEXT) {
#pragma unroll
        for (int i = 0; i < KF - 1; ++i) kf[i] = read_k(kimg, i);
      }
      stage_a(0);
#pragma unroll
      for (int i = 0; i < G1; ++i) {
        if constexpr (HAVE_NEXT) {
          if (i + KF - 1 < G1) kf[(i + KF - 1) % KF] = read_k(kimg, i + KF - 1);
          f32x16_t acc = sn[i & 1];
          if (i < 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          }
          sn[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(kf[i % KF]), as_frag(qreg[i >> 1]), acc, 0, 0, 0);
          if (i % RSTRIDE == 0 && i / RSTRIDE < NREQ) request(i / RSTRIDE);
        }
        if (i >= 1) stage_c(i - 1);
        stage_b();
        if (i + 1 < G1) stage_a(i + 1);
        if constexpr (HAVE_NEXT) __builtin_amdgcn_sched_barrier(0);   // groups stay as written
      }
      stage_c(G1 - 1);
      // P(t) and its sum are "used" here: hipcc otherwise sinks the exponential stream below later branches
      asm volatile("" : "+v"(psum));
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(pp[i].x), "+v"(pp[i].y), "+v"(pp[i].z), "+v"(pp[i].w));
    }
    EXT_T(1);
    if constexpr (HAVE_NEXT) mask_tile(sn, t + 1);
    EXT_T(2);
    // ---- phase 2 ------------------------------------------------------------------------------------------
    float m4[4];
    {
      const unsigned char* vbase = reinterpret_cast<const unsigned char*>(sm.v[t & 1]) + v_lane;
      typedef __attribute__((address_space(3))) v4s16_t* lds_v4_t;
      auto read_v = [&](int i) __attribute__((always_inline)) {   // product i = (output block i % ND, k-step i / ND of the tile)
        const unsigned char* a = vbase + (i % ND) * VSUB + (i / ND) * (16 * 64);
        const v4s16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(a));
        const v4s16_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(a + 8 * 64));
        U4 o;
        o.x = __builtin_bit_cast(uint2, lo).x; o.y = __builtin_bit_cast(uint2, lo).y;
        o.z = __builtin_bit_cast(uint2, hi4).x; o.w = __builtin_bit_cast(uint2, hi4).y;
        return o;
      };
      constexpr int LSTART = NPV / 4;                // the first products run while the last S^T products still finish
      constexpr int PER = LINKS / (NPV - LSTART);    // maxima per product behind them (2 at D = 128, 4 at D = 64)
      static_assert(PER * (NPV - LSTART) == LINKS, "maxima spread evenly");
      U4 vf[3];
      vf[0] = read_v(0);
      vf[1] = read_v(1);
#pragma unroll
      for (int i = 0; i < NPV; ++i) {
        if (i + 2 < NPV) vf[(i + 2) % 3] = read_v(i + 2);
        ot[i % ND] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(vf[i % 3]), as_frag(pp[i / ND]), ot[i % ND], 0, 0, 0);
        if constexpr (HAVE_NEXT) {
          if (i == LSTART) {
#pragma unroll
            for (int c = 0; c < 4; ++c) m4[c] = fmaxf(sn[0][c], sn[1][c]);
          }
          if (i >= LSTART) {
#pragma unroll
            for (int j = PER * (i - LSTART); j < PER * (i - LSTART + 1); ++j) link(sn, m4, j);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    EXT_T(3);
    // ---- the decision for tile t + 1: P(t) is inside O and l, nothing at the new maximum exists yet ----------
    if constexpr (HAVE_NEXT) decide(m4, psum);
    else l_run += psum;
    EXT_T(4);
    __syncthreads();
    EXT_T(5);
  };
  {
    int t = 0;
    for (; t + 2 < n_tiles; t += 2) {
      step(t, s_a, s_b, std::true_type{});
      step(t + 1, s_b, s_a, std::true_type{});
    }
    if (n_tiles - t == 2) {
      step(t, s_a, s_b, std::true_type{});
      step(t + 1, s_b, s_a, std::false_type{});
    } else {
      step(t, s_a, s_b, std::false_type{});
    }
  }
#ifdef EXT_TRACE
  if (g_ext_trace && lane == 0) {
    uint64_t* o = g_ext_trace + static_cast<int64_t>(blockIdx.x) * 64 + wid * 8;
    for (int i = 0; i < 6; ++i) o[i] = tacc[i];
    o[6] = n_tiles;
    o[7] = tstart;
  }
#endif

  // ---- epilogue: lane (q, h) holds O^T[d = 32 n + 8 c + 4 h + e][q], c = 0..3, e = 0..3.  The lane halves trade
  // 4-dim pieces (v_permlane32_swap on the packed words) so that h = 0 owns the 8 dims of c = 0, 2 and h = 1 those of
  // c = 1, 3: eight 16-byte stores per lane instead of sixteen 8-byte ones ------------------------------------
  float l = l_run;
  l += __shfl_xor(l, 32, 64);
  const float inv = (l > 0.f) ? p.v_scale / l : 0.f;
  uint16_t* op = p.out + static_cast<int64_t>(q_begin + q0 + r_tok) * p.out_stride + row_off;
#pragma unroll
  for (int n = 0; n < ND; ++n) {
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {                  // the piece pair (c = 2 cp, c = 2 cp + 1)
      uint32_t a0 = pack_bf2(ot[n][8 * cp + 0] * inv, ot[n][8 * cp + 1] * inv);
      uint32_t a1 = pack_bf2(ot[n][8 * cp + 2] * inv, ot[n][8 * cp + 3] * inv);
      uint32_t b0 = pack_bf2(ot[n][8 * cp + 4] * inv, ot[n][8 * cp + 5] * inv);
      uint32_t b1 = pack_bf2(ot[n][8 * cp + 6] * inv, ot[n][8 * cp + 7] * inv);
      // swap: upper half of a <-> lower half of b.  Afterwards a = dims 0-3, b = dims 4-7 of the lane's own piece
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a0), "+v"(b0));
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a1), "+v"(b1));
      if (row_ok) {
        U4 w;
        w.x = a0; w.y = a1; w.z = b0; w.w = b1;
        *reinterpret_cast<U4*>(op + n * 32 + (2 * cp + hi) * 8) = w;
      }
    }
  }
}

// Test-only probe: C[16x16] = A[16x32] . B[32x16] with the operand/result lane
// maps this file assumes (A row = lane&15, k = 8*(lane>>4)+j; C row = 4*(lane>>4)+r,
// col = lane&15).  tests/ checks it against a plain matmul.
__global__ void mfma_probe_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ bm,
                                  float* __restrict__ c) {
  const int lane = threadIdx.x;
  const int l15 = lane & 15, g = lane >> 4;
  U4 af, bf;
  uint16_t* ap = reinterpret_cast<uint16_t*>(&af);
  uint16_t* bp = reinterpret_cast<uint16_t*>(&bf);
  for (int j = 0; j < 8; ++j) {
    ap[j] = a[l15 * 32 + g * 8 + j];          // A[i = l15][k = 8g + j]
    bp[j] = bm[(g * 8 + j) * 16 + l15];        // B[k = 8g + j][n = l15]
  }
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af), as_frag(bf), acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) c[(g * 4 + r) * 16 + l15] = acc[r];
}

}  // namespace

// test / tuning overrides (process-wide, set through sgl_amd_debug_extend_attention_shape; never read from the environment)
static int g_extend_debug_shape = 0;      // 0: automatic; 41 / 42 / 82: waves x M-tiles per wave
static int g_extend_debug_flags = 0;      // bit 0: bf16 8-wave launches stay on the general single-image kernel; bit 1: they take the ping-pong kernel

extern "C" {

int sgl_amd_debug_extend_attention_shape(int shape, int flags) {
  if (shape != 0 && shape != 41 && shape != 42 && shape != 82) {
    set_last_error("debug_extend_attention_shape: shape must be 0 (automatic), 41, 42 or 82");
    return -1;
  }
  g_extend_debug_shape = shape;
  g_extend_debug_flags = flags;
  return 0;
}

int sgl_amd_extend_attention(const void* q, void* out, const void* k_cache, const void* v_cache,
                             const int32_t* req_to_token, int64_t req_to_token_stride,
                             const int64_t* req_pool_indices, const int32_t* seq_lens,
                             const int32_t* prefix_lens, const int32_t* qo_indptr, int64_t batch,
                             int max_extend_len, int num_q_heads, int num_kv_heads, int head_dim,
                             int64_t q_token_stride, int64_t out_token_stride,
                             int64_t k_cache_row_stride, int64_t v_cache_row_stride, float sm_scale,
                             int causal, void* stream) {
  return sgl_amd_extend_attention_ex(q, out, k_cache, v_cache, req_to_token, req_to_token_stride, req_pool_indices, seq_lens,
                                     prefix_lens, qo_indptr, batch, max_extend_len, num_q_heads, num_kv_heads, head_dim,
                                     q_token_stride, out_token_stride, k_cache_row_stride, v_cache_row_stride, sm_scale, causal,
                                     0, 1.0f, 1.0f, 1, 0, -1, 0.0f, nullptr, nullptr, 0, stream);
}

int sgl_amd_extend_attention_ex(const void* q, void* out, const void* k_cache, const void* v_cache,
                                const int32_t* req_to_token, int64_t req_to_token_stride,
                                const int64_t* req_pool_indices, const int32_t* seq_lens,
                                const int32_t* prefix_lens, const int32_t* qo_indptr, int64_t batch,
                                int max_extend_len, int num_q_heads, int num_kv_heads, int head_dim,
                                int64_t q_token_stride, int64_t out_token_stride,
                                int64_t k_cache_row_stride, int64_t v_cache_row_stride, float sm_scale,
                                int causal, int kv_fp8, float k_scale, float v_scale, int page_size, int kv_layout_hnd,
                                int sliding_window, float logit_cap, const void* custom_mask, const int64_t* mask_indptr,
                                int skip_prefix_custom_mask, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(k_cache_row_stride == v_cache_row_stride, "extend_attention: K and V pools must share a row stride");
  SGL_CHECK_ARG(!kv_fp8 || (k_scale > 0.f && v_scale > 0.f), "extend_attention: fp8 KV needs positive k_scale / v_scale");
  SGL_CHECK_ARG(logit_cap >= 0.f, "extend_attention: logit_cap must be >= 0 (0 = off)");
  SGL_CHECK_ARG((custom_mask == nullptr) == (mask_indptr == nullptr), "extend_attention: custom_mask and mask_indptr come together");
  SGL_CHECK_ARG(head_dim == 64 || head_dim == 128, "extend_attention: head_dim=%d not supported (64/128)", head_dim);
  SGL_CHECK_ARG(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0,
                "extend_attention: num_q_heads=%d not a multiple of num_kv_heads=%d", num_q_heads, num_kv_heads);
  const int group = num_q_heads / num_kv_heads;
  SGL_CHECK_ARG(group <= kRows, "extend_attention: GQA group %d too large", group);
  SGL_CHECK_ARG(q_token_stride % 8 == 0 && out_token_stride % 4 == 0 && k_cache_row_stride % 8 == 0 && v_cache_row_stride % 8 == 0,
                "extend_attention: strides must keep 16-byte row alignment");
  SGL_CHECK_ARG(batch <= 65535 && num_kv_heads <= 65535, "extend_attention: grid too large");
  if (batch == 0 || max_extend_len <= 0) return 0;
  ExtendParams p;
  p.q = static_cast<const uint16_t*>(q);
  p.out = static_cast<uint16_t*>(out);
  p.k_cache = static_cast<const uint16_t*>(k_cache);
  p.v_cache = static_cast<const uint16_t*>(v_cache);
  p.req_to_token = req_to_token;
  p.req_pool_indices = req_pool_indices;
  p.seq_lens = seq_lens;
  p.prefix_lens = prefix_lens;
  p.qo_indptr = qo_indptr;
  p.q_stride = q_token_stride;
  p.out_stride = out_token_stride;
  p.kc_stride = k_cache_row_stride;
  p.vc_stride = v_cache_row_stride;
  p.r2t_stride = req_to_token_stride;
  p.num_kv_heads = num_kv_heads;
  p.group = group;
  // Workgroup shape (waves x M-tiles per wave -> rows that share one staged K/V tile):
  //   8 x 2 = 256 rows, two waves per SIMD and the whole register file: fewest K/V gathers per flop; measured
  //                    fastest on MI355X for every shape that gives each CU a workgroup (benchmarks/micro.py);
  //   4 x 1 =  64 rows, 3 workgroups / CU: short extends whose wider grids would leave CUs idle;
  //   4 x 2 = 128 rows: what is left (GQA groups above 64 on a small grid).
  const int64_t rows_total = static_cast<int64_t>(max_extend_len) * group;
  auto wgs = [&](int rows) { return ((rows_total + rows - 1) / rows) * num_kv_heads * batch; };
  int mtw = 2, nwv = 4;
  if (group <= 256 && wgs(256) >= 256) { mtw = 2; nwv = 8; }
  else if (group <= 64) { mtw = 1; nwv = 4; }
  if (const int v = g_extend_debug_shape) {                   // sgl_amd_debug_extend_attention_shape(): tests / tuning
    if (v == 41 && group <= 64) { nwv = 4; mtw = 1; }
    if (v == 42) { nwv = 4; mtw = 2; }
    if (v == 82) { nwv = 8; mtw = 2; }
  }
  p.tokens_per_tile = (nwv * 16 * mtw) / group;
  p.causal = causal;
  p.scale_log2 = sm_scale * 1.4426950408889634f * (kv_fp8 ? k_scale : 1.0f);
  p.v_scale = kv_fp8 ? v_scale : 1.0f;
  p.window = sliding_window;
  p.cap_log2 = logit_cap > 0.f ? logit_cap * 1.4426950408889634f : 0.f;
  p.inv_cap_log2 = logit_cap > 0.f ? 1.0f / p.cap_log2 : 0.f;
  p.custom_mask = static_cast<const uint8_t*>(custom_mask);
  p.mask_indptr = mask_indptr;
  p.mask_skip_prefix = skip_prefix_custom_mask ? 1 : 0;
  SGL_CHECK_ARG(make_kv_format(&p.fmt, k_cache_row_stride, num_kv_heads, head_dim, page_size, kv_layout_hnd, kv_fp8),
                "extend_attention: HND pools need a power-of-two page_size (got %d); row / page strides must stay below 4 GiB", page_size);
  const int tiles = (max_extend_len + p.tokens_per_tile - 1) / p.tokens_per_tile;
  dim3 grid(tiles, num_kv_heads, batch);
  hipStream_t st = as_stream(stream);
  // token-major pools (bf16 or e4m3 rows), 8-wave shape: the 32x32 two-score-set kernel (debug flag bit 1: the bf16
  // ping-pong kernel it replaced)
  const bool fast = !kv_layout_hnd && nwv == 8 && sliding_window < 0 && logit_cap == 0.f && custom_mask == nullptr &&
                    (g_extend_debug_flags & 1) == 0 && !(kv_fp8 && (g_extend_debug_flags & 2) != 0);
  if (fast) {
    p.num_tiles = tiles; p.batch = static_cast<int>(batch);
    const dim3 grid1(static_cast<unsigned>(tiles) * num_kv_heads * static_cast<unsigned>(batch));
    if ((g_extend_debug_flags & 2) != 0) {
      if (head_dim == 128) hipLaunchKernelGGL((extend_attention_dbuf_kernel<128>), grid1, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((extend_attention_dbuf_kernel<64>), grid1, dim3(512), 0, st, p);
    } else if (kv_fp8) {
      if (head_dim == 128) hipLaunchKernelGGL((extend_attention_pipe_kernel<128, 8, true>), grid1, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((extend_attention_pipe_kernel<64, 8, true>), grid1, dim3(512), 0, st, p);
    } else if (head_dim == 128) hipLaunchKernelGGL((extend_attention_pipe_kernel<128, 8, false>), grid1, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((extend_attention_pipe_kernel<64, 8, false>), grid1, dim3(512), 0, st, p);
    SGL_CHECK_LAUNCH("extend_attention");
    return 0;
  }
#define SGL_LAUNCH_EXT(D_, M_, W_)                                                                                        \
  do {                                                                                                                    \
    if (kv_fp8) hipLaunchKernelGGL((extend_attention_kernel<D_, M_, W_, true>), grid, dim3(64 * W_), 0, st, p);           \
    else hipLaunchKernelGGL((extend_attention_kernel<D_, M_, W_, false>), grid, dim3(64 * W_), 0, st, p);                 \
  } while (0)
  if (head_dim == 128) {
    if (nwv == 8) SGL_LAUNCH_EXT(128, 2, 8);
    else if (mtw == 1) SGL_LAUNCH_EXT(128, 1, 4);
    else SGL_LAUNCH_EXT(128, 2, 4);
  } else {
    if (nwv == 8) SGL_LAUNCH_EXT(64, 2, 8);
    else if (mtw == 1) SGL_LAUNCH_EXT(64, 1, 4);
    else SGL_LAUNCH_EXT(64, 2, 4);
  }
#undef SGL_LAUNCH_EXT
  SGL_CHECK_LAUNCH("extend_attention");
  return 0;
}

#ifdef EXT_TRACE
int sgl_amd_debug_ext_trace(void* buf) {
  uint64_t* p = static_cast<uint64_t*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_ext_trace), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

int sgl_amd_probe_mfma_16x16x32(const void* a, const void* b, void* c, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, as_stream(stream),
                     static_cast<const uint16_t*>(a), static_cast<const uint16_t*>(b),
                     static_cast<float*>(c));
  SGL_CHECK_LAUNCH("probe_mfma_16x16x32");
  return 0;
}

}  // extern "C"

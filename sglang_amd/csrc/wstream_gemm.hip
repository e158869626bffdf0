// Weight-streaming GEMM for decode batches on gfx950:  y[M, N] = x[M, K] . W[N, K]^T,  M <= 64.
//
// Replaces, for decode-sized M, the library matmul behind the dense projections of the
// decoder layer (reference, /root/reference/python/sglang):
//   srt/layers/linear.py:1596-1660   UnquantizedLinearMethod.apply  (qkv / o / gate_up / down / lm_head)
// and, through the fused epilogues of the split-K combine kernel,
//   srt/layers/activation.py:141-143 SiluAndMul.forward_native           (after gate_up)
//   srt/layers/layernorm.py:786-820  RMSNorm.forward_native with residual (after o_proj / down_proj)
// Oracle: torch F.linear + oracle/ops.py.
//
// Every weight byte is used exactly once per decode step, so the kernel is an HBM stream of W
// with the matrix cores riding along.  Differences from skinny_gemm.hip (which stays for the
// grouped / ragged-K cases):
//   * a wave owns ONE 16-row tile of W and a K range; its weight rows go straight from HBM into
//     the MFMA A-operand registers (non-temporal 16 B loads, 64 contiguous bytes per row per
//     instruction), kPD K-chunks (kPD KiB per lane group) in flight per wave at all times -- the
//     activation chunk of the same K range travels in the same register ring, so the in-order
//     vmcnt never drains the weight prefetch;
//   * the activation rows are shared by the 4-8 waves of a workgroup through a double-buffered,
//     XOR-swizzled LDS image (one barrier per 128-wide K chunk);
//   * the (waves per workgroup, K splits) pair is chosen by the host per shape so that
//     tiles x splits fills the 256 CUs evenly (N=28672: 7 waves x 2 splits = 14 waves on every CU);
//   * split-K partials are plain row-major fp32 [split][M][N]; one combine kernel sums them in
//     split order (deterministic) and applies the epilogue the next operator would have been:
//     bias, silu(gate)*up, or residual-add + RMSNorm, with the torch-native bf16 rounding points.
#include "common.hpp"
#include "kv_format.hpp"
#include "sglang_amd.h"

using namespace sgl_amd;

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int kKC = 128;              // K elements per chunk (256 B of every weight row)
constexpr int kKSteps = kKC / 32;     // MFMA k-steps per chunk
constexpr int kSlotsPerRow = kKC / 8; // 16-byte slots per staged activation row

__device__ __forceinline__ bf16x8_t as_frag(const U4& v) { return __builtin_bit_cast(bf16x8_t, v); }

// streamed-once data: keep it out of the way of the L2-resident activations
__device__ __forceinline__ U4 ld16_stream(const uint16_t* p) {
  u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return __builtin_bit_cast(U4, v);
}

// Activations come row-major [M, K] or "chunk-major" [K/128][M][128] (every 128-wide K chunk of all rows contiguous:
// the 16 KiB a workgroup stages per step is then one burst over all L2 channels instead of 64 pieces 2K bytes
// apart).  Both are (row stride, chunk stride) pairs; element (m, n) lives at (n / 128) * cstride + m * stride + n % 128.
__device__ __forceinline__ int64_t blocked_off(int64_t m, int n, int64_t stride, int64_t cstride) {
  return static_cast<int64_t>(n >> 7) * cstride + m * stride + (n & 127);
}

// Sum of the split partials in split order (deterministic).  The loads of four consecutive splits are issued
// together (clamped to the last split, so they need no predicate); only the adds are predicated.
__device__ __forceinline__ f32x4_t sum_splits(const float* p0, int64_t split_stride, int splits) {
  f32x4_t s = *reinterpret_cast<const f32x4_t*>(p0);
  for (int sp = 1; sp < splits; sp += 4) {
    f32x4_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = sp + j < splits ? sp + j : splits - 1;
      v[j] = *reinterpret_cast<const f32x4_t*>(p0 + q * split_stride);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (sp + j < splits) s += v[j];
  }
  return s;
}

// ---------------------------------------------------------------------------------------------
// QKV combine: split-K sum + bias -> bf16 (the qkv_proj output), neox rotary embedding on the q and k
// heads with the torch-native rounding points (rotary_embedding/utils.py:49-57: cos/sin in bf16,
// o1 = bf16(bf16(x1 c) - bf16(x2 s)), o2 = bf16(bf16(x2 c) + bf16(x1 s))), q written to q_out, the
// rotated k row and the v row written straight into the token->KV pool at cache_loc
// (base.py:385-417 fused_set_kv_buffer).  grid (token, item block): one 4-wide item per thread, so the
// kernel is two dependent loads deep whatever the head count.
// ---------------------------------------------------------------------------------------------
struct RopeParams {
  const float* part;            // [splits, M, N]
  const uint16_t* bias;         // optional [N]
  uint16_t* q_out;              // [M, Hq * D]
  uint16_t* k_cache;            // [slots, Hkv * D]
  uint16_t* v_cache;
  const int64_t* positions;     // [M]
  const int64_t* cache_loc;     // [M]
  const void* cos_sin;          // [max_pos, D]  cos | sin halves, bf16 or fp32
  int64_t q_stride, cache_row_stride;
  int M, N, splits, num_q_heads, num_kv_heads, head_dim, cache_f32;
  // pools other than bf16 token-major (memory_pool.py:2061-2117 HND pages, :2364-2374 fp8 rows of x / scale)
  int generic;
  KvFormat fmt;
  float inv_k_scale, inv_v_scale;
};

// four values of a KV row to the pool: bf16 (8 bytes) or e4m3 of x / scale (4 bytes, the bf16-rounded value is what
// the reference's set_kv_buffer quantises)
__device__ __forceinline__ void put_kv4(unsigned char* row, int elem, const float (&v)[4], bool fp8, float inv_scale) {
  if (fp8) {
    float c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = fminf(fmaxf(rbf(v[j]) * inv_scale, -448.f), 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w, true);
    *reinterpret_cast<uint32_t*>(row + elem) = static_cast<uint32_t>(w);
  } else {
    uint2 w2;
    w2.x = pack_bf2(v[0], v[1]); w2.y = pack_bf2(v[2], v[3]);
    *reinterpret_cast<uint2*>(row + elem * 2) = w2;
  }
}

// one 4-wide neox pair item of a q / k head: columns h * D + i .. + 4 and their partners half a head on
__device__ __forceinline__ void rope_pair_item(const RopeParams& p, int m, int h, int i) {
  const int D = p.head_dim, half = D >> 1;
  const int64_t ss = static_cast<int64_t>(p.M) * p.N;
  const float* row = p.part + static_cast<int64_t>(m) * p.N;
  const int64_t pos = p.positions[m];
  const int64_t slot = p.cache_loc[m];
  const int n1 = h * D + i, n2 = n1 + half;
  f32x4_t a = sum_splits(row + n1, ss, p.splits);
  f32x4_t b = sum_splits(row + n2, ss, p.splits);
  float c[4], sn[4];
  if (p.cache_f32) {
    const float* cs = static_cast<const float*>(p.cos_sin) + pos * D;
#pragma unroll
    for (int r = 0; r < 4; ++r) { c[r] = rbf(cs[i + r]); sn[r] = rbf(cs[half + i + r]); }
  } else {
    const uint16_t* cs = static_cast<const uint16_t*>(p.cos_sin) + pos * D;
#pragma unroll
    for (int r = 0; r < 4; ++r) { c[r] = bf2f(cs[i + r]); sn[r] = bf2f(cs[half + i + r]); }
  }
  float o1[4], o2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x1 = a[r], x2 = b[r];
    if (p.bias) { x1 += bf2f(p.bias[n1 + r]); x2 += bf2f(p.bias[n2 + r]); }
    x1 = rbf(x1); x2 = rbf(x2);
    o1[r] = rbf(x1 * c[r]) - rbf(x2 * sn[r]);
    o2[r] = rbf(x2 * c[r]) + rbf(x1 * sn[r]);
  }
  if (p.generic && h >= p.num_q_heads) {
    unsigned char* kr = const_cast<unsigned char*>(kv_row(p.k_cache, p.fmt, static_cast<int>(slot), h - p.num_q_heads));
    put_kv4(kr, i, o1, p.fmt.fp8 != 0, p.inv_k_scale);
    put_kv4(kr, i + half, o2, p.fmt.fp8 != 0, p.inv_k_scale);
    return;
  }
  uint16_t* dst = h < p.num_q_heads ? p.q_out + static_cast<int64_t>(m) * p.q_stride + n1
                                    : p.k_cache + slot * p.cache_row_stride + (h - p.num_q_heads) * D + i;
  uint2 w1, w2;
  w1.x = pack_bf2(o1[0], o1[1]); w1.y = pack_bf2(o1[2], o1[3]);
  w2.x = pack_bf2(o2[0], o2[1]); w2.y = pack_bf2(o2[2], o2[3]);
  *reinterpret_cast<uint2*>(dst) = w1;
  *reinterpret_cast<uint2*>(dst + half) = w2;
}

// one 4-wide item of the v row: element e .. e + 4 of the token's [Hkv * D] values
__device__ __forceinline__ void rope_v_item(const RopeParams& p, int m, int e) {
  const int D = p.head_dim;
  const int64_t ss = static_cast<int64_t>(p.M) * p.N;
  const int v0 = (p.num_q_heads + p.num_kv_heads) * D;
  const int64_t slot = p.cache_loc[m];
  f32x4_t a = sum_splits(p.part + static_cast<int64_t>(m) * p.N + v0 + e, ss, p.splits);
  if (p.bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] += bf2f(p.bias[v0 + e + r]);
  }
  if (p.generic) {
    const int vh = e / D;
    const float av[4] = {a[0], a[1], a[2], a[3]};
    put_kv4(const_cast<unsigned char*>(kv_row(p.v_cache, p.fmt, static_cast<int>(slot), vh)), e - vh * D, av, p.fmt.fp8 != 0,
            p.inv_v_scale);
    return;
  }
  uint2 w;
  w.x = pack_bf2(a[0], a[1]); w.y = pack_bf2(a[2], a[3]);
  *reinterpret_cast<uint2*>(p.v_cache + slot * p.cache_row_stride + e) = w;
}

struct WsParams {
  const uint16_t* x;     // [M, K]
  const uint16_t* w;     // [N, K]
  const uint16_t* bias;  // optional [N] (only applied when splits == 1)
  uint16_t* y;           // [M, N] bf16 (splits == 1)
  float* part;           // [splits, M, N] fp32, or NULL: write y directly
  int64_t x_stride, w_stride, y_stride;
  int64_t x_cstride;     // elements between consecutive 128-wide K chunks of an x row (128 for a row-major x)
  int64_t y_cstride;     // the same for the 128-wide column blocks of y (bf16 outputs)
  int M, N, K, splits, ntiles;
  // grouped (mixture-of-experts) form: blockIdx.y = row block of moe_align_block_size's output, rows gathered /
  // scattered through sorted_ids, weights of expert expert_ids[block]; M = number of valid (token, k) pair ids
  const int32_t* sorted_ids;    // [>= blocks * 16 MT]  pair ids, padding >= M
  const int32_t* expert_ids;    // [blocks], < 0: filtered expert
  const int32_t* num_post_pad;  // [1]
  const float* row_scale;       // optional [M] router weights
  int64_t w_expert_stride;
  int topk_div;                 // source row of pair id = id / topk_div
  int out_f32;                  // y is fp32 (the down projection ahead of moe_sum_reduce)
  int round_before_scale;       // round the accumulator to bf16 before the router weight (fused_moe_native.py:157-163)
  int dbg;                      // sgl_amd_debug_wstream_flags(): bit 1 = plain (write-back) partial stores; forced for partial buffers of
                                // 2 GiB and more (32-bit buffer offsets)
  int pair_silu;                // 1: two-tile waves, y[:, 16 t ..] = silu(tile t) * tile (t + ntiles/2) instead of two outputs;
                                // 2: one-tile waves whose 16 weight rows are [8 gate rows | 8 up rows] of output columns 8 t .. 8 t + 7
};

// ---- LDS-DMA plumbing ------------------------------------------------------------------------
// The weight and activation chunks go HBM/L2 -> LDS with global_load_lds_dwordx4 (1 KiB per wave
// instruction, lane i lands at dst + 16 i, no VGPR round trip).  hipcc orders every LDS read it can
// see behind ALL pending LDS-DMA (s_waitcnt vmcnt(0)), which would collapse the prefetch ring, so
// the fragment reads are inline-asm ds_read_b128 and the two counters are waited on by hand.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) unsigned char* lds_bytes_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

#ifdef WS_TRACE
// timeline probe (benchmarks/r02_exp5_ws_trace.py): per workgroup, 100 MHz chip-wide clock at entry, first chunk
// landed, last chunk multiplied, epilogue stored
__device__ uint64_t* g_ws_trace = nullptr;
#define WS_STAMP(i)                                                                                   \
  do {                                                                                                \
    if (g_ws_trace && threadIdx.x == 0)                                                               \
      g_ws_trace[(static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 4 + (i)] = wall_clock64(); \
  } while (0)
#else
#define WS_STAMP(i)
#endif

template <int AUX>
__device__ __forceinline__ void dma16(const uint16_t* src, lds_ptr_t dst) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)(src), dst, 16, 0, AUX);
}

template <int OFF>
__device__ __forceinline__ u32x4_t lds_rd(uint32_t addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// s_waitcnt lgkmcnt(N), then pin every fragment behind it: volatile asm statements keep their order, and a
// consumer of `v` depends on the empty asm that "rewrites" it, so it cannot be scheduled above the wait.
template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N)); }
__device__ __forceinline__ void pin(u32x4_t& v) { asm volatile("" : "+v"(v)); }

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ring depth: as many chunk slots as fit the 160 KiB LDS (1 KiB is the dummy landing zone), at most 6
constexpr int ring_depth(int mt, int nw, int tpw) {
  const int slot = nw * tpw * 4096 + mt * 4 * 1024;
  const int d = (160 * 1024 - 1024) / slot;
  return d > 6 ? 6 : d;
}

// TPW = 16-row weight tiles per wave.  TPW == 2: the wave owns tiles t and t + ntiles/2 (half the activation reads
// per weight byte); with pair_silu == 1 they are a gate tile and its up tile and the wave writes
// y[m, 16 t ..] = silu(gate) * up (no partials, no second launch).
// pair_silu == 2 (TPW == 1, round 4): a wave's ONE tile is made of 8 gate rows and the 8 up rows of the same output
// columns, so any number of waves per workgroup keeps gate and up together: N / 16 tiles can then be dealt out in whole
// rounds of 256 workgroups where N / 32 pairs cannot (Llama-3-8B gate_up: 1792 tiles = 256 x 7 waves, against 896 pairs =
// 224 x 4 waves: 32 CUs idle).  Gate and up of a column sit 32 lanes apart in the accumulator: one lane swap in the epilogue.
template <int MT, int NW, int TPW, bool GROUPED>
__global__ __launch_bounds__(64 * NW, 1) void wstream_gemm_kernel(WsParams p) {
  constexpr int PD = ring_depth(MT, NW, TPW);
  constexpr int XPIECES = MT * 4;                       // 1 KiB pieces (4 rows x 256 B) of an activation chunk
  constexpr int XP = (XPIECES + NW - 1) / NW;           // pieces each wave fetches (surplus -> dummy slot)
  constexpr int WPC = TPW * kKSteps;                    // weight DMA instructions per wave per chunk
  constexpr int WWAVE = TPW * 4096;                     // weight bytes per wave per chunk
  constexpr int WCH = NW * WWAVE;                       // weight bytes per chunk
  constexpr int CH = WCH + XPIECES * 1024;              // ring slot: [weights of wave 0..NW) | activations]
  static_assert(PD >= 3 && PD * CH + 1024 <= 160 * 1024, "LDS ring does not fit");
  static_assert((PD - 1) * WPC + (PD - 2) * XP < 64, "vmcnt range");
  __shared__ __attribute__((aligned(1024))) unsigned char ring[PD * CH + 1024];

  const int tid = threadIdx.x, lane = tid & 63;
  WS_STAMP(0);
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int wtiles = p.ntiles / TPW;                    // tiles a wave index ranges over
  const int bx = blockIdx.x, by = GROUPED ? 0 : blockIdx.y;
  const int tile = bx * NW + wid;
  const bool active = tile < wtiles;
  const int split = by;
  const int rb = GROUPED ? blockIdx.y : 0;              // row block (grouped form)
  const uint16_t* wbase = p.w;
  if constexpr (GROUPED) {
    if (rb * (16 * MT) >= p.num_post_pad[0]) return;
    const int e = p.expert_ids[rb];
    if (e < 0) return;                                  // filtered expert (EP): its rows stay untouched
    wbase += static_cast<int64_t>(e) * p.w_expert_stride;
  }
  const int nch = p.K / kKC;
  const int cb = static_cast<int>(static_cast<int64_t>(split) * nch / p.splits);
  const int ce = static_cast<int>(static_cast<int64_t>(split + 1) * nch / p.splits);
  const int n = ce - cb;

  // ---- DMA roles.  A 1 KiB piece = 4 rows x 16 slots; the lane at (row q, slot s) of the LDS image
  // fetches global slot s ^ (row & 15): the XOR swizzle that makes the fragment reads conflict-free
  // is applied on the global side, each row's 256 B still move as two whole cache lines.
  const int q4 = lane >> 4, s16 = lane & 15;
  const uint16_t* wsrc[TPW][kKSteps];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int j = 0; j < kKSteps; ++j) {
      const int row = 4 * j + q4;
      const int64_t tl = active ? tile : wtiles - 1;
      const int64_t wrow = (TPW == 1 && p.pair_silu == 2) ? (row < 8 ? 0 : p.N / 2) + tl * 8 + (row & 7)
                                                          : (tl + static_cast<int64_t>(t) * wtiles) * 16 + row;
      wsrc[t][j] = wbase + wrow * p.w_stride + ((s16 ^ row) & 15) * 8 + static_cast<int64_t>(cb) * kKC;
    }
  const uint16_t* xsrc[XP];
  int xoff[XP];                                          // wave-uniform LDS offset inside a ring slot
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int piece = wid + NW * i;
    const bool real = piece < XPIECES;
    const int row = real ? 4 * piece + q4 : 0;
    int srow;
    if constexpr (GROUPED) {
      const int id = p.sorted_ids[rb * (16 * MT) + row];
      srow = id < p.M ? id / p.topk_div : 0;            // padding rows read row 0: their columns are discarded
    } else {
      srow = row < p.M ? row : p.M - 1;
    }
    xsrc[i] = p.x + static_cast<int64_t>(srow) * p.x_stride + ((s16 ^ row) & 15) * 8 + static_cast<int64_t>(cb) * p.x_cstride;
    xoff[i] = real ? WCH + piece * 1024 : -1;
  }
  lds_bytes_t ring3 = (lds_bytes_t)(ring);
  // Chunk c lives in ring slot c % PD.  A wave's weight tiles of a slot are private to it: the wave copies them to
  // registers (TPW * 4 fragments) at the top of compute() and refills the slot with chunk c + PD at once, so all PD
  // slots' worth of weights stay in flight; the activation image is shared, so chunk c + PD - 1's goes into slot
  // c - 1 after the barrier that retires it.  Behind the end of the K range a dummy DMA (L2-resident source, the
  // 1 KiB landing zone) is issued instead, which keeps the in-order vmcnt arithmetic the same in every iteration.
  const uint16_t* dummy_src = p.x + s16 * 8;
  lds_ptr_t dummy_dst = (lds_ptr_t)(ring3 + PD * CH);
  // Without split-K the workgroups walk the K range from staggered starting chunks (wrapping around): launched
  // together they would otherwise all pull the same 256-byte column of their rows at the same moment, which the row
  // stride (a multiple of 8 KiB) maps onto a few memory channels (gate_up 46.5 -> 38.6 us).  Split-K launches already
  // start at `splits` different columns and measured 0.3-1 us slower staggered.
  const int rot = p.splits == 1 && n > 1 ? static_cast<int>((static_cast<int64_t>(bx) * n) / gridDim.x) : 0;
  auto issue_w = [&](int c_rel, int slot) {
    if (c_rel < n) {
      const int cr = c_rel + rot < n ? c_rel + rot : c_rel + rot - n;
      const int koff = cr * kKC;
      const int base = slot * CH + wid * WWAVE;
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int j = 0; j < kKSteps; ++j) dma16<2>(wsrc[t][j] + koff, (lds_ptr_t)(ring3 + base + t * 4096 + j * 1024));
    } else {
#pragma unroll
      for (int i = 0; i < WPC; ++i) dma16<0>(dummy_src, dummy_dst);
    }
  };
  auto issue_x = [&](int c_rel, int slot) {
    if (c_rel < n) {
      const int cr = c_rel + rot < n ? c_rel + rot : c_rel + rot - n;
      const int64_t koff = cr * p.x_cstride;
      const int base = slot * CH;
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        const int off = xoff[i] >= 0 ? base + xoff[i] : PD * CH;
        dma16<0>(xsrc[i] + koff, (lds_ptr_t)(ring3 + off));
      }
    } else {
#pragma unroll
      for (int i = 0; i < XP; ++i) dma16<0>(dummy_src, dummy_dst);
    }
  };

  // ---- fragment read addresses (bytes, LDS address space) ----
  const uint32_t ring_addr = (uint32_t)(uintptr_t)(ring3);
  uint32_t foff[kKSteps];
#pragma unroll
  for (int kk = 0; kk < kKSteps; ++kk) foff[kk] = r16 * 256 + (((kk * 4 + g) ^ r16) & 15) * 16;

  f32x4_t acc[TPW][MT];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // activation fragments of one "stage": up to 64 rows the stage is a whole k-step (all B tiles); beyond, a k-step is
  // two stages (first half of the B tiles, then the second half) so that the reads of two consecutive stages stay
  // inside the 4-bit lgkmcnt while one stage is always read ahead of the MFMAs
  constexpr bool WIDE = MT > 4;
  constexpr int H = WIDE ? MT / 2 : MT;                // B tiles per stage
  constexpr int NST = WIDE ? 2 * kKSteps : kKSteps;    // stages per chunk
  auto read_stage = [&](uint32_t slot_addr, int st, u32x4_t (&b)[H]) {
    const int kk = WIDE ? st >> 1 : st;
    const int half = WIDE ? st & 1 : 0;
    const uint32_t bd = slot_addr + foff[kk] + half * (H * 4096);
    b[0] = lds_rd<WCH>(bd);
    if constexpr (H > 1) b[1] = lds_rd<WCH + 4096>(bd);
    if constexpr (H > 2) b[2] = lds_rd<WCH + 8192>(bd);
    if constexpr (H > 3) b[3] = lds_rd<WCH + 12288>(bd);
  };
  auto compute = [&](int slot, int c) {
    const uint32_t slot_addr = ring_addr + slot * CH;
    u32x4_t a[kKSteps][TPW], b[2][H];                    // a: the wave's weight tiles of the chunk; b: by stage parity
#pragma unroll
    for (int kk = 0; kk < kKSteps; ++kk) {
      a[kk][0] = lds_rd<0>(slot_addr + foff[kk] + wid * WWAVE);
      if constexpr (TPW > 1) a[kk][1] = lds_rd<4096>(slot_addr + foff[kk] + wid * WWAVE);
    }
    read_stage(slot_addr, 0, b[0]);
    wait_lgkm<H>();                                      // LDS returns in order: the weight fragments are in
#pragma unroll
    for (int kk = 0; kk < kKSteps; ++kk)
#pragma unroll
      for (int t = 0; t < TPW; ++t) pin(a[kk][t]);
    issue_w(c + PD, slot);                               // the tiles just copied out are free: refill them now
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int kk = WIDE ? st >> 1 : st;
      const int half = WIDE ? st & 1 : 0;
      const int bc = st & 1;
      if (st + 1 < NST) {
        read_stage(slot_addr, st + 1, b[bc ^ 1]);
        wait_lgkm<H>();
      } else {
        wait_lgkm<0>();
      }
#pragma unroll
      for (int j = 0; j < H; ++j) pin(b[bc][j]);
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int j = 0; j < H; ++j)
          acc[t][half * H + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[kk][t]),
                                                                       __builtin_bit_cast(bf16x8_t, b[bc][j]),
                                                                       acc[t][half * H + j], 0, 0, 0);
    }
  };

  // ---- pipeline: weights PD chunks ahead, activations PD - 1, one barrier per chunk ----
#pragma unroll
  for (int u = 0; u < PD - 1; ++u) {
    issue_w(u, u);
    issue_x(u, u);
  }
  issue_w(PD - 1, PD - 1);
  int slot = 0, nslot = PD - 1;                          // slot of chunk c, slot of chunk c - 1 (= c + PD - 1)
  for (int c = 0; c < n; ++c) {
    // chunk c has landed once only the DMAs issued after its activations are outstanding:
    // weights of c+1 .. c+PD-1 and activations of c+1 .. c+PD-2
    wait_vm<(PD - 1) * WPC + (PD - 2) * XP>();
    __builtin_amdgcn_s_barrier();                        // everyone's pieces of chunk c are visible,
                                                         // everyone is done reading the activations of chunk c - 1
    issue_x(c + PD - 1, nslot);
#ifdef WS_TRACE
    if (c == 0) WS_STAMP(1);
#endif
    compute(slot, c);
    slot = slot + 1 == PD ? 0 : slot + 1;
    nslot = nslot + 1 == PD ? 0 : nslot + 1;
  }

  WS_STAMP(2);
#ifdef WS_TRACE
  struct StampAtExit { __device__ ~StampAtExit() { __builtin_amdgcn_s_waitcnt(0); WS_STAMP(3); } } stamp_at_exit;
#endif
  if (!active) return;
  // lane holds C[m = 16 mt + r16][n = 16 tile + 4 g + r]
  const int n0 = tile * 16 + g * 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int64_t m = mt * 16 + r16;                           // output row
    float scale = 1.f;
    if constexpr (GROUPED) {
      const int id = p.sorted_ids[rb * (16 * MT) + mt * 16 + r16];
      if (id >= p.M) continue;
      m = id;
      if (p.row_scale) scale = p.row_scale[id];
    } else {
      if (m >= p.M) continue;
    }
    // one 4-wide piece of an output row: partial, or bias / router weight and the store
    auto put = [&](float (&o)[4], int nn, bool plain) {
      if (plain && !GROUPED && p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += bf2f(p.bias[nn + r]);
      }
      if (GROUPED && p.row_scale) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (p.round_before_scale ? rbf(o[r]) : o[r]) * scale;
      }
      if (GROUPED && p.out_f32) {
        *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(p.y) + m * p.y_stride + nn) = f32x4_t{o[0], o[1], o[2], o[3]};
      } else {
        uint2 w2;
        w2.x = pack_bf2(o[0], o[1]);
        w2.y = pack_bf2(o[2], o[3]);
        *reinterpret_cast<uint2*>(p.y + blocked_off(m, nn, p.y_stride, p.y_cstride)) = w2;
      }
    };
    if (TPW == 1 && !GROUPED && p.pair_silu == 2) {
      // lanes g = 0, 1 hold the gate values of columns 8 tile + 4 g + r, lanes g = 2, 3 the up values of the same columns
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float mine = acc[0][mt][r];
        const float other = __shfl_xor(mine, 32, 64);
        const float gb = rbf(g < 2 ? mine : other);
        const float sl = rbf(gb / (1.0f + expf(-gb)));
        o[r] = sl * rbf(g < 2 ? other : mine);
      }
      if (g < 2) put(o, tile * 8 + g * 4, false);
      continue;
    }
    if (TPW == 2 && p.pair_silu) {
      // linear -> bf16, silu -> bf16, product -> bf16 (activation.py:141-143)
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gb = rbf(acc[0][mt][r]);
        const float sl = rbf(gb / (1.0f + expf(-gb)));
        o[r] = sl * rbf(acc[TPW - 1][mt][r]);
      }
      put(o, n0, false);
      continue;
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {                      // plain form: tile t of the wave is output tile `tile + t * wtiles`
      const int nn = n0 + t * wtiles * 16;
      if (!GROUPED && p.part) {
        const int64_t e = (static_cast<int64_t>(split) * p.M + m) * p.N + nn;
        if (!(p.dbg & 2)) {   // write-through (sc1): the partials leave the L2 while the stream runs, not at the kernel boundary
                              // (qkv 18.7 -> 17.0, o 15.2 -> 14.7, down 29.1 -> 28.6 us per GEMM + combine pair: profiles/r06_exp2_gemm_ab.json)
          const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p.part, 0, static_cast<int>(static_cast<int64_t>(p.splits) * p.M * p.N * 4), 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[t][mt]), r, static_cast<int>(e * 4), 0, 16);
        } else {
          *reinterpret_cast<f32x4_t*>(p.part + e) = acc[t][mt];
        }
        continue;
      }
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = acc[t][mt][r];
      put(o, nn, true);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Split-K combine + epilogue.  part [splits, M, N] fp32, summed in split order.
//   mode 0: y[m, n]  = bf16(sum + bias[n])
//   mode 1: y[m, j]  = bf16( bf16(silu(bf16(sum[m, j]))) * bf16(sum[m, N/2 + j]) )       j < N/2
//           (linear -> bf16, activation.py:141-143 silu -> bf16, product -> bf16)
//   mode 2: h = bf16(sum + bias);  t = h + residual (fp32);  residual <- bf16(t);
//           y = bf16(t * rsqrt(mean(t^2) + eps) * norm_w)   (layernorm.py:786-820)
// mode 0/1: grid (column blocks, M); mode 2: one workgroup per row.
// ---------------------------------------------------------------------------------------------
struct CombineParams {
  const float* part;
  const uint16_t* bias;
  uint16_t* y;
  uint16_t* residual;
  const uint16_t* norm_w;
  int64_t y_stride, y_cstride, res_stride;
  int M, N, splits;
  float eps;
};

__global__ __launch_bounds__(256) void wstream_combine_kernel(CombineParams p) {
  const int m = blockIdx.y;
  const int n0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (n0 >= p.N) return;
  const int64_t ss = static_cast<int64_t>(p.M) * p.N;
  f32x4_t s = sum_splits(p.part + static_cast<int64_t>(m) * p.N + n0, ss, p.splits);
  if (p.bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] += bf2f(p.bias[n0 + r]);
  }
  uint2 w2;
  w2.x = pack_bf2(s[0], s[1]);
  w2.y = pack_bf2(s[2], s[3]);
  *reinterpret_cast<uint2*>(p.y + blocked_off(m, n0, p.y_stride, p.y_cstride)) = w2;
}

__global__ __launch_bounds__(256) void wstream_combine_silu_kernel(CombineParams p) {
  const int m = blockIdx.y;
  const int half = p.N >> 1;
  const int j0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (j0 >= half) return;
  const int64_t ss = static_cast<int64_t>(p.M) * p.N;
  const float* row = p.part + static_cast<int64_t>(m) * p.N;
  const f32x4_t gt = sum_splits(row + j0, ss, p.splits);
  const f32x4_t up = sum_splits(row + half + j0, ss, p.splits);
  float o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float gb = rbf(gt[r]);
    const float sl = rbf(gb / (1.0f + expf(-gb)));
    o[r] = sl * rbf(up[r]);
  }
  uint2 w2;
  w2.x = pack_bf2(o[0], o[1]);
  w2.y = pack_bf2(o[2], o[3]);
  *reinterpret_cast<uint2*>(p.y + blocked_off(m, j0, p.y_stride, p.y_cstride)) = w2;
}

constexpr int kNormThreads = 1024;
constexpr int kNormMaxVec = 4;   // float4 per thread -> N <= 16384

__global__ __launch_bounds__(kNormThreads) void wstream_combine_norm_kernel(CombineParams p) {
  __shared__ float scratch[16];
  const int m = blockIdx.x;
  const int64_t ss = static_cast<int64_t>(p.M) * p.N;
  const float* row = p.part + static_cast<int64_t>(m) * p.N;
  uint16_t* res = p.residual + static_cast<int64_t>(m) * p.res_stride;
  float t[kNormMaxVec][4];
  uint2 wv[kNormMaxVec];                 // norm weights: loaded with the partials, not behind the row reduction
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < kNormMaxVec; ++i) {
    const int n0 = (i * kNormThreads + threadIdx.x) * 4;
    if (n0 < p.N) {
      wv[i] = *reinterpret_cast<const uint2*>(p.norm_w + n0);
      f32x4_t s = sum_splits(row + n0, ss, p.splits);
      const uint2 rv = *reinterpret_cast<const uint2*>(res + n0);
      const float rr[4] = {bf_lo(rv.x), bf_hi(rv.x), bf_lo(rv.y), bf_hi(rv.y)};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float h = s[r];
        if (p.bias) h += bf2f(p.bias[n0 + r]);
        t[i][r] = rbf(h) + rr[r];
        sq += t[i][r] * t[i][r];
      }
      uint2 w2;
      w2.x = pack_bf2(t[i][0], t[i][1]);
      w2.y = pack_bf2(t[i][2], t[i][3]);
      *reinterpret_cast<uint2*>(res + n0) = w2;
    }
  }
  sq = block_sum(sq, scratch);
  const float rs = 1.0f / sqrtf(sq / static_cast<float>(p.N) + p.eps);
#pragma unroll
  for (int i = 0; i < kNormMaxVec; ++i) {
    const int n0 = (i * kNormThreads + threadIdx.x) * 4;
    if (n0 < p.N) {
      const float ww[4] = {bf_lo(wv[i].x), bf_hi(wv[i].x), bf_lo(wv[i].y), bf_hi(wv[i].y)};
      uint2 w2;
      w2.x = pack_bf2((t[i][0] * rs) * ww[0], (t[i][1] * rs) * ww[1]);
      w2.y = pack_bf2((t[i][2] * rs) * ww[2], (t[i][3] * rs) * ww[3]);
      *reinterpret_cast<uint2*>(p.y + blocked_off(m, n0, p.y_stride, p.y_cstride)) = w2;
    }
  }
}

// grid (token, item block)
__global__ __launch_bounds__(256) void wstream_combine_rope_kernel(RopeParams p) {
  const int m = blockIdx.x;
  const int per_head = p.head_dim >> 3;                 // 4-wide pair items per head
  const int n_rope = (p.num_q_heads + p.num_kv_heads) * per_head;
  const int it = blockIdx.y * 256 + threadIdx.x;        // rope items first, then the v row
  if (it < n_rope) {
    const int h = it / per_head;
    rope_pair_item(p, m, h, (it - h * per_head) * 4);
    return;
  }
  const int e = (it - n_rope) * 4;
  if (e < p.num_kv_heads * p.head_dim) rope_v_item(p, m, e);
}

#ifdef WS_EXPERIMENT
__global__ __launch_bounds__(256) void prefetch_head_kernel(const uint16_t* __restrict__ w, int64_t w_stride, int64_t N, int64_t K,
                                                            int splits, int pd, uint32_t* __restrict__ sink, int64_t pieces) {
  const int64_t i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= pieces) return;
  const int per_row = splits * pd * 16;
  const int64_t row = i / per_row;
  const int rem = static_cast<int>(i - row * per_row);
  const int sp = rem / (pd * 16), piece = rem - sp * (pd * 16);
  const int64_t nch = K / kKC;
  const int64_t cb = static_cast<int64_t>(sp) * nch / splits;
  const U4 v = ld16(w + row * w_stride + cb * kKC + piece * 8);
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x9e3779b9u && sink) sink[0] = v.x;       // keeps the load alive, practically never true
}
#endif

// the activation image of a chunk grows with M: beyond 64 rows only the narrower groups keep a 3-deep ring
template <int MT, int NW, int TPW>
int launch_main(const WsParams& p, hipStream_t st, int row_blocks = 0) {
  if constexpr (ring_depth(MT, NW, TPW) >= 3) {
    dim3 grid((p.ntiles / TPW + NW - 1) / NW, row_blocks ? row_blocks : p.splits);
    if (row_blocks) {
      if constexpr (MT <= 4) hipLaunchKernelGGL((wstream_gemm_kernel<MT, NW, TPW, true>), grid, dim3(64 * NW), 0, st, p);
      else return -1;
    } else {
      hipLaunchKernelGGL((wstream_gemm_kernel<MT, NW, TPW, false>), grid, dim3(64 * NW), 0, st, p);
    }
    return 0;
  } else {
    return -1;
  }
}

template <int MT>
int launch_nw(const WsParams& p, int nw, bool two_tiles, hipStream_t st, int row_blocks = 0) {
  if (two_tiles) {                                    // two tiles per wave: half the waves for the same LDS ring
    switch (nw) {
      case 2: return launch_main<MT, 2, 2>(p, st, row_blocks);
      case 3: return launch_main<MT, 3, 2>(p, st, row_blocks);
      case 4: return launch_main<MT, 4, 2>(p, st, row_blocks);
      default: return -1;
    }
  }
  switch (nw) {
    case 4: return launch_main<MT, 4, 1>(p, st, row_blocks);
    case 5: return launch_main<MT, 5, 1>(p, st, row_blocks);
    case 6: return launch_main<MT, 6, 1>(p, st, row_blocks);
    case 7: return launch_main<MT, 7, 1>(p, st, row_blocks);
    case 8: return launch_main<MT, 8, 1>(p, st, row_blocks);
    default: return -1;
  }
}

}  // namespace

static int g_ws_debug_flags = 0;

extern "C" {

int sgl_amd_debug_wstream_flags(int flags) {
  const int old = g_ws_debug_flags;
  if (flags >= 0) g_ws_debug_flags = flags;
  return old;
}

#ifdef WS_EXPERIMENT
// benchmarks/r02_exp16_prefetch.py: read the bytes a following weight-streaming launch asks for first (the first `pd`
// K chunks of every row in each of its `splits` K ranges), so that they wait in the memory-side cache
int sgl_amd_debug_prefetch_head(const void* w, int64_t w_row_stride, int64_t N, int64_t K, int splits, int pd, void* sink, void* stream) {
  const int64_t pieces = N * splits * pd * 16;              // 16-byte pieces
  hipLaunchKernelGGL(prefetch_head_kernel, dim3(static_cast<unsigned>((pieces + 255) / 256)), dim3(256), 0, as_stream(stream),
                     static_cast<const uint16_t*>(w), w_row_stride, N, K, splits, pd, static_cast<uint32_t*>(sink), pieces);
  return 0;
}
#endif

#ifdef WS_TRACE
int sgl_amd_debug_ws_trace(void* buf) {
  uint64_t* p = static_cast<uint64_t*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_ws_trace), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

int sgl_amd_wstream_gemm_max_rows(void) { return 128; }

int64_t sgl_amd_wstream_gemm_workspace_floats(int64_t M, int64_t N, int num_k_splits) {
  return static_cast<int64_t>(num_k_splits) * M * N;   /* needed when num_k_splits > 1 or epilogue != 0 */
}

// shared by the two entry points: validates the GEMM part and launches the main kernel
static int wstream_launch_main(const char* who, const void* x, const void* w, const void* bias, void* y, int64_t M, int64_t N,
                               int64_t K, int64_t x_row_stride, int64_t x_chunk_stride, int64_t w_row_stride,
                               int64_t y_row_stride, int64_t y_chunk_stride, bool fused_silu, int tiles_per_wave,
                               bool to_partials, int waves_per_group, int num_k_splits, void* ws_partials, hipStream_t st) {
  SGL_CHECK_ARG(M >= 1 && M <= 128, "%s: M=%lld rows (supported: 1..128)", who, (long long)M);
  SGL_CHECK_ARG(N > 0 && N % 16 == 0 && K >= kKC && K % kKC == 0,
                "%s: need N %% 16 == 0 and K %% %d == 0 (got N=%lld K=%lld)", who, kKC, (long long)N, (long long)K);
  SGL_CHECK_ARG(x_row_stride % 8 == 0 && w_row_stride % 8 == 0 && y_row_stride % 4 == 0,
                "%s: row strides must keep 16-byte (x, w) / 8-byte (y) alignment", who);
  // the fused silu epilogue has two forms: two tiles per wave (gate tile + up tile), or -- tiles_per_wave == 1 -- one tile of
  // 8 gate + 8 up rows per wave
  const bool two_tiles = (fused_silu && tiles_per_wave != 1) || tiles_per_wave == 2;
  SGL_CHECK_ARG(tiles_per_wave == 0 || tiles_per_wave == 1 || tiles_per_wave == 2, "%s: tiles_per_wave must be 1 or 2 (0: the default of the form)", who);
  SGL_CHECK_ARG(!two_tiles || N % 32 == 0, "%s: two tiles per wave need N %% 32 == 0", who);
  SGL_CHECK_ARG(two_tiles ? (waves_per_group >= 2 && waves_per_group <= 4) : (waves_per_group >= 4 && waves_per_group <= 8),
                "%s: waves_per_group must be 4..8 (2..4 with two tiles per wave), got %d", who, waves_per_group);
  SGL_CHECK_ARG(num_k_splits >= 1 && num_k_splits <= K / kKC, "%s: bad split count %d", who, num_k_splits);
  SGL_CHECK_ARG(!to_partials || ws_partials, "%s: needs the fp32 partials workspace", who);
  WsParams p{};
  p.x = static_cast<const uint16_t*>(x);
  p.w = static_cast<const uint16_t*>(w);
  p.bias = to_partials ? nullptr : static_cast<const uint16_t*>(bias);
  p.y = static_cast<uint16_t*>(y);
  p.part = to_partials ? static_cast<float*>(ws_partials) : nullptr;
  p.x_stride = x_row_stride; p.w_stride = w_row_stride; p.y_stride = y_row_stride;
  SGL_CHECK_ARG(x_chunk_stride % 8 == 0 && y_chunk_stride % 4 == 0 && x_chunk_stride >= 0 && y_chunk_stride >= 0,
                "%s: chunk strides must keep 16-byte (x) / 8-byte (y) alignment", who);
  p.x_cstride = x_chunk_stride ? x_chunk_stride : kKC;
  p.y_cstride = y_chunk_stride ? y_chunk_stride : 128;
  p.M = static_cast<int>(M); p.N = static_cast<int>(N); p.K = static_cast<int>(K);
  p.splits = num_k_splits; p.ntiles = static_cast<int>(N / 16);
  p.pair_silu = fused_silu ? (two_tiles ? 1 : 2) : 0;
  p.dbg = g_ws_debug_flags;
  if (static_cast<int64_t>(num_k_splits) * M * N * 4 >= (int64_t{1} << 31)) p.dbg |= 2;
  int rc;
  switch (static_cast<int>((M + 15) / 16)) {
    case 1: rc = launch_nw<1>(p, waves_per_group, two_tiles, st); break;
    case 2: rc = launch_nw<2>(p, waves_per_group, two_tiles, st); break;
    case 3: rc = launch_nw<3>(p, waves_per_group, two_tiles, st); break;
    case 4: rc = launch_nw<4>(p, waves_per_group, two_tiles, st); break;
    case 5: case 6: rc = launch_nw<6>(p, waves_per_group, two_tiles, st); break;
    default: rc = launch_nw<8>(p, waves_per_group, two_tiles, st); break;
  }
  SGL_CHECK_ARG(rc == 0, "%s: unsupported configuration", who);
  return 0;
}

int sgl_amd_wstream_gemm(const void* x, const void* w, const void* bias, void* y, int64_t M, int64_t N, int64_t K,
                         int64_t x_row_stride, int64_t x_chunk_stride, int64_t w_row_stride, int64_t y_row_stride,
                         int64_t y_chunk_stride, int epilogue, void* residual, int64_t residual_row_stride,
                         const void* norm_weight, float eps, int waves_per_group, int tiles_per_wave, int num_k_splits,
                         void* ws_partials, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  if (M == 0) return 0;
  SGL_CHECK_ARG(epilogue >= 0 && epilogue <= 2, "wstream_gemm: epilogue must be 0 (bias), 1 (silu_and_mul) or 2 (add_rmsnorm)");
  // silu(gate)*up in the GEMM's own epilogue: two tiles per wave (N % 32 == 0) or one interleaved tile (tiles_per_wave == 1, N % 16 == 0)
  const bool fused_silu = epilogue == 1 && num_k_splits == 1 && !bias && (tiles_per_wave == 1 ? N % 16 == 0 : N % 32 == 0);
  const bool combine = !fused_silu && (num_k_splits > 1 || epilogue != 0);
  SGL_CHECK_ARG(epilogue != 1 || N % 8 == 0, "wstream_gemm: silu_and_mul needs N %% 8 == 0");
  SGL_CHECK_ARG(epilogue != 1 || !bias, "wstream_gemm: the silu_and_mul epilogue takes no bias");
  SGL_CHECK_ARG(epilogue != 2 || (residual && norm_weight && N <= kNormThreads * kNormMaxVec * 4 && residual_row_stride % 4 == 0),
                "wstream_gemm: add_rmsnorm needs residual, norm_weight and N <= %d", kNormThreads * kNormMaxVec * 4);
  hipStream_t st = as_stream(stream);
  SGL_CHECK_ARG(y_chunk_stride == 0 || (epilogue == 1 ? N % 256 == 0 : N % 128 == 0),
                "wstream_gemm: a chunk-major y needs whole 128-column blocks");
  if (int rc = wstream_launch_main("wstream_gemm", x, w, bias, y, M, N, K, x_row_stride, x_chunk_stride, w_row_stride, y_row_stride,
                                   y_chunk_stride, fused_silu, tiles_per_wave, combine, waves_per_group, num_k_splits, ws_partials, st))
    return rc;
  if (combine) {
    CombineParams c{};
    c.part = static_cast<const float*>(ws_partials); c.bias = static_cast<const uint16_t*>(bias); c.y = static_cast<uint16_t*>(y);
    c.residual = static_cast<uint16_t*>(residual); c.norm_w = static_cast<const uint16_t*>(norm_weight);
    c.y_stride = y_row_stride; c.y_cstride = y_chunk_stride ? y_chunk_stride : 128; c.res_stride = residual_row_stride;
    c.M = static_cast<int>(M); c.N = static_cast<int>(N); c.splits = num_k_splits; c.eps = eps;
    if (epilogue == 0) {
      hipLaunchKernelGGL(wstream_combine_kernel, dim3((c.N / 4 + 255) / 256, c.M), dim3(256), 0, st, c);
    } else if (epilogue == 1) {
      hipLaunchKernelGGL(wstream_combine_silu_kernel, dim3((c.N / 8 + 255) / 256, c.M), dim3(256), 0, st, c);
    } else {
      hipLaunchKernelGGL(wstream_combine_norm_kernel, dim3(c.M), dim3(kNormThreads), 0, st, c);
    }
  }
  SGL_CHECK_LAUNCH("wstream_gemm");
  return 0;
}

int sgl_amd_wstream_moe_gemm(const void* a, const void* w, void* c, const int32_t* sorted_token_ids,
                             const int32_t* expert_ids, const int32_t* num_tokens_post_padded,
                             const float* topk_weights, int mul_routed_weight, int round_before_scale, int top_k_div,
                             int64_t num_valid_ids, int64_t N, int64_t K, int64_t a_row_stride, int64_t w_row_stride,
                             int64_t w_expert_stride, int64_t c_row_stride, int block_m, int64_t max_m_blocks,
                             int fuse_silu, int out_f32, int waves_per_group, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(block_m == 16 || block_m == 32 || block_m == 48 || block_m == 64,
                "wstream_moe_gemm: block_m=%d (supported: 16/32/48/64, must equal the moe_align block size)", block_m);
  const int64_t wn = fuse_silu ? 2 * N : N;             // weight rows per expert
  SGL_CHECK_ARG(N > 0 && wn % (fuse_silu ? 32 : 16) == 0 && K >= kKC && K % kKC == 0,
                "wstream_moe_gemm: need N %% 16 == 0 and K %% %d == 0 (got N=%lld K=%lld)", kKC, (long long)N, (long long)K);
  SGL_CHECK_ARG(a_row_stride % 8 == 0 && w_row_stride % 8 == 0 && c_row_stride % 4 == 0,
                "wstream_moe_gemm: row strides must keep 16-byte (a, w) / 8-byte (c) alignment");
  SGL_CHECK_ARG(top_k_div >= 1 && max_m_blocks <= 65535, "wstream_moe_gemm: bad top_k_div / too many row blocks");
  SGL_CHECK_ARG(!mul_routed_weight || topk_weights, "wstream_moe_gemm: mul_routed_weight needs topk_weights");
  SGL_CHECK_ARG(!(fuse_silu && out_f32), "wstream_moe_gemm: the silu form writes bf16");
  SGL_CHECK_ARG(fuse_silu ? (waves_per_group >= 2 && waves_per_group <= 4) : (waves_per_group >= 4 && waves_per_group <= 8),
                "wstream_moe_gemm: waves_per_group must be 4..8 (2..4 for the silu form), got %d", waves_per_group);
  if (max_m_blocks == 0 || num_valid_ids == 0) return 0;
  WsParams p{};
  p.x = static_cast<const uint16_t*>(a);
  p.w = static_cast<const uint16_t*>(w);
  p.y = static_cast<uint16_t*>(c);
  p.x_stride = a_row_stride; p.w_stride = w_row_stride; p.y_stride = c_row_stride; p.w_expert_stride = w_expert_stride;
  p.x_cstride = kKC; p.y_cstride = 128;
  p.M = static_cast<int>(num_valid_ids); p.N = static_cast<int>(wn); p.K = static_cast<int>(K);
  p.splits = 1; p.ntiles = static_cast<int>(wn / 16);
  p.sorted_ids = sorted_token_ids; p.expert_ids = expert_ids; p.num_post_pad = num_tokens_post_padded;
  p.row_scale = mul_routed_weight ? topk_weights : nullptr;
  p.topk_div = top_k_div; p.out_f32 = out_f32; p.round_before_scale = round_before_scale;
  p.pair_silu = fuse_silu ? 1 : 0;
  hipStream_t st = as_stream(stream);
  const int blocks = static_cast<int>(max_m_blocks);
  int rc;
  switch (block_m / 16) {
    case 1: rc = launch_nw<1>(p, waves_per_group, fuse_silu != 0, st, blocks); break;
    case 2: rc = launch_nw<2>(p, waves_per_group, fuse_silu != 0, st, blocks); break;
    case 3: rc = launch_nw<3>(p, waves_per_group, fuse_silu != 0, st, blocks); break;
    default: rc = launch_nw<4>(p, waves_per_group, fuse_silu != 0, st, blocks); break;
  }
  SGL_CHECK_ARG(rc == 0, "wstream_moe_gemm: unsupported configuration");
  SGL_CHECK_LAUNCH("wstream_moe_gemm");
  return 0;
}

int sgl_amd_wstream_qkv_rope(const void* x, const void* w_qkv, const void* bias, void* q_out, int64_t M, int64_t K,
                             int num_q_heads, int num_kv_heads, int head_dim, int64_t x_row_stride, int64_t x_chunk_stride,
                             int64_t w_row_stride, int64_t q_row_stride, const int64_t* positions, const void* cos_sin_cache, int cache_is_f32,
                             int64_t rotary_dim, void* k_cache, void* v_cache, const int64_t* cache_loc,
                             int64_t cache_row_stride, int kv_fp8, float k_scale, float v_scale, int page_size,
                             int kv_layout_hnd, int waves_per_group, int tiles_per_wave, int num_k_splits,
                             void* ws_partials, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  if (M == 0) return 0;
  SGL_CHECK_ARG(num_q_heads > 0 && num_kv_heads > 0 && head_dim % 16 == 0 && rotary_dim == head_dim,
                "wstream_qkv_rope: full-head neox rotary only (head_dim %% 16 == 0, rotary_dim == head_dim)");
  SGL_CHECK_ARG(positions && cos_sin_cache && k_cache && v_cache && cache_loc && q_out, "wstream_qkv_rope: null argument");
  SGL_CHECK_ARG(q_row_stride % 4 == 0 && cache_row_stride % 4 == 0, "wstream_qkv_rope: q / cache row strides must be multiples of 4 elements");
  const int64_t N = static_cast<int64_t>(num_q_heads + 2 * num_kv_heads) * head_dim;
  hipStream_t st = as_stream(stream);
  RopeParams r{};
  r.part = static_cast<const float*>(ws_partials); r.bias = static_cast<const uint16_t*>(bias);
  r.q_out = static_cast<uint16_t*>(q_out); r.k_cache = static_cast<uint16_t*>(k_cache); r.v_cache = static_cast<uint16_t*>(v_cache);
  r.positions = positions; r.cache_loc = cache_loc; r.cos_sin = cos_sin_cache;
  r.q_stride = q_row_stride; r.cache_row_stride = cache_row_stride;
  r.M = static_cast<int>(M); r.N = static_cast<int>(N); r.splits = num_k_splits;
  r.num_q_heads = num_q_heads; r.num_kv_heads = num_kv_heads; r.head_dim = head_dim; r.cache_f32 = cache_is_f32;
  r.generic = (kv_fp8 || kv_layout_hnd) ? 1 : 0;
  SGL_CHECK_ARG(!kv_fp8 || (k_scale > 0.f && v_scale > 0.f), "wstream_qkv_rope: fp8 KV needs positive k_scale / v_scale");
  SGL_CHECK_ARG(make_kv_format(&r.fmt, cache_row_stride, num_kv_heads, head_dim, page_size, kv_layout_hnd, kv_fp8),
                "wstream_qkv_rope: HND pools need a power-of-two page_size (got %d); row / page strides below 4 GiB", page_size);
  r.inv_k_scale = kv_fp8 ? 1.0f / k_scale : 1.0f; r.inv_v_scale = kv_fp8 ? 1.0f / v_scale : 1.0f;
  if (int rc = wstream_launch_main("wstream_qkv_rope", x, w_qkv, nullptr, nullptr, M, N, K, x_row_stride, x_chunk_stride, w_row_stride,
                                   4, 0, false, tiles_per_wave, true, waves_per_group, num_k_splits, ws_partials, st))
    return rc;
  const int items = (num_q_heads + num_kv_heads) * (head_dim / 8) + num_kv_heads * head_dim / 4;
  hipLaunchKernelGGL(wstream_combine_rope_kernel, dim3(r.M, (items + 255) / 256), dim3(256), 0, st, r);
  SGL_CHECK_LAUNCH("wstream_qkv_rope");
  return 0;
}

}  // extern "C"

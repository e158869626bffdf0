// Row-tiled grouped GEMM for prefill-sized MoE batches on gfx950 (MFMA bound form).
//
// Replaces (reference, /root/reference/python/sglang):
//   srt/layers/moe/moe_runner/triton_utils/fused_moe_triton_kernels.py:324 fused_moe_kernel, :771
//   invoke_fused_moe_kernel with BLOCK_SIZE_M >= 64 (the tuned configs of fused_moe_triton/configs/*.json), called
//   twice by fused_experts (triton_utils/fused_moe.py:242-455): up projection (+ silu_and_mul) and down projection
//   (x router weight).  Oracle: srt/layers/moe/fused_moe_native.py:61-164 (oracle/ops.py moe_forward).
//
// The weight-streaming form (wstream_gemm.hip) reads an expert's weights once per 16..64-row block: right for decode,
// where every expert sees a handful of rows, ruinous for prefill, where an expert owns ~1000 rows and would stream its
// 100+ MB sixteen times.  Here a workgroup owns a 128 (rows of ONE expert) x 128 (output columns) tile and walks K in
// 64-wide steps with both operands staged in LDS:
//   * rows are gathered through sorted_token_ids (moe_align_block_size with block 128), weights of expert
//     expert_ids[block]; both tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
//     instruction), two tiles deep, the XOR swizzle applied on the global side so the fragment reads
//     (ds_read_b128) are conflict-free;
//   * 4 waves as 2 (rows) x 2 (columns), each 64 x 64 = 4 x 4 MFMA 16x16x32 tiles, computed transposed
//     (C^T = W x^T) so a lane ends up with 4 consecutive output columns of one token: 8-byte stores;
//   * up projection: the 128 weight rows of a tile are [32 gate | 32 up] x 2 halves, so every wave holds the gate
//     and the up value of the same (token, column) in the same lane and applies silu(gate) * up in registers
//     (torch-native rounding points: activation.py:141-143) -- the [rows, 2N] intermediate never exists;
//   * down projection: fp32(bf16(acc)) x router weight into the fp32 buffer moe_sum_reduce adds up
//     (fused_moe_native.py:157-163).
//
// Round 4: a 256 x 256 x 64 form (moe_gemm256_kernel) for experts that own many rows (prefill batches of a few thousand
// tokens): 8 waves as 2 (weight rows) x 4 (token rows), 128 x 64 outputs per wave on MFMA 16x16x32 (24 LDS fragment reads
// per 64 products instead of the 128 x 128 form's 16 per 32; half the L2 -> LDS bytes per flop), both operands by LDS-DMA
// into TWO 64 KiB stages with ONE barrier per K step (the next step's tiles fly while this one multiplies), the
// fragment reads of a quarter step issued ahead of the products of the quarter before it, and a workgroup order that
// keeps the 32 workgroups an XCD runs at a time on a 4 (row blocks) x 8 (column tiles) patch of the output, so that its
// L2 fetches 12 operand tiles per step for 32 workgroups.  Same contract, same rounding points, same epilogues.
#include "common.hpp"
#include "sglang_amd.h"

using namespace sgl_amd;

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) unsigned char* lds_bytes_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int kBM = 128, kBN = 128, kBK = 64;
constexpr int kTileBytes = kBM * kBK * 2;          // 16 KiB per operand tile
constexpr int kThreads = 256;

struct TiledParams {
  const uint16_t* a;            // [rows, K]
  const uint16_t* w;            // [E, WN, K]   (WN = 2N for the up projection)
  void* c;                      // [num_valid_ids, N] bf16 or fp32
  const int32_t* sorted_ids;    // [>= blocks * 128] pair ids, padding >= num_valid
  const int32_t* expert_ids;    // [blocks]
  const int32_t* num_post_pad;  // [1]
  const float* row_scale;       // optional [num_valid_ids]
  int64_t a_stride, w_row_stride, w_expert_stride, c_stride;
  int num_valid, N, K, topk_div;
  int fuse_silu, out_f32, round_before_scale;
  int eid_shift;                // expert_ids is indexed by (row block >> eid_shift): 128-row tiles over a 256-row alignment
  int n_tiles;                  // 256-form: output-column tiles (its grid is one-dimensional)
};

__device__ __forceinline__ void dma16(const uint16_t* src, lds_ptr_t dst) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)(src), dst, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ u32x4_t lds_rd(uint32_t addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N)); }
__device__ __forceinline__ void pin(u32x4_t& v) { asm volatile("" : "+v"(v)); }

__global__ __launch_bounds__(kThreads, 2) void moe_tiled_gemm_kernel(TiledParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * 2 * kTileBytes];   // [buf][A | B]
  const int rb = blockIdx.y;
  if (rb * kBM >= p.num_post_pad[0]) return;
  const int e = p.expert_ids[rb >> p.eid_shift];
  if (e < 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;               // this wave's 64-row / 64-column quadrant
  const int r16 = lane & 15, g = lane >> 4;
  const int nt0 = blockIdx.x;                          // output-column tile (64 columns when fused, 128 otherwise)
  const uint16_t* wbase = p.w + static_cast<int64_t>(e) * p.w_expert_stride;

  // ---- DMA roles: instruction j of a tile covers its rows 8j .. 8j+7; lane i -> (row 8j + i/8, LDS slot i%8) and
  // fetches global 16-byte piece (slot ^ (row & 7)) of that row.  Wave w issues j = 4w .. 4w+3 of A and of B.
  const int dr = lane >> 3, ds = lane & 7;
  const uint16_t* asrc[4];
  const uint16_t* bsrc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int row = (wid * 4 + u) * 8 + dr;            // 0..127
    const int piece = (ds ^ (row & 7)) * 8;
    const int id = p.sorted_ids[rb * kBM + row];
    const int srow = id < p.num_valid ? id / p.topk_div : 0;     // padding rows read row 0, their outputs are dropped
    asrc[u] = p.a + static_cast<int64_t>(srow) * p.a_stride + piece;
    int wrow;
    if (p.fuse_silu) {
      // tile rows: [32 gate | 32 up] for column half 0, then for half 1; output columns nt0 * 64 + half * 32 + c
      const int half = row >> 6, in = row & 63;
      const int col = nt0 * 64 + half * 32 + (in & 31);
      wrow = (in < 32 ? 0 : p.N) + (col < p.N ? col : p.N - 1);
    } else {
      const int col = nt0 * kBN + row;
      wrow = col < p.N ? col : p.N - 1;
    }
    bsrc[u] = wbase + static_cast<int64_t>(wrow) * p.w_row_stride + piece;
  }
  lds_bytes_t sm3 = (lds_bytes_t)(smem);
  auto issue = [&](int kt, int buf) {
    const int koff = kt * kBK;
#pragma unroll
    for (int u = 0; u < 4; ++u) dma16(asrc[u] + koff, (lds_ptr_t)(sm3 + buf * 2 * kTileBytes + (wid * 4 + u) * 1024));
#pragma unroll
    for (int u = 0; u < 4; ++u) dma16(bsrc[u] + koff, (lds_ptr_t)(sm3 + buf * 2 * kTileBytes + kTileBytes + (wid * 4 + u) * 1024));
  };

  // ---- fragment addresses: row r of a tile at r * 128 bytes, piece pc in slot pc ^ (r & 7) ----
  const uint32_t sm_addr = (uint32_t)(uintptr_t)(sm3);
  uint32_t foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) foff[kk] = r16 * 128 + (((kk * 4 + g) ^ (r16 & 7)) & 7) * 16;
  const uint32_t a_quad = wm * 64 * 128;               // this wave's rows of the A tile
  const uint32_t b_quad = kTileBytes + wn * 64 * 128;  // and of the B tile

  f32x4_t acc[4][4];                                   // [weight 16-row tile][token 16-row tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / kBK;
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {
      issue(kt + 1, buf ^ 1);                          // every wave finished reading that buffer before the last barrier
      wait_vm<8>();                                    // tile kt landed (the 8 pieces of tile kt + 1 may still fly)
    } else {
      wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    const uint32_t base = sm_addr + buf * 2 * kTileBytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4_t wf[4], xf[4];
      const uint32_t ad = base + foff[kk];
      wf[0] = lds_rd<0>(ad + b_quad); wf[1] = lds_rd<2048>(ad + b_quad); wf[2] = lds_rd<4096>(ad + b_quad); wf[3] = lds_rd<6144>(ad + b_quad);
      xf[0] = lds_rd<0>(ad + a_quad); xf[1] = lds_rd<2048>(ad + a_quad); xf[2] = lds_rd<4096>(ad + a_quad); xf[3] = lds_rd<6144>(ad + a_quad);
      wait_lgkm<0>();
#pragma unroll
      for (int i = 0; i < 4; ++i) { pin(wf[i]); pin(xf[i]); }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[i]), __builtin_bit_cast(bf16x8_t, xf[j]),
                                                              acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();                      // all reads of `buf` done before iteration kt + 1 refills it
  }

  // ---- epilogue: lane holds C^T[n = 16 i + 4 g + r][m = 16 j + r16] ----
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = wm * 64 + j * 16 + r16;
    const int id = p.sorted_ids[rb * kBM + row];
    if (id >= p.num_valid) continue;
    const float scale = p.row_scale ? p.row_scale[id] : 1.f;
    if (p.fuse_silu) {
      // weight tiles 0,1 = gate columns [0,32) of this wave's half, tiles 2,3 = the matching up columns
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int col = nt0 * 64 + wn * 32 + i * 16 + g * 4;
        if (col >= p.N) continue;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gb = rbf(acc[i][j][r]);
          const float sl = rbf(gb / (1.0f + expf(-gb)));
          o[r] = sl * rbf(acc[i + 2][j][r]);
        }
        uint2 w2;
        w2.x = pack_bf2(o[0], o[1]); w2.y = pack_bf2(o[2], o[3]);
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.c) + static_cast<int64_t>(id) * p.c_stride + col) = w2;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = nt0 * kBN + wn * 64 + i * 16 + g * 4;
        if (col >= p.N) continue;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[i][j][r];
          o[r] = p.row_scale ? (p.round_before_scale ? rbf(v) : v) * scale : v;
        }
        if (p.out_f32) {
          *reinterpret_cast<f32x4_t*>(static_cast<float*>(p.c) + static_cast<int64_t>(id) * p.c_stride + col) = f32x4_t{o[0], o[1], o[2], o[3]};
        } else {
          uint2 w2;
          w2.x = pack_bf2(o[0], o[1]); w2.y = pack_bf2(o[2], o[3]);
          *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.c) + static_cast<int64_t>(id) * p.c_stride + col) = w2;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 (weight rows) x 256 (token rows) x 64 form.  LDS: stage s at s * 64 KiB = [W tile 256 x 128 B | X tile 256 x 128 B],
// row r at r * 128, 16-byte piece pc of a row in slot pc ^ (r & 7) (the swizzle is applied on the global side of the
// LDS-DMA; the ds_read_b128 fragment reads of a 16-lane group then cover all 16 slots of a 256-byte bank row).
// ---------------------------------------------------------------------------------------------------------------------
// experiment switches (benchmarks/build_variant.py): loop form, order pinning, priority, and two ablations
// (measured, profiles/r04_exp4_moe_gemm_ab.json: the plain loop with the quarters pinned is the fastest or within noise of
// it on every shape; the rotated form -- barrier ahead of the last quarter, next stage's first fragments read behind it --
// measured equal to 5 % slower although it hides every LDS latency: what bounds the walk is the L2 -> LDS stream of a
// 64 KiB stage per step, which has exactly one step to land in either form)
#ifndef MOE256_ROTATE
#define MOE256_ROTATE 0
#endif
#ifndef MOE256_L2_AHEAD
#define MOE256_L2_AHEAD 0
#endif
#ifndef MOE256_GM
#define MOE256_GM 8
#endif
#ifndef MOE256_PIN
#define MOE256_PIN 1
#endif
#if MOE256_PIN
#define MOE256_PINNED() __builtin_amdgcn_sched_barrier(0)
#else
#define MOE256_PINNED()
#endif
#ifdef MOE256_SETPRIO
#define MOE256_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define MOE256_PRIO(x)
#endif
#ifdef MOE256_NODMA      // timing only (wrong results): the K walk without its LDS-DMA traffic
#define MOE256_ISSUE(kt, st)
#else
#define MOE256_ISSUE(kt, st) issue(kt, st)
#endif
constexpr int kT2 = 256;                             // tile edge (both operands)
constexpr int kOp2 = kT2 * kBK * 2;                  // 32 KiB per operand tile
constexpr int kStage2 = 2 * kOp2;                    // 64 KiB per stage
constexpr int kThreads2 = 512;

__global__ __launch_bounds__(kThreads2, 1) void moe_gemm256_kernel(TiledParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * kStage2];
  // ---- workgroup -> (row block, column tile).  Workgroups go to the XCDs round robin by their linear index; XCD x takes
  // the patches x, x + 8, ... of 32 workgroups = GM (8) row blocks x GN (4) column tiles (8 x 4 measured 1-4 % ahead of 4 x 8 and 16 x 2, 20 % ahead of 1 x 32: profiles/r04_exp4_moe_gemm_ab.json), consecutive patches walk the column
  // tiles of one group of row blocks (the eight XCDs then work on the same token rows: one copy in the memory-side cache).
  const int id = blockIdx.x;
  const int patch = (id >> 8) * 8 + (id & 7);
  const int within = (id >> 3) & 31;
  constexpr int GM = MOE256_GM, GN = 32 / GM;          // row blocks x column tiles of a patch
  const int npn = p.n_tiles / GN;                      // column patches (n_tiles is padded to a multiple of GN by the host)
  const int rb = (patch / npn) * GM + (within % GM);
  const int nt0 = (patch % npn) * GN + (within / GM);
  if (rb * kT2 >= p.num_post_pad[0]) return;
  const int e = p.expert_ids[rb];
  if (e < 0) return;
  const int cols_per_tile = p.fuse_silu ? 128 : 256;
  if (nt0 * cols_per_tile >= p.N) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;               // this wave: weight rows 128 wr .., token rows 64 wc ..
  const int r16 = lane & 15, g = lane >> 4;
  const uint16_t* wbase = p.w + static_cast<int64_t>(e) * p.w_expert_stride;

  // ---- DMA roles: instruction j of an operand tile covers its rows 8j .. 8j+7 (lane i -> row 8j + i/8, slot i%8, global
  // piece slot ^ (row & 7)); wave w issues j = 4w .. 4w+3 of the W tile and of the X tile: 8 x 1 KiB per wave and K step.
  const int dr = lane >> 3, ds = lane & 7;
  const uint16_t* asrc[4];                             // token rows (X tile)
  const uint16_t* bsrc[4];                             // weight rows (W tile)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int row = (wid * 4 + u) * 8 + dr;            // 0..255
    const int piece = (ds ^ (row & 7)) * 8;
    const int sid = p.sorted_ids[rb * kT2 + row];
    const int srow = sid < p.num_valid ? sid / p.topk_div : 0;     // padding rows read row 0, their outputs are dropped
    asrc[u] = p.a + static_cast<int64_t>(srow) * p.a_stride + piece;
    int wrow;
    if (p.fuse_silu) {
      // tile rows: per half of 128, [64 gate | 64 up] of output columns nt0 * 128 + half * 64 + c
      const int half = row >> 7, in = row & 127;
      const int col = nt0 * 128 + half * 64 + (in & 63);
      wrow = (in < 64 ? 0 : p.N) + (col < p.N ? col : p.N - 1);
    } else {
      const int col = nt0 * 256 + row;
      wrow = col < p.N ? col : p.N - 1;
    }
    bsrc[u] = wbase + static_cast<int64_t>(wrow) * p.w_row_stride + piece;
  }
  lds_bytes_t sm3 = (lds_bytes_t)(smem);
  auto issue = [&](int kt, int stage) {
    const int koff = kt * kBK;
#pragma unroll
    for (int u = 0; u < 4; ++u) dma16(bsrc[u] + koff, (lds_ptr_t)(sm3 + stage * kStage2 + (wid * 4 + u) * 1024));
#pragma unroll
    for (int u = 0; u < 4; ++u) dma16(asrc[u] + koff, (lds_ptr_t)(sm3 + stage * kStage2 + kOp2 + (wid * 4 + u) * 1024));
  };

  const uint32_t sm_addr = (uint32_t)(uintptr_t)(sm3);
  uint32_t foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) foff[kk] = r16 * 128 + (((kk * 4 + g) ^ (r16 & 7)) & 7) * 16;
  const uint32_t w_base = wr * (128 * 128);            // this wave's 128 weight rows of the W tile
  const uint32_t x_base = kOp2 + wc * (64 * 128);      // and its 64 token rows of the X tile

  f32x4_t acc[8][4];                                   // [weight 16-row tile][token 16-row tile]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // a quarter step = 16 products: one k-step (32 of the 64) x one half (4 of 8) of the wave's weight-row tiles
  auto read_w = [&](uint32_t base, int kk, int mi, u32x4_t (&wf)[4]) {
    const uint32_t ad = base + w_base + foff[kk] + mi * (64 * 128);
    wf[0] = lds_rd<0>(ad); wf[1] = lds_rd<2048>(ad); wf[2] = lds_rd<4096>(ad); wf[3] = lds_rd<6144>(ad);
  };
  auto read_x = [&](uint32_t base, int kk, u32x4_t (&xf)[4]) {
    const uint32_t ad = base + x_base + foff[kk];
    xf[0] = lds_rd<0>(ad); xf[1] = lds_rd<2048>(ad); xf[2] = lds_rd<4096>(ad); xf[3] = lds_rd<6144>(ad);
  };
  auto products = [&](int mi, u32x4_t (&wf)[4], u32x4_t (&xf)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) pin(wf[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) pin(xf[j]);
#ifdef MOE256_NOMFMA     // timing only (wrong results): staging and fragment reads without the matrix work
    acc[mi * 4][0][0] += __uint_as_float(wf[0].x ^ xf[0].x ^ wf[3].y ^ xf[3].y);
    return;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[mi * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[i]), __builtin_bit_cast(bf16x8_t, xf[j]),
                                                                     acc[mi * 4 + i][j], 0, 0, 0);
  };

#if MOE256_ROTATE
  // One barrier per K step, placed BEHIND the step's last fragment reads and ahead of its last quarter of products:
  //   quarters 0..2 of step t (their fragments read a quarter ahead) | all reads of stage s waited for; this wave's pieces of
  //   step t + 1 landed (vmcnt) | BARRIER: step t + 1 is complete in LDS and nobody reads stage s any more | issue the
  //   LDS-DMA of step t + 2 into stage s, read the first quarter's fragments of step t + 1 | products of quarter 3.
  // So a step's tiles have a whole step to arrive, the fragment reads of a new stage and the DMA issue hide behind 16
  // products, and the matrix pipe never waits for LDS except inside the barrier itself.
  const int nk = p.K / kBK;
  u32x4_t wf[2][4], xf[2][4];
  issue(0, 0);
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  if (nk > 1) MOE256_ISSUE(1, 1);
  read_w(sm_addr, 0, 0, wf[0]);
  read_x(sm_addr, 0, xf[0]);
  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt & 1;
    const uint32_t base = sm_addr + stage * kStage2;
    read_w(base, 0, 1, wf[1]);
    wait_lgkm<4>();                                    // LDS returns in order: quarter 0's eight fragments are in
    MOE256_PRIO(1);
    products(0, wf[0], xf[0]);
    MOE256_PRIO(0);
    MOE256_PINNED();
    read_w(base, 1, 0, wf[0]);
    read_x(base, 1, xf[1]);
    wait_lgkm<8>();
    MOE256_PRIO(1);
    products(1, wf[1], xf[0]);
    MOE256_PRIO(0);
    MOE256_PINNED();
    read_w(base, 1, 1, wf[1]);
    wait_lgkm<4>();
    MOE256_PRIO(1);
    products(0, wf[0], xf[1]);
    MOE256_PRIO(0);
    MOE256_PINNED();
    wait_lgkm<0>();                                    // every fragment of this stage is in registers
    if (kt + 1 < nk) {
      wait_vm<0>();                                    // this wave's pieces of step kt + 1 have landed
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < nk) MOE256_ISSUE(kt + 2, stage);
      const uint32_t nbase = sm_addr + (stage ^ 1) * kStage2;
      read_w(nbase, 0, 0, wf[0]);
      read_x(nbase, 0, xf[0]);
    }
    MOE256_PRIO(1);
    products(1, wf[1], xf[1]);
    MOE256_PRIO(0);
    MOE256_PINNED();
  }

#else
  // The plain form: issue the next step's tiles, read and multiply this step's, wait, barrier.
  // MOE256_L2_AHEAD (experiment, default 0 = off; measured 5-7 % SLOWER at 1 / 2 / 3 steps ahead, so first-touch latency is
  // not what the walk waits for -- profiles/r04_exp4_moe_gemm_ab.json): a step's tiles have ONE step to get from wherever they live into LDS -- and a weight
  // or token row contributes one new 128-byte line per step, 8 KiB behind the previous one, so every line's first touch
  // pays the full HBM / memory-side-cache latency inside that one step.  One extra 4-byte load per wave and step touches
  // the 64 lines (32 weight rows, 32 token rows: the rows this wave's LDS-DMA moves) that the DMA of `AHEAD` steps later
  // will ask for, so that they wait in L2 by then; its result is never used, only kept alive until the counted wait
  // of the next step has certainly retired it.
  const int nk = p.K / kBK;
#if MOE256_L2_AHEAD
  const uint16_t* pf_row;
  {
    const int prow = wid * 32 + (lane & 31);             // the rows (wid * 4 + u) * 8 + dr of this wave's DMA
    if (lane < 32) {
      int wrow;
      if (p.fuse_silu) {
        const int half = prow >> 7, in = prow & 127;
        const int col = nt0 * 128 + half * 64 + (in & 63);
        wrow = (in < 64 ? 0 : p.N) + (col < p.N ? col : p.N - 1);
      } else {
        const int col = nt0 * 256 + prow;
        wrow = col < p.N ? col : p.N - 1;
      }
      pf_row = wbase + static_cast<int64_t>(wrow) * p.w_row_stride;
    } else {
      const int sid = p.sorted_ids[rb * kT2 + prow];
      pf_row = p.a + static_cast<int64_t>(sid < p.num_valid ? sid / p.topk_div : 0) * p.a_stride;
    }
  }
  uint32_t pf_prev = 0, pf_cur = 0;
  auto touch = [&](int kt) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(pf_row + static_cast<int64_t>(kt) * kBK) : "memory");
    return v;
  };
  if (MOE256_L2_AHEAD < nk) pf_cur = touch(MOE256_L2_AHEAD);      // (step 1's lines are asked for by the DMA right below)
#endif
  issue(0, 0);
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt & 1;
    if (kt + 1 < nk) MOE256_ISSUE(kt + 1, stage ^ 1);
#if MOE256_L2_AHEAD
    pf_prev = pf_cur;
    {
      const int ahead = kt + 1 + MOE256_L2_AHEAD < nk ? kt + 1 + MOE256_L2_AHEAD : nk - 1;   // (clamped: re-touches a resident line)
      pf_cur = touch(ahead);
    }
#endif
    const uint32_t base = sm_addr + stage * kStage2;
    u32x4_t wf[2][4], xf[2][4];
    read_w(base, 0, 0, wf[0]);
    read_x(base, 0, xf[0]);
    read_w(base, 0, 1, wf[1]);
    wait_lgkm<4>();
    products(0, wf[0], xf[0]);
    MOE256_PINNED();
    read_w(base, 1, 0, wf[0]);
    read_x(base, 1, xf[1]);
    wait_lgkm<8>();
    products(1, wf[1], xf[0]);
    MOE256_PINNED();
    read_w(base, 1, 1, wf[1]);
    wait_lgkm<4>();
    products(0, wf[0], xf[1]);
    MOE256_PINNED();
    wait_lgkm<0>();
    products(1, wf[1], xf[1]);
#if MOE256_L2_AHEAD
    wait_vm<1>();                                      // in order: everything older than this step's touch, i.e. the DMA of step kt + 1
    asm volatile("" ::"v"(pf_prev));                   // the touch of the step before has certainly returned: its register is free
#else
    wait_vm<0>();
#endif
    __builtin_amdgcn_s_barrier();
  }
#if MOE256_L2_AHEAD
  wait_vm<0>();
  asm volatile("" ::"v"(pf_cur));
#endif

#endif
  // ---- epilogue: lane holds C^T[n = 16 i + 4 g + r][m = 16 j + r16] ----
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = wc * 64 + j * 16 + r16;
    const int sid = p.sorted_ids[rb * kT2 + row];
    if (sid >= p.num_valid) continue;
    const float scale = p.row_scale ? p.row_scale[sid] : 1.f;
    if (p.fuse_silu) {
      // weight tiles 0..3 = gate columns [0, 64) of this wave's half, tiles 4..7 = the matching up columns
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = nt0 * 128 + wr * 64 + i * 16 + g * 4;
        if (col >= p.N) continue;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gb = rbf(acc[i][j][r]);
          const float sl = rbf(gb / (1.0f + expf(-gb)));
          o[r] = sl * rbf(acc[i + 4][j][r]);
        }
        uint2 w2;
        w2.x = pack_bf2(o[0], o[1]); w2.y = pack_bf2(o[2], o[3]);
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.c) + static_cast<int64_t>(sid) * p.c_stride + col) = w2;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int col = nt0 * 256 + wr * 128 + i * 16 + g * 4;
        if (col >= p.N) continue;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[i][j][r];
          o[r] = p.row_scale ? (p.round_before_scale ? rbf(v) : v) * scale : v;
        }
        if (p.out_f32) {
          *reinterpret_cast<f32x4_t*>(static_cast<float*>(p.c) + static_cast<int64_t>(sid) * p.c_stride + col) = f32x4_t{o[0], o[1], o[2], o[3]};
        } else {
          uint2 w2;
          w2.x = pack_bf2(o[0], o[1]); w2.y = pack_bf2(o[2], o[3]);
          *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.c) + static_cast<int64_t>(sid) * p.c_stride + col) = w2;
        }
      }
    }
  }
}

}  // namespace

extern "C" {

int sgl_amd_moe_tiled_gemm_block_m(void) { return kBM; }

int sgl_amd_moe_tiled_gemm(const void* a, const void* w, void* c, const int32_t* sorted_token_ids, const int32_t* expert_ids,
                           const int32_t* num_tokens_post_padded, const float* topk_weights, int mul_routed_weight,
                           int round_before_scale, int top_k_div, int64_t num_valid_ids, int64_t N, int64_t K,
                           int64_t a_row_stride, int64_t w_row_stride, int64_t w_expert_stride, int64_t c_row_stride,
                           int64_t max_m_blocks, int fuse_silu, int out_f32, void* stream) {
  return sgl_amd_moe_tiled_gemm_ex(a, w, c, sorted_token_ids, expert_ids, num_tokens_post_padded, topk_weights, mul_routed_weight,
                                   round_before_scale, top_k_div, num_valid_ids, N, K, a_row_stride, w_row_stride, w_expert_stride,
                                   c_row_stride, max_m_blocks, fuse_silu, out_f32, kBM, kBM, stream);
}

int sgl_amd_moe_tiled_gemm_ex(const void* a, const void* w, void* c, const int32_t* sorted_token_ids, const int32_t* expert_ids,
                              const int32_t* num_tokens_post_padded, const float* topk_weights, int mul_routed_weight,
                              int round_before_scale, int top_k_div, int64_t num_valid_ids, int64_t N, int64_t K,
                              int64_t a_row_stride, int64_t w_row_stride, int64_t w_expert_stride, int64_t c_row_stride,
                              int64_t max_m_blocks, int fuse_silu, int out_f32, int align_block_m, int tile_rows, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG((align_block_m == 128 || align_block_m == 256) && (tile_rows == 128 || tile_rows == 256) && tile_rows <= align_block_m,
                "moe_tiled_gemm: align_block_m / tile_rows must be 128 or 256, tile_rows <= align_block_m (got %d / %d)", align_block_m, tile_rows);
  SGL_CHECK_ARG(N > 0 && N % 4 == 0 && K >= kBK && K % kBK == 0, "moe_tiled_gemm: need N %% 4 == 0 and K %% %d == 0 (got N=%lld K=%lld)",
                kBK, (long long)N, (long long)K);
  SGL_CHECK_ARG(!fuse_silu || N % 32 == 0, "moe_tiled_gemm: the silu form needs N %% 32 == 0");
  SGL_CHECK_ARG(a_row_stride % 8 == 0 && w_row_stride % 8 == 0 && c_row_stride % 4 == 0,
                "moe_tiled_gemm: row strides must keep 16-byte (a, w) / 8-byte (c) alignment");
  SGL_CHECK_ARG(top_k_div >= 1 && max_m_blocks <= 65535, "moe_tiled_gemm: bad top_k_div / too many row blocks");
  SGL_CHECK_ARG(!mul_routed_weight || topk_weights, "moe_tiled_gemm: mul_routed_weight needs topk_weights");
  SGL_CHECK_ARG(!(fuse_silu && out_f32), "moe_tiled_gemm: the silu form writes bf16");
  if (max_m_blocks == 0 || num_valid_ids == 0) return 0;
  TiledParams p{};
  p.a = static_cast<const uint16_t*>(a); p.w = static_cast<const uint16_t*>(w); p.c = c;
  p.sorted_ids = sorted_token_ids; p.expert_ids = expert_ids; p.num_post_pad = num_tokens_post_padded;
  p.row_scale = mul_routed_weight ? topk_weights : nullptr;
  p.a_stride = a_row_stride; p.w_row_stride = w_row_stride; p.w_expert_stride = w_expert_stride; p.c_stride = c_row_stride;
  p.num_valid = static_cast<int>(num_valid_ids); p.N = static_cast<int>(N); p.K = static_cast<int>(K); p.topk_div = top_k_div;
  p.fuse_silu = fuse_silu; p.out_f32 = out_f32; p.round_before_scale = round_before_scale;
  if (tile_rows == 256) {
    // one-dimensional grid of patches: 4 row blocks x 8 column tiles each, eight patches (one per XCD) per 256 workgroups
    const int cols = fuse_silu ? 128 : 256;
    constexpr int GM = MOE256_GM, GN = 32 / GM;
    const int64_t nt = ((N + cols - 1) / cols + GN - 1) / GN * GN;
    const int64_t patches = ((max_m_blocks + GM - 1) / GM) * (nt / GN);
    const int64_t wgs = (patches + 7) / 8 * 256;
    SGL_CHECK_ARG(wgs <= 0x7fffffffLL, "moe_tiled_gemm: too many tiles");
    p.n_tiles = static_cast<int>(nt);
    hipLaunchKernelGGL(moe_gemm256_kernel, dim3(static_cast<unsigned>(wgs)), dim3(kThreads2), 0, as_stream(stream), p);
    SGL_CHECK_LAUNCH("moe_tiled_gemm(256)");
    return 0;
  }
  p.eid_shift = align_block_m == 256 ? 1 : 0;
  const int64_t m_blocks = max_m_blocks << p.eid_shift;      // 128-row tiles over the alignment's row blocks
  SGL_CHECK_ARG(m_blocks <= 65535, "moe_tiled_gemm: too many row blocks");
  const int cols_per_tile = fuse_silu ? 64 : kBN;
  dim3 grid(static_cast<unsigned>((N + cols_per_tile - 1) / cols_per_tile), static_cast<unsigned>(m_blocks));
  hipLaunchKernelGGL(moe_tiled_gemm_kernel, grid, dim3(kThreads), 0, as_stream(stream), p);
  SGL_CHECK_LAUNCH("moe_tiled_gemm");
  return 0;
}

}  // extern "C"

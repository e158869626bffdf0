// Weight-streaming "skinny" GEMM for gfx950:  y[M, N] = x[M, K] . W[N, K]^T  with M <= 64
// rows per row block, plus its grouped (mixture-of-experts) form.
//
// Replaces (reference, /root/reference/python/sglang):
//   kernels/ops/moe/fused_moe_triton_kernels.py:324 fused_moe_kernel, :771 invoke_fused_moe_kernel
//     (grouped GEMM over sorted_token_ids / expert_ids, optional router-weight multiply),
//   srt/layers/moe/moe_runner/triton_utils/fused_moe.py:457 _fused_moe_kernel_sequence,
//   and, for decode batches, the torch / hipBLASLt matmul behind
//   srt/layers/linear.py:1596-1660 (UnquantizedLinearMethod.apply) and
//   srt/layers/activation.py:130 SiluAndMul when fused into the gate_up projection.
// Oracle: F.linear / srt/layers/moe/fused_moe_native.py:61-164.
//
// At decode every weight byte is used once per step, so the kernel is an HBM stream
// of W with the matrix cores riding along (MFMA 16x16x32 bf16, fp32 accumulate):
//   * the product is computed transposed, C^T = W . x^T: a wave owns 16 rows of W
//     (16 output columns) and streams them straight from HBM into VGPRs in the MFMA
//     A-operand layout (lane = row, 16 bytes of K each) -- no LDS round trip for the
//     operand that is read exactly once; the next K-chunk's 8 loads per lane are in
//     flight while the current chunk is multiplied;
//   * the activation rows (<= 64, shared by the 4 waves of the workgroup) are staged
//     through a double-buffered, XOR-swizzled LDS image and read back as B operands
//     with conflict-free ds_read_b128;
//   * small-N projections do not fill 256 CUs with full-K tiles, so K is split over
//     gridDim.z; partial tiles go to fp32 slabs and the last-arriving workgroup of a
//     tile (agent-scope release / acquire around one atomic ticket) sums the slabs in
//     split order -- deterministic -- and writes bf16;
//   * rows can be gathered / scattered through sorted_token_ids (grouped GEMM), scaled
//     by the router weight, and the gate/up halves can be combined as silu(g)*u in the
//     epilogue with the same bf16 rounding points as the unfused torch ops.
#include <cstdlib>
#include "common.hpp"
#include "sglang_amd.h"

using namespace sgl_amd;

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = 4;
constexpr int kKC = 128;         // K elements per chunk
constexpr int kKSteps = kKC / 32;
constexpr int kSlotsPerRow = kKC / 8;   // 16-byte slots per staged activation row (16)

__device__ __forceinline__ bf16x8_t as_frag(const U4& v) { return __builtin_bit_cast(bf16x8_t, v); }

struct SkinnyParams {
  const uint16_t* x;            // activations [rows, K]
  const uint16_t* w;            // weights [E, w_rows, K]
  const uint16_t* bias;         // optional [N] (dense only)
  void* y;                      // [out_rows, N] bf16 (or fp32 when out_f32)
  float* slabs;                 // split-K partial tiles (fragment-major fp32)
  const int32_t* sorted_ids;    // grouped: [>= row_blocks*ROWS] flat (token, k) pair ids, pad >= numel
  const int32_t* expert_ids;    // grouped: [row_blocks]
  const int32_t* num_post_pad;  // grouped: [1]
  const float* row_scale;       // grouped: optional [numel] router weights
  int64_t x_stride, w_stride, w_expert_stride, y_stride;
  int M;                        // dense: rows of x;  grouped: numel (valid pair ids are < M)
  int N;                        // output columns
  int K;
  int topk_div;                 // grouped: source row = id / topk_div
  int splits;
  int out_f32;
  int round_before_scale;       // grouped: round the accumulator to bf16 before the router weight
};

// Columns per workgroup: each of the 4 waves owns NTW 16-column tiles (FUSE: one gate + one up tile).
template <int NTW, bool FUSE>
struct Geo {
  static constexpr int kColsPerWave = FUSE ? 16 : 16 * NTW;
  static constexpr int kBN = kWavesPerBlock * kColsPerWave;
};

// ---- epilogue shared by the single-pass kernel and the split-K reduce kernel ---------------
// lane holds C[row = 16 mt + r16][col = tile*BN + wid*colsPerWave + 16 nt + 4 g + r]
template <int MT, int NTW, bool FUSE, bool GROUPED>
__device__ __forceinline__ void epilogue(const SkinnyParams& p, f32x4_t (&acc)[NTW][MT], int tile_n, int mb,
                                         int wid, int r16, int g) {
  constexpr int ROWS = 16 * MT;
  constexpr int NOUT = FUSE ? 1 : NTW;
  using G = Geo<NTW, FUSE>;
#pragma unroll
  for (int nt = 0; nt < NOUT; ++nt) {
    const int col0 = tile_n * G::kBN + wid * G::kColsPerWave + nt * 16 + g * 4;
    if (col0 >= p.N) continue;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (!GROUPED && p.bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (col0 + r < p.N) bias4[r] = bf2f(p.bias[col0 + r]);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 16 + r16;
      int64_t dst;
      float scale = 1.f;
      bool ok;
      if (GROUPED) {
        const int id = p.sorted_ids[mb * ROWS + row];
        ok = id < p.M;
        dst = id;
        if (ok && p.row_scale) scale = p.row_scale[id];
      } else {
        dst = mb * ROWS + row;
        ok = dst < p.M;
      }
      if (!ok) continue;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v;
        if (FUSE) {
          // torch: gate_up = linear(x) (bf16) -> silu(gate) (bf16) -> * up (bf16)
          const float gt = rbf(acc[0][mt][r]);
          const float up = rbf(acc[NTW - 1][mt][r]);
          const float sl = rbf(gt / (1.0f + expf(-gt)));
          v = sl * up;
        } else {
          v = acc[nt][mt][r] + bias4[r];
        }
        if (GROUPED && p.row_scale) {
          if (p.round_before_scale) v = rbf(v);
          v *= scale;
        }
        o[r] = v;
      }
      if (p.out_f32) {
        float* yp = static_cast<float*>(p.y) + dst * p.y_stride + col0;
        if (col0 + 3 < p.N) *reinterpret_cast<float4*>(yp) = make_float4(o[0], o[1], o[2], o[3]);
        else for (int r = 0; r < 4; ++r) if (col0 + r < p.N) yp[r] = o[r];
      } else {
        uint16_t* yp = static_cast<uint16_t*>(p.y) + dst * p.y_stride + col0;
        if (col0 + 3 < p.N) {
          uint2 w2;
          w2.x = pack_bf2(o[0], o[1]);
          w2.y = pack_bf2(o[2], o[3]);
          *reinterpret_cast<uint2*>(yp) = w2;
        } else {
          for (int r = 0; r < 4; ++r) if (col0 + r < p.N) yp[r] = f2bf(o[r]);
        }
      }
    }
  }
}

// ---- main kernel: MT = 16-row activation tiles, NTW = column tiles per wave -----------------
template <int MT, int NTW, bool FUSE, bool GROUPED>
__global__ __launch_bounds__(kThreads, (MT * NTW <= 4) ? 4 : 3) void skinny_gemm_kernel(SkinnyParams p) {
  constexpr int ROWS = 16 * MT;
  constexpr int XL = MT;                  // activation 16-byte loads per thread per chunk
  using G = Geo<NTW, FUSE>;
  __shared__ U4 xs[2][ROWS * kSlotsPerRow];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int tile_n = blockIdx.x, mb = blockIdx.y, split = blockIdx.z;

  const uint16_t* wbase = p.w;
  if (GROUPED) {
    if (mb * ROWS >= p.num_post_pad[0]) return;
    const int e = p.expert_ids[mb];
    if (e < 0) return;                    // filtered expert (EP): rows stay untouched
    wbase += static_cast<int64_t>(e) * p.w_expert_stride;
  }

  // ---- K range of this split -------------------------------------------------
  const int nchunks = (p.K + kKC - 1) / kKC;
  const int cb = static_cast<int>(static_cast<int64_t>(split) * nchunks / p.splits);
  const int ce = static_cast<int>(static_cast<int64_t>(split + 1) * nchunks / p.splits);

  // ---- activation staging roles: thread -> (row = q / 16, slot = q % 16), q = tid + 256 j ----
  const uint16_t* xrow[XL];
  bool xvalid[XL];
  int xdst[XL];
#pragma unroll
  for (int j = 0; j < XL; ++j) {
    const int q = tid + kThreads * j;
    const int row = q / kSlotsPerRow, slot = q % kSlotsPerRow;
    int64_t src;
    bool ok;
    if (GROUPED) {
      const int id = p.sorted_ids[mb * ROWS + row];
      ok = id < p.M;
      src = ok ? id / p.topk_div : 0;
    } else {
      const int m = mb * ROWS + row;
      ok = m < p.M;
      src = ok ? m : 0;
    }
    xrow[j] = p.x + src * p.x_stride;
    xvalid[j] = ok;
    xdst[j] = row * kSlotsPerRow + ((slot ^ row) & 15);
  }
  const int my_slot_k = (tid % kSlotsPerRow) * 8;   // same slot for every j (256 % 16 == 0)

  // ---- weight rows of this wave --------------------------------------------------
  const uint16_t* wrow[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    int wr;
    if (FUSE) {
      const int n_col = tile_n * G::kBN + wid * 16 + r16;
      wr = (n_col < p.N ? n_col : p.N - 1) + (nt == NTW - 1 ? p.N : 0);   // gate rows [0,N), up rows [N,2N)
    } else {
      const int n_col = tile_n * G::kBN + wid * G::kColsPerWave + nt * 16 + r16;
      wr = n_col < p.N ? n_col : p.N - 1;
    }
    wrow[nt] = wbase + static_cast<int64_t>(wr) * p.w_stride + g * 8;
  }

  f32x4_t acc[NTW][MT];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  U4 wa[NTW][kKSteps], wb[NTW][kKSteps], xr[XL];
  const int k_last8 = p.K - 8;

  auto load_w = [&](int c, U4 (&dst)[NTW][kKSteps]) {
    const int k0 = c * kKC;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int kk = 0; kk < kKSteps; ++kk) {
        int k = k0 + kk * 32;               // + g*8 is folded into wrow
        if (k + g * 8 > k_last8) k = k_last8 - g * 8;   // tail: stay inside the row (x is zero there)
        dst[nt][kk] = ld16(wrow[nt] + k);
      }
  };
  auto load_x = [&](int c) {
    int k = c * kKC + my_slot_k;
    const bool in_k = k <= k_last8;
    if (!in_k) k = k_last8;
#pragma unroll
    for (int j = 0; j < XL; ++j) {
      U4 v = ld16(xrow[j] + k);
      if (!(xvalid[j] && in_k)) v = U4{0u, 0u, 0u, 0u};
      xr[j] = v;
    }
  };
  auto store_x = [&](int stage) {
#pragma unroll
    for (int j = 0; j < XL; ++j) xs[stage][xdst[j]] = xr[j];
  };
  auto compute = [&](int stage, const U4 (&wreg)[NTW][kKSteps]) {
#pragma unroll
    for (int kk = 0; kk < kKSteps; ++kk) {
      const int sw = ((kk * 4 + g) ^ r16) & 15;
      U4 xf[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[mt] = xs[stage][(mt * 16 + r16) * kSlotsPerRow + sw];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wreg[nt][kk]), as_frag(xf[mt]),
                                                                acc[nt][mt], 0, 0, 0);
    }
  };

  if (cb < ce) {
    load_x(cb);
    load_w(cb, wa);
    store_x(0);
    __syncthreads();
    // two chunks per trip so the register double buffer (wa / wb) is statically indexed;
    // the "next" chunk index is clamped: the last trip re-loads the final chunk (unused).
    for (int c = cb; c < ce; c += 2) {
      const int c1 = (c + 1 < ce) ? c + 1 : ce - 1;
      load_x(c1);
      load_w(c1, wb);
      compute(0, wa);
      store_x(1);
      __syncthreads();
      if (c + 1 >= ce) break;
      const int c2 = (c + 2 < ce) ? c + 2 : ce - 1;
      load_x(c2);
      load_w(c2, wa);
      compute(1, wb);
      store_x(0);
      __syncthreads();
    }
  }

  if (p.splits > 1) {
    // publish the fp32 partial tile (fragment-major, 1 KiB per wave store); the reduce kernel
    // launched behind this one sums the splits in order and runs the epilogue
    constexpr int kSlabVec = kWavesPerBlock * NTW * MT * 64;   // float4 per slab
    const int tile_id = mb * gridDim.x + tile_n;
    f32x4_t* slab = reinterpret_cast<f32x4_t*>(p.slabs) +
                    (static_cast<int64_t>(tile_id) * p.splits + split) * kSlabVec;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) slab[((wid * NTW + nt) * MT + mt) * 64 + lane] = acc[nt][mt];
    return;
  }
  epilogue<MT, NTW, FUSE, GROUPED>(p, acc, tile_n, mb, wid, r16, g);
}

// ---- split-K reduce: same thread geometry as the main kernel, splits summed in order --------
template <int MT, int NTW, bool FUSE, bool GROUPED>
__global__ __launch_bounds__(kThreads) void skinny_reduce_kernel(SkinnyParams p) {
  constexpr int ROWS = 16 * MT;
  constexpr int kSlabVec = kWavesPerBlock * NTW * MT * 64;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int tile_n = blockIdx.x, mb = blockIdx.y;
  if (GROUPED) {
    if (mb * ROWS >= p.num_post_pad[0]) return;
    if (p.expert_ids[mb] < 0) return;
  }
  const int tile_id = mb * gridDim.x + tile_n;
  const f32x4_t* s0 = reinterpret_cast<const f32x4_t*>(p.slabs) + static_cast<int64_t>(tile_id) * p.splits * kSlabVec;
  f32x4_t acc[NTW][MT];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
      for (int sp = 0; sp < p.splits; ++sp)
        s += s0[static_cast<int64_t>(sp) * kSlabVec + ((wid * NTW + nt) * MT + mt) * 64 + lane];
      acc[nt][mt] = s;
    }
  epilogue<MT, NTW, FUSE, GROUPED>(p, acc, tile_n, mb, wid, r16, g);
}

template <int MT, int NTW, bool FUSE, bool GROUPED>
void launch_one(const SkinnyParams& p, int row_blocks, hipStream_t st) {
  using G = Geo<NTW, FUSE>;
  dim3 grid((p.N + G::kBN - 1) / G::kBN, row_blocks, p.splits);
  hipLaunchKernelGGL((skinny_gemm_kernel<MT, NTW, FUSE, GROUPED>), grid, dim3(kThreads), 0, st, p);
  if (p.splits > 1) {
    grid.z = 1;
    hipLaunchKernelGGL((skinny_reduce_kernel<MT, NTW, FUSE, GROUPED>), grid, dim3(kThreads), 0, st, p);
  }
}

template <int NTW, bool FUSE, bool GROUPED>
void launch(const SkinnyParams& p, int mt, int row_blocks, hipStream_t st) {
  switch (mt) {
    case 1: launch_one<1, NTW, FUSE, GROUPED>(p, row_blocks, st); break;
    case 2: launch_one<2, NTW, FUSE, GROUPED>(p, row_blocks, st); break;
    case 3: launch_one<3, NTW, FUSE, GROUPED>(p, row_blocks, st); break;
    default: launch_one<4, NTW, FUSE, GROUPED>(p, row_blocks, st); break;
  }
}

template <bool GROUPED>
void dispatch(const SkinnyParams& p, int mt, int row_blocks, int fuse_silu, int ntw, hipStream_t st) {
  if (fuse_silu) launch<2, true, GROUPED>(p, mt, row_blocks, st);
  else if (ntw == 2) launch<2, false, GROUPED>(p, mt, row_blocks, st);
  else launch<1, false, GROUPED>(p, mt, row_blocks, st);
}

int bn_of(int fuse_silu, int ntw) { return fuse_silu ? 64 : 64 * ntw; }

int check_common(const char* who, int64_t N, int64_t K, int64_t x_stride, int64_t w_stride, int64_t y_stride,
                 int splits, const void* slabs, int ntw) {
  SGL_CHECK_ARG(N > 0 && K >= 32 && K % 8 == 0, "%s: need N > 0, K >= 32 and K %% 8 == 0 (got N=%lld K=%lld)", who,
                (long long)N, (long long)K);
  SGL_CHECK_ARG(x_stride % 8 == 0 && w_stride % 8 == 0, "%s: x / w row strides must be multiples of 8 elements", who);
  SGL_CHECK_ARG(y_stride % 4 == 0, "%s: y row stride must be a multiple of 4 elements", who);
  SGL_CHECK_ARG(splits >= 1 && splits <= (K + kKC - 1) / kKC, "%s: bad split count %d", who, splits);
  SGL_CHECK_ARG(splits == 1 || slabs, "%s: split-K needs the slab workspace", who);
  SGL_CHECK_ARG(ntw == 1 || ntw == 2, "%s: cols_per_wave_tiles must be 1 or 2", who);
  return 0;
}

}  // namespace

extern "C" {

int sgl_amd_skinny_gemm_max_rows(void) { return 64; }
int sgl_amd_skinny_gemm_chunk(void) { return kKC; }

int64_t sgl_amd_skinny_gemm_slab_floats(int64_t row_blocks, int64_t N, int splits, int fuse_silu, int ntw) {
  const int bn = bn_of(fuse_silu, ntw);
  const int64_t tiles = row_blocks * ((N + bn - 1) / bn);
  return tiles * splits * (kWavesPerBlock * (fuse_silu ? 2 : ntw) * 4 * 64 * 4);
}

int sgl_amd_skinny_gemm(const void* x, const void* w, const void* bias, void* y, int64_t M, int64_t N,
                        int64_t K, int64_t x_row_stride, int64_t w_row_stride, int64_t y_row_stride,
                        int fuse_silu, int tiles_per_wave, int num_k_splits, void* ws_slabs, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(M >= 0 && M <= 64, "skinny_gemm: M=%lld rows (supported: <= 64; larger batches use the library GEMM)", (long long)M);
  if (int rc = check_common("skinny_gemm", N, K, x_row_stride, w_row_stride, y_row_stride, num_k_splits, ws_slabs, tiles_per_wave)) return rc;
  SGL_CHECK_ARG(!(fuse_silu && bias), "skinny_gemm: bias is not supported together with the silu fusion");
  if (M == 0) return 0;
  SkinnyParams p{};
  p.x = static_cast<const uint16_t*>(x);
  p.w = static_cast<const uint16_t*>(w);
  p.bias = static_cast<const uint16_t*>(bias);
  p.y = y;
  p.slabs = static_cast<float*>(ws_slabs);
  p.x_stride = x_row_stride; p.w_stride = w_row_stride; p.w_expert_stride = 0; p.y_stride = y_row_stride;
  p.M = static_cast<int>(M); p.N = static_cast<int>(N); p.K = static_cast<int>(K);
  p.topk_div = 1; p.splits = num_k_splits; p.out_f32 = 0; p.round_before_scale = 0;
  dispatch<false>(p, static_cast<int>((M + 15) / 16), 1, fuse_silu, tiles_per_wave, as_stream(stream));
  SGL_CHECK_LAUNCH("skinny_gemm");
  return 0;
}

int sgl_amd_moe_grouped_gemm(const void* a, const void* w, void* c, const int32_t* sorted_token_ids,
                             const int32_t* expert_ids, const int32_t* num_tokens_post_padded,
                             const float* topk_weights, int mul_routed_weight, int round_before_scale,
                             int top_k_div, int64_t num_valid_ids, int64_t N, int64_t K, int64_t num_experts,
                             int64_t a_row_stride, int64_t w_row_stride, int64_t w_expert_stride,
                             int64_t c_row_stride, int block_m, int64_t max_m_blocks, int fuse_silu, int out_f32,
                             int tiles_per_wave, int num_k_splits, void* ws_slabs, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(block_m == 16 || block_m == 32 || block_m == 48 || block_m == 64,
                "moe_grouped_gemm: block_m=%d (supported: 16/32/48/64, must equal the moe_align block size)", block_m);
  if (int rc = check_common("moe_grouped_gemm", N, K, a_row_stride, w_row_stride, c_row_stride, num_k_splits, ws_slabs, tiles_per_wave)) return rc;
  SGL_CHECK_ARG(top_k_div >= 1 && num_experts >= 1, "moe_grouped_gemm: bad top_k_div / num_experts");
  SGL_CHECK_ARG(!mul_routed_weight || topk_weights, "moe_grouped_gemm: mul_routed_weight needs topk_weights");
  SGL_CHECK_ARG(max_m_blocks <= 65535, "moe_grouped_gemm: too many row blocks (%lld)", (long long)max_m_blocks);
  if (max_m_blocks == 0 || num_valid_ids == 0) return 0;
  SkinnyParams p{};
  p.x = static_cast<const uint16_t*>(a);
  p.w = static_cast<const uint16_t*>(w);
  p.bias = nullptr;
  p.y = c;
  p.slabs = static_cast<float*>(ws_slabs);
  p.sorted_ids = sorted_token_ids; p.expert_ids = expert_ids; p.num_post_pad = num_tokens_post_padded;
  p.row_scale = mul_routed_weight ? topk_weights : nullptr;
  p.x_stride = a_row_stride; p.w_stride = w_row_stride; p.w_expert_stride = w_expert_stride; p.y_stride = c_row_stride;
  p.M = static_cast<int>(num_valid_ids); p.N = static_cast<int>(N); p.K = static_cast<int>(K);
  p.topk_div = top_k_div; p.splits = num_k_splits; p.out_f32 = out_f32; p.round_before_scale = round_before_scale;
  dispatch<true>(p, block_m / 16, static_cast<int>(max_m_blocks), fuse_silu, tiles_per_wave, as_stream(stream));
  SGL_CHECK_LAUNCH("moe_grouped_gemm");
  return 0;
}

}  // extern "C"

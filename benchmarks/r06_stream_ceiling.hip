// Round 6, experiment 1: what does the L2 -> LDS side of a CU cost a weight stream?
//
// The decode projections are LDS-DMA weight streams (csrc/wstream_gemm.hip).  Two questions, both asked by the round-5 verdict
// ("measure the L2 -> LDS cost instead of estimating it"):
//   (1) ceiling: how fast does a pure LDS-DMA stream run as a function of waves per workgroup, ring depth, nt policy and
//       access pattern (contiguous per wave vs 16-row tiles of a row-major matrix)?
//   (2) a workgroup that owns a 16-row tile over the WHOLE K (no split-K partials, no combine launch) stages 4 bytes of
//       L2-resident activations per weight byte at 64 rows: what does that do to the stream?
// Not part of the library: a standalone gfx950 program.   hipcc -O3 --offload-arch=gfx950 benchmarks/r06_stream_ceiling.hip -o /tmp/sc
//
// A wave owns a private ring of PD slots x WP KiB.  Per step it waits for its oldest slot, reads it back (ds_read_b128, folded into
// a checksum), and refills it with WP weight pieces (1 KiB per instruction) + AP activation pieces that land in a 1 KiB scratch
// (the bandwidth is what is measured; the bytes are not used).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                             \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(1);                                                                          \
    }                                                                                        \
  } while (0)

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) unsigned char* lds_bytes_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct Params {
  const unsigned char* w;    // this launch's weight copy
  const unsigned char* act;  // shared activation image (L2 resident), act_chunks x 16 KiB
  unsigned* sink;
  int S;                     // steps (128-wide K chunks) per wave
  int pattern;               // 0: contiguous region per wave; 1: rows, the waves of a workgroup split K over its tile(s);
                             // 2: rows, every wave owns its own tile(s) over the workgroup's K range
  int stagger;
  int gdiv;                  // pattern 2: workgroups per K split (blockIdx.x % gdiv = tile group, / gdiv = split)
  long rowbytes;             // patterns 1, 2: bytes per weight row
  int act_chunks;
};

template <int AUX>
__device__ __forceinline__ void dma16(const unsigned char* src, lds_ptr_t dst) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)(src), dst, 16, 0, AUX);
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NW, int WP, int AP, int PD, int AUX>
__global__ __launch_bounds__(64 * NW, 1) void stream_kernel(Params p) {
  constexpr int I = WP + AP;
  static_assert((PD - 1) * I < 64, "vmcnt range");
  static_assert(NW * (PD * WP + 1) * 1024 <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NW * (PD * WP + 1) * 1024];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x, G = gridDim.x;
  lds_bytes_t base = (lds_bytes_t)(lds);
  lds_bytes_t ring = base + wid * (PD * WP + 1) * 1024;
  lds_ptr_t scratch = (lds_ptr_t)(ring + PD * WP * 1024);
  const uint32_t ring_addr = (uint32_t)(uintptr_t)(ring);

  const unsigned char* wsrc[WP];
  long cstep;           // bytes between consecutive steps of this wave
  int krange = 0;       // which K range (in units of S chunks) this wave walks
  if (p.pattern == 0) {
#pragma unroll
    for (int j = 0; j < WP; ++j) wsrc[j] = p.w + (static_cast<long>(b) * NW + wid) * p.S * (WP * 1024L) + j * 1024 + lane * 16;
    cstep = WP * 1024L;
  } else if (p.pattern == 1) {
    krange = p.stagger ? (wid + b) % NW : wid;
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const long tile = b + static_cast<long>(j >> 2) * G;
      wsrc[j] = p.w + (tile * 16 + 4 * (j & 3) + (lane >> 4)) * p.rowbytes + static_cast<long>(krange) * p.S * 256 + (lane & 15) * 16;
    }
    cstep = 256;
  } else {
    const int grp = b % p.gdiv, split = b / p.gdiv;
    krange = split;
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const long tile = static_cast<long>(grp) * NW + wid + static_cast<long>(j >> 2) * p.gdiv * NW;
      wsrc[j] = p.w + (tile * 16 + 4 * (j & 3) + (lane >> 4)) * p.rowbytes + static_cast<long>(split) * p.S * 256 + (lane & 15) * 16;
    }
    cstep = 256;
  }
  const int rot = (p.pattern == 2 && p.stagger) ? static_cast<int>((static_cast<long>(b % p.gdiv) * p.S) / p.gdiv) : 0;
  const unsigned char* dummy = p.act + lane * 16;

  auto issue = [&](int c, int slot) {
    if (c < p.S) {
      const int cr = c + rot < p.S ? c + rot : c + rot - p.S;
#pragma unroll
      for (int j = 0; j < WP; ++j) dma16<AUX>(wsrc[j] + cr * cstep, (lds_ptr_t)(ring + (slot * WP + j) * 1024));
      if constexpr (AP > 0) {
        const int kc = (krange * p.S + cr) % p.act_chunks;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
          const int piece = (p.pattern == 1 ? i : wid * AP + i) & 15;
          dma16<0>(p.act + kc * 16384L + piece * 1024 + lane * 16, scratch);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < I; ++i) dma16<0>(dummy, scratch);
    }
  };

  u32x4_t acc = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int u = 0; u < PD; ++u) issue(u, u);
  int slot = 0;
  for (int c = 0; c < p.S; ++c) {
    wait_vm<(PD - 1) * I>();
    u32x4_t v[WP];
#pragma unroll
    for (int j = 0; j < WP; ++j)
      asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(ring_addr + (slot * WP + j) * 1024 + lane * 16));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      asm volatile("" : "+v"(v[j]));
      acc ^= v[j];
    }
    issue(c + PD, slot);
    slot = slot + 1 == PD ? 0 : slot + 1;
  }
  wait_vm<0>();
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) p.sink[0] = acc.x;
}

// Round 6, experiment 6 ("mall" mode): is a weight matrix that a PREVIOUS kernel pulled through the 256 MiB infinity cache
// streamed faster than one that comes from HBM, and can a small side kernel do that pulling while the chip runs a launch that
// leaves HBM idle (the decode layer's four finish launches, ~20 us per layer)?
template <int POLICY>   // 0 default, 1 nt, 2 sc1
__global__ __launch_bounds__(256) void prefetch_kernel(const unsigned char* w, long bytes, unsigned* sink) {
  const long n16 = bytes / 16;
  u32x4_t acc = {0u, 0u, 0u, 0u};
  const long stride = static_cast<long>(gridDim.x) * 256;
  long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    u32x4_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned char* a = w + (i + j * stride) * 16;
      if constexpr (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v[j]) : "v"(a) : "memory");
      else if constexpr (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[j]) : "v"(a) : "memory");
      else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(a) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(v[j])); acc ^= v[j]; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[1] = acc.x;
}

// a launch that keeps the chip "busy" without touching HBM for `ticks` of the 100 MHz counter (a finish launch's shadow)
__global__ __launch_bounds__(256) void idle_kernel(long ticks, unsigned* sink) {
  const uint64_t t0 = __builtin_readcyclecounter();
  (void)t0;
  const uint64_t s0 = wall_clock64();
  while (static_cast<long>(wall_clock64() - s0) < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks < 0) sink[2] = 1;
}

struct Config {
  const char* label;
  int G, S, pattern, stagger, gdiv;
  long rowbytes;
  int nw, wp, ap, pd, aux;
};

typedef void (*kernel_t)(Params);

template <int NW, int WP, int AP, int PD>
kernel_t pick_aux(int aux) {
  return aux ? static_cast<kernel_t>(stream_kernel<NW, WP, AP, PD, 2>) : static_cast<kernel_t>(stream_kernel<NW, WP, AP, PD, 0>);
}

kernel_t pick(const Config& c) {
#define CASE(NW, WP, AP, PD) \
  if (c.nw == NW && c.wp == WP && c.ap == AP && c.pd == PD) return pick_aux<NW, WP, AP, PD>(c.aux)
  CASE(1, 4, 0, 16); CASE(2, 4, 0, 16); CASE(4, 4, 0, 8); CASE(8, 4, 0, 4); CASE(4, 8, 0, 4); CASE(2, 8, 0, 8); CASE(1, 8, 0, 8);
  CASE(8, 4, 0, 2); CASE(8, 4, 0, 3); CASE(4, 4, 0, 4); CASE(4, 4, 0, 2);
  CASE(4, 8, 4, 4); CASE(4, 4, 4, 8); CASE(8, 4, 2, 4);
  CASE(8, 4, 16, 4); CASE(8, 4, 8, 4); CASE(8, 4, 4, 4); CASE(4, 4, 16, 4); CASE(4, 8, 16, 3); CASE(4, 8, 8, 4);
  CASE(8, 4, 16, 2); CASE(8, 4, 16, 3);
#undef CASE
  return nullptr;
}

}  // namespace

int main(int argc, char** argv) {
  const std::string out_path = argc > 1 ? argv[1] : "";
  const long kb256 = 256, R8K = 8192, R28K = 28672;
  std::vector<Config> cfgs = {
      // ---- (1) pure streams, 235 MB (gate_up sized), contiguous per wave
      {"pure_contig_nw1_pd16_nt", 256, 224, 0, 0, 1, 0, 1, 4, 0, 16, 1},
      {"pure_contig_nw2_pd16_nt", 256, 112, 0, 0, 1, 0, 2, 4, 0, 16, 1},
      {"pure_contig_nw4_pd8_nt", 256, 56, 0, 0, 1, 0, 4, 4, 0, 8, 1},
      {"pure_contig_nw4_pd8_default", 256, 56, 0, 0, 1, 0, 4, 4, 0, 8, 0},
      {"pure_contig_nw8_pd4_nt", 256, 28, 0, 0, 1, 0, 8, 4, 0, 4, 1},
      {"pure_contig_nw8_pd3_nt", 256, 28, 0, 0, 1, 0, 8, 4, 0, 3, 1},
      {"pure_contig_nw8_pd2_nt", 256, 28, 0, 0, 1, 0, 8, 4, 0, 2, 1},
      {"pure_contig_nw4w8_pd4_nt", 256, 28, 0, 0, 1, 0, 4, 8, 0, 4, 1},
      {"pure_contig_nw2w8_pd8_nt", 256, 56, 0, 0, 1, 0, 2, 8, 0, 8, 1},
      {"pure_contig_512wg_nw4_pd4_nt", 512, 28, 0, 0, 1, 0, 4, 4, 0, 4, 1},
      // rows of a [28672, 4096] matrix, every wave two tiles (gate_up's real geometry: 224 workgroups x 4 waves), with / without activations
      {"gateup_rows_noact_nt", 224, 32, 2, 1, 224, R8K, 4, 8, 0, 4, 1},
      {"gateup_rows_noact_nostagger_nt", 224, 32, 2, 0, 224, R8K, 4, 8, 0, 4, 1},
      {"gateup_rows_act_nt", 224, 32, 2, 1, 224, R8K, 4, 8, 4, 4, 1},
      {"gateup_rows_act_default", 224, 32, 2, 1, 224, R8K, 4, 8, 4, 4, 0},
      // ---- (2) o_proj [4096, 4096] = 33.5 MB
      {"o_contig_nw8_pd4_nt", 256, 4, 0, 0, 1, 0, 8, 4, 0, 4, 1},
      {"o_contig_nw4_pd8_nt", 256, 8, 0, 0, 1, 0, 4, 4, 0, 8, 1},
      // today's structure: 64 tile groups x 4 K splits, 4 waves x 1 tile, 16 KiB of activations per step and workgroup
      {"o_splitk4_rows_act_nt", 256, 8, 2, 0, 64, R8K, 4, 4, 4, 8, 1},
      {"o_splitk4_rows_noact_nt", 256, 8, 2, 0, 64, R8K, 4, 4, 0, 8, 1},
      // whole K in the workgroup: one tile, 8 waves x 4 chunks; activations 0 / 1 / 2 / 4 bytes per weight byte
      {"o_wholeK_nw8_act0", 256, 4, 1, 0, 1, R8K, 8, 4, 0, 4, 1},
      {"o_wholeK_nw8_act1", 256, 4, 1, 0, 1, R8K, 8, 4, 4, 4, 1},
      {"o_wholeK_nw8_act2", 256, 4, 1, 0, 1, R8K, 8, 4, 8, 4, 1},
      {"o_wholeK_nw8_act4", 256, 4, 1, 0, 1, R8K, 8, 4, 16, 4, 1},
      {"o_wholeK_nw8_act4_stagger", 256, 4, 1, 1, 1, R8K, 8, 4, 16, 4, 1},
      {"o_wholeK_nw4_act4", 256, 8, 1, 0, 1, R8K, 4, 4, 16, 4, 1},
      // two tiles per workgroup (128 workgroups), activations 2 : 1
      {"o_wholeK_2tiles_128wg_act2", 128, 8, 1, 0, 1, R8K, 4, 8, 16, 3, 1},
      // ---- qkv [6144, 4096] = 50.3 MB: rope pairs -> two tiles per workgroup, 192 workgroups
      {"qkv_contig_nw4_pd8_nt", 256, 12, 0, 0, 1, 0, 4, 4, 0, 8, 1},
      {"qkv_wholeK_2tiles_192wg_act2", 192, 8, 1, 0, 1, R8K, 4, 8, 16, 3, 1},
      {"qkv_wholeK_2tiles_192wg_act1", 192, 8, 1, 0, 1, R8K, 4, 8, 8, 4, 1},
      {"qkv_wholeK_2tiles_192wg_act0", 192, 8, 1, 0, 1, R8K, 4, 8, 0, 4, 1},
      // ---- down [4096, 14336] = 117 MB
      {"down_contig_nw8_pd4_nt", 256, 14, 0, 0, 1, 0, 8, 4, 0, 4, 1},
      {"down_splitk_rows_act_nt", 256, 28, 2, 0, 64, R28K, 4, 4, 4, 8, 1},
      {"down_wholeK_nw8_act0", 256, 14, 1, 0, 1, R28K, 8, 4, 0, 4, 1},
      {"down_wholeK_nw8_act1", 256, 14, 1, 0, 1, R28K, 8, 4, 4, 4, 1},
      {"down_wholeK_nw8_act2", 256, 14, 1, 0, 1, R28K, 8, 4, 8, 4, 1},
      {"down_wholeK_nw8_act4", 256, 14, 1, 0, 1, R28K, 8, 4, 16, 4, 1},
      {"down_wholeK_nw8_act4_pd3", 256, 14, 1, 0, 1, R28K, 8, 4, 16, 3, 1},
      {"down_wholeK_nw8_act4_stagger", 256, 14, 1, 1, 1, R28K, 8, 4, 16, 4, 1},
  };
  (void)kb256;

  const long act_bytes = 112L * 16384;   // [64, 14336] bf16, chunk-major
  unsigned char* act;
  unsigned* sink;
  CHECK(hipMalloc(&act, act_bytes));
  CHECK(hipMemset(act, 1, act_bytes));
  CHECK(hipMalloc(&sink, 64));
  const long pool_bytes = 1536L << 20;
  unsigned char* pool;
  CHECK(hipMalloc(&pool, pool_bytes));
  CHECK(hipMemset(pool, 3, pool_bytes));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));

  if (argc > 2 && std::string(argv[2]) == "mall") {
    auto time_graph = [&](auto&& body, int launches) {
      hipGraph_t graph;
      hipGraphExec_t exec;
      CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      body();
      CHECK(hipStreamEndCapture(st, &graph));
      CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      for (int i = 0; i < 2; ++i) CHECK(hipGraphLaunch(exec, st));
      CHECK(hipStreamSynchronize(st));
      float sum = 0.f;
      const int reps = 6;
      for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(e0, st));
        CHECK(hipGraphLaunch(exec, st));
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        sum += ms;
      }
      CHECK(hipGraphExecDestroy(exec));
      CHECK(hipGraphDestroy(graph));
      return sum / reps * 1e3 / launches;
    };
    hipStream_t side;
    CHECK(hipStreamCreate(&side));
    hipEvent_t fork, join;
    CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    std::string js = "{\"what\": \"weight stream from HBM vs after a prefetch through the infinity cache; us per launch\", \"rows\": [";
    bool jf = true;
    const Config mc[] = {{"o_33MB", 256, 4, 0, 0, 1, 0, 8, 4, 0, 4, 1}, {"o_33MB_default", 256, 4, 0, 0, 1, 0, 8, 4, 0, 4, 0},
                         {"down_117MB", 256, 14, 0, 0, 1, 0, 8, 4, 0, 4, 1}, {"down_117MB_default", 256, 14, 0, 0, 1, 0, 8, 4, 0, 4, 0},
                         {"gateup_235MB", 256, 56, 0, 0, 1, 0, 4, 4, 0, 8, 1}, {"gateup_235MB_default", 256, 56, 0, 0, 1, 0, 4, 4, 0, 8, 0}};
    for (const Config& c : mc) {
      kernel_t k = pick(c);
      const long bytes = static_cast<long>(c.G) * c.nw * c.S * c.wp * 1024;
      int copies = static_cast<int>(pool_bytes / bytes);
      if (copies > 24) copies = 24;
      Params p{};
      p.act = act; p.sink = sink; p.S = c.S; p.pattern = 0; p.act_chunks = 32;
      auto launch = [&](int r) { p.w = pool + static_cast<long>(r) * bytes; hipLaunchKernelGGL(k, dim3(c.G), dim3(64 * c.nw), 0, st, p); };
      const double t_hbm = time_graph([&] { for (int r = 0; r < copies; ++r) launch(r); }, copies);
      const double t_res = time_graph([&] { for (int r = 0; r < copies; ++r) launch(0); }, copies);
      std::printf("%-22s %6.1f MB  from HBM %7.2f us   same copy again %7.2f us\n", c.label, bytes / 1e6, t_hbm, t_res);
      char buf[640];
      std::snprintf(buf, sizeof buf, "%s{\"stream\": \"%s\", \"MB\": %.1f, \"hbm_us\": %.2f, \"resident_us\": %.2f", jf ? "" : ", ", c.label, bytes / 1e6, t_hbm, t_res);
      js += buf;
      jf = false;
      // a prefetch launch in front of every stream launch (serial): pf alone, pf + stream
      for (int pol = 0; pol < 3; ++pol) {
        for (int pg : {64, 256, 1024}) {
          auto pf = [&](int r, hipStream_t s) {
            const unsigned char* w = pool + static_cast<long>(r) * bytes;
            if (pol == 0) hipLaunchKernelGGL(prefetch_kernel<0>, dim3(pg), dim3(256), 0, s, w, bytes, sink);
            else if (pol == 1) hipLaunchKernelGGL(prefetch_kernel<1>, dim3(pg), dim3(256), 0, s, w, bytes, sink);
            else hipLaunchKernelGGL(prefetch_kernel<2>, dim3(pg), dim3(256), 0, s, w, bytes, sink);
          };
          const double t_pf = time_graph([&] { for (int r = 0; r < copies; ++r) pf(r, st); }, copies);
          const double t_both = time_graph([&] { for (int r = 0; r < copies; ++r) { pf(r, st); launch(r); } }, copies);
          std::printf("    prefetch policy %d, %4d workgroups: alone %7.2f us, prefetch + stream %7.2f us -> stream after prefetch %7.2f us\n", pol, pg, t_pf,
                      t_both, t_both - t_pf);
          std::snprintf(buf, sizeof buf, ", \"pf_pol%d_wg%d\": {\"prefetch_us\": %.2f, \"both_us\": %.2f}", pol, pg, t_pf, t_both);
          js += buf;
        }
      }
      // concurrency: main = [idle launch of T us, stream(r)], side = prefetch(r) forked before the idle launch, joined after the stream
      for (long idle_us : {5L, 20L}) {
        for (int pg : {32, 64, 128}) {
          auto body = [&](bool with_pf) {
            for (int r = 0; r < copies; ++r) {
              if (with_pf) {
                CHECK(hipEventRecord(fork, st));
                CHECK(hipStreamWaitEvent(side, fork, 0));
                hipLaunchKernelGGL(prefetch_kernel<0>, dim3(pg), dim3(256), 0, side, pool + static_cast<long>(r) * bytes, bytes, sink);
                CHECK(hipEventRecord(join, side));
              }
              hipLaunchKernelGGL(idle_kernel, dim3(256), dim3(256), 0, st, idle_us * 100, sink);
              launch(r);
              if (with_pf) CHECK(hipStreamWaitEvent(st, join, 0));
            }
          };
          const double t0_ = time_graph([&] { body(false); }, copies);
          const double t1_ = time_graph([&] { body(true); }, copies);
          std::printf("    idle %2ld us + stream: %7.2f us; with a %3d-workgroup prefetch running beside it: %7.2f us\n", idle_us, t0_, pg, t1_);
          std::snprintf(buf, sizeof buf, ", \"idle%ld_pfwg%d\": {\"plain_us\": %.2f, \"with_prefetch_us\": %.2f}", idle_us, pg, t0_, t1_);
          js += buf;
        }
      }
      js += "}";
    }
    js += "]}";
    if (!out_path.empty()) {
      FILE* f = std::fopen(out_path.c_str(), "w");
      if (f) { std::fputs(js.c_str(), f); std::fputc('\n', f); std::fclose(f); }
    }
    return 0;
  }

  std::string json = "{\"what\": \"LDS-DMA weight stream: us per launch (hipGraph of rotated copies, boundary included), TB/s of weight bytes\", \"configs\": {";
  bool first = true;
  for (const Config& c : cfgs) {
    kernel_t k = pick(c);
    if (!k) { std::fprintf(stderr, "no kernel for %s\n", c.label); continue; }
    const long bytes = static_cast<long>(c.G) * c.nw * c.S * c.wp * 1024;
    int copies = static_cast<int>(pool_bytes / bytes);
    if (copies > 32) copies = 32;
    if (copies < 2) { std::fprintf(stderr, "%s does not fit\n", c.label); continue; }
    Params p{};
    p.act = act; p.sink = sink; p.S = c.S; p.pattern = c.pattern; p.stagger = c.stagger; p.gdiv = c.gdiv; p.rowbytes = c.rowbytes;
    p.act_chunks = c.rowbytes ? static_cast<int>(c.rowbytes / 256) : 32;
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int r = 0; r < copies; ++r) {
      p.w = pool + static_cast<long>(r) * bytes;
      hipLaunchKernelGGL(k, dim3(c.G), dim3(64 * c.nw), 0, st, p);
    }
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 2; ++i) CHECK(hipGraphLaunch(exec, st));
    CHECK(hipStreamSynchronize(st));
    float best = 1e30f, sum = 0.f;
    const int reps = 6;
    for (int i = 0; i < reps; ++i) {
      CHECK(hipEventRecord(e0, st));
      CHECK(hipGraphLaunch(exec, st));
      CHECK(hipEventRecord(e1, st));
      CHECK(hipStreamSynchronize(st));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      sum += ms;
      if (ms < best) best = ms;
    }
    const double us = sum / reps * 1e3 / copies, us_min = best * 1e3 / copies;
    const double act_per_w = c.wp ? static_cast<double>(c.ap) / c.wp : 0.0;
    std::printf("%-36s %7.1f MB  %7.2f us (min %7.2f)  %5.2f TB/s  act:w %.1f  copies %d\n", c.label, bytes / 1e6, us, us_min, bytes / us / 1e6,
                act_per_w, copies);
    char buf[512];
    std::snprintf(buf, sizeof buf, "%s\"%s\": {\"MB\": %.2f, \"us\": %.2f, \"us_min\": %.2f, \"TBps\": %.3f, \"act_per_weight_byte\": %.1f, \"workgroups\": %d, \"waves\": %d, \"ring_slots\": %d, \"nt\": %d}",
                  first ? "" : ", ", c.label, bytes / 1e6, us, us_min, bytes / us / 1e6, act_per_w, c.G, c.nw, c.pd, c.aux);
    json += buf;
    first = false;
    CHECK(hipGraphExecDestroy(exec));
    CHECK(hipGraphDestroy(graph));
  }
  json += "}}";
  if (!out_path.empty()) {
    FILE* f = std::fopen(out_path.c_str(), "w");
    if (f) { std::fputs(json.c_str(), f); std::fputc('\n', f); std::fclose(f); }
  }
  return 0;
}

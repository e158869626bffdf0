// What does a GEMM -> GEMM boundary cost a weight stream, as a launch boundary and inside ONE persistent launch?
//
// Prototype for the step the decode layer has not taken ("the whole layer as ONE persistent launch whose
// loader waves keep the weight stream running across the hand-offs").  Not part of the library: a standalone gfx950 program.
//
//   hipcc -O3 --offload-arch=gfx950 benchmarks/persistent_stream_proto.hip -o /tmp/proto && /tmp/proto [json path]
//
// A "layer" is four weight matrices streamed once each (Llama-3-8B: qkv 50.3 MB, o 33.6 MB, gate_up 234.9 MB, down 117.4 MB), by
// 256 workgroups of 8 streaming waves (16-byte loads, U in flight per lane; the loaded words are folded into a checksum, the
// stand-in for the MFMA work that a 64-row decode GEMM hides under the stream anyway).  Between two matrices sits what the real
// layer has there -- every workgroup publishes its 2 KiB share of a 512 KiB activation image and then needs a 128 KiB slice of
// everybody's image before it may use the next matrix -- in three forms:
//
//   launches   one launch per matrix (what the decode hipGraph does today, minus the combine launches): the boundary is the
//              kernel boundary; activations written / read with plain accesses
//   barrier    ONE launch for the whole chain; boundary = device-scope (sc1, write-through) stores of the share, their completion,
//              a sense-reversing grid barrier run by a ninth wave, device-scope loads of the slice
//   prefetch   the same, but every lane issues its first U loads of the NEXT matrix before it waits at the barrier: the weights
//              do not depend on the hand-off, so up to 256 x 8 x 64 x U x 16 B are in flight across it
//
// Output: one JSON object with the time per layer of each form, for U = 8 and 16, and the pure-stream floor (one launch, no
// hand-off at all).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                                  \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) {                                                                       \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));      \
      std::exit(1);                                                                               \
    }                                                                                             \
  } while (0)

namespace {

constexpr int kStreamThreads = 512;   // 8 waves stream
constexpr int kThreads = 576;         // + 1 wave that runs the grid barrier
constexpr int kGrid = 256;            // one workgroup per CU, all resident
constexpr int kMaxSeg = 16;
constexpr int kActBytes = 512 * 1024;          // [64, 4096] bf16
constexpr int kSliceBytes = 128 * 1024;
constexpr int kPartBytes = 4 * 1024 * 1024;       // [4 splits, 64, 4096] fp32        // what one workgroup needs of it (a K quarter of all 64 rows)

struct Seg {
  const uint4* w;
  long iters;          // 16-byte loads per lane (multiple of 16)
};
struct Params {
  Seg seg[kMaxSeg];
  int n_seg;
  int handoff;         // 0 none, 1 plain accesses (kernel boundary does the rest), 2 device-scope accesses + grid barrier
  int prefetch;        // issue the next segment's first U loads before the barrier
  int act_mode;        // in-launch forms: 0 no activation traffic (the barrier alone), 1 sc1 stores + sc1 loads, 2 sc1 stores + L2 invalidate + plain loads
  int hier;            // 1: arrivals counted per XCD first (workgroup b runs on XCD b % 8), generation words per XCD
  unsigned* bar;       // flat: [0] arrivals, [1] generation, [2] timed out; hierarchical: 32-word lines, see grid_barrier_hier
  unsigned char* act;  // kActBytes
  unsigned* sink;      // [grid * threads]
  int combine;         // 1: a matrix publishes split-K partials (16 KiB per workgroup, 4 MiB) and a COMBINE phase -- 64 workgroups, one row
                       //    each: 64 KiB of partials in, 8 KiB of normed activations out -- sits between it and the next matrix
  int phase;           // launches + combine: 1 = this launch is a matrix (+ its partials), 2 = this launch is the combine
  unsigned char* part; // kPartBytes
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned fold(const uint4& v) { return v.x ^ v.y ^ v.z ^ v.w; }

__device__ __forceinline__ void grid_barrier(unsigned* bar) {
  // lane 0 of the ninth wave; the generation is read BEFORE arriving
  const unsigned gen = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned old = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old == kGrid - 1) {
    __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    int spins = 0;          // bounded: a workgroup that is not resident must not wedge the device (bar[2] tells the host)
    while (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) {
        __hip_atomic_store(bar + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
}

// a workgroup's share of a matrix is contiguous: [block][iteration][lane] x 16 B (every matrix is below 2 GiB: 32-bit offsets)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t seg_rsrc(const Seg& sg) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(sg.w), 0, static_cast<int>(sg.iters * kGrid * kStreamThreads * 16), 0x00020000);
}
__device__ __forceinline__ int lane_offset(const Seg& sg, int t) {
  return (static_cast<int>(blockIdx.x) * static_cast<int>(sg.iters) * kStreamThreads + t) * 16;
}
__device__ __forceinline__ uint4 ldw(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// arrivals per XCD (line 1 + x), then one arrival per XCD at line 0; the last one bumps every XCD's generation word (line 9 + x)
__device__ __forceinline__ void grid_barrier_hier(unsigned* bar) {
  const int x = blockIdx.x & 7;
  unsigned* cnt = bar + 32 * (1 + x);
  unsigned* gen_p = bar + 32 * (9 + x);
  const unsigned gen = __hip_atomic_load(gen_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kGrid / 8 - 1) {
    __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 7) {
      __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int k = 0; k < 8; ++k) __hip_atomic_fetch_add(bar + 32 * (9 + k), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
  int spins = 0;
  while (__hip_atomic_load(gen_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 22)) {
      __hip_atomic_store(bar + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
}

template <int U>
__global__ __launch_bounds__(kThreads) void stream_chain_kernel(Params p) {
  const int t = threadIdx.x;
  const bool streamer = t < kStreamThreads;
  unsigned acc = 0;
  uint4 pf[U];
  bool have = false;
  const __amdgpu_buffer_rsrc_t act = __builtin_amdgcn_make_buffer_rsrc(p.act, 0, kActBytes, 0x00020000);
  const bool in_launch = p.handoff == 2;
  const __amdgpu_buffer_rsrc_t part = __builtin_amdgcn_make_buffer_rsrc(p.part, 0, kPartBytes, 0x00020000);

  auto boundary = [&]() {        // in-launch only: stores of this workgroup have left the CU, then the grid barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == kStreamThreads) {
      if (p.hier) grid_barrier_hier(p.bar);
      else grid_barrier(p.bar);
    }
    __syncthreads();
  };
  auto combine_phase = [&]() {   // one row per workgroup: its four split-K partial rows in, the normed row out
    if (blockIdx.x < 64 && streamer) {
      u32x4_t pv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int off = blockIdx.x * 65536 + (i * kStreamThreads + t) * 16;
        pv[i] = in_launch ? __builtin_amdgcn_raw_buffer_load_b128(part, off, 0, 16) : *reinterpret_cast<const u32x4_t*>(p.part + off);
      }
      unsigned c = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) c ^= pv[i].x ^ pv[i].y ^ pv[i].z ^ pv[i].w;
      const u32x4_t v = {c, c + 1u, c + 2u, c + 3u};
      const int off = blockIdx.x * 8192 + t * 16;
      if (in_launch) __builtin_amdgcn_raw_buffer_store_b128(v, act, off, 0, 16);
      else *reinterpret_cast<u32x4_t*>(p.act + off) = v;
      acc ^= c;
    }
  };
  if (p.combine && p.phase == 2) {            // a combine launch of the launches form
    combine_phase();
    p.sink[blockIdx.x * kThreads + t] = acc;
    return;
  }

  for (int s = 0; s < p.n_seg; ++s) {
    // ---- what this matrix's GEMM needs of the previous one's output ------------------------------------
    if (streamer && ((p.handoff == 1) || (in_launch && p.act_mode && s > 0))) {
      const int slice = (blockIdx.x & 3) * kSliceBytes;
      if (in_launch && p.act_mode == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // buffer_inv sc1
      // the whole slice in flight at once (16 loads per lane): the exchange is a latency chain otherwise
      constexpr int NL = kSliceBytes / (kStreamThreads * 16);
      u32x4_t xv[NL];
      if (in_launch && p.act_mode == 1) {
#pragma unroll
        for (int i = 0; i < NL; ++i) xv[i] = __builtin_amdgcn_raw_buffer_load_b128(act, slice + (i * kStreamThreads + t) * 16, 0, 16);   // sc1
      } else {
#pragma unroll
        for (int i = 0; i < NL; ++i) xv[i] = *reinterpret_cast<const u32x4_t*>(p.act + slice + (i * kStreamThreads + t) * 16);
      }
#pragma unroll
      for (int i = 0; i < NL; ++i) acc ^= xv[i].x ^ xv[i].y ^ xv[i].z ^ xv[i].w;
    }
    // ---- the weight stream: buffer loads, ONE vector register of address (the lane's offset), the walk in scalar offsets ----
    if (streamer) {
      const long iters = p.seg[s].iters;
      const __amdgpu_buffer_rsrc_t wr = seg_rsrc(p.seg[s]);
      const int voff = lane_offset(p.seg[s], t);
      if (!have) {
#pragma unroll
        for (int u = 0; u < U; ++u) pf[u] = ldw(wr, voff, u * kStreamThreads * 16);
      }
      for (long i = U; i < iters; i += U) {
        const int so = static_cast<int>(i) * kStreamThreads * 16;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc ^= fold(pf[u]);
          pf[u] = ldw(wr, voff, so + u * kStreamThreads * 16);
          __builtin_amdgcn_sched_barrier(0);      // a rolling window: the slot is refilled as soon as it is consumed (vmcnt(U-1) waits)
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= fold(pf[u]);
      have = false;
    }
    // ---- publish this matrix's output share; the boundary --------------------------------------------------
    if (p.combine) {
      if (streamer) {                          // split-K partials: 16 KiB per workgroup
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int off = blockIdx.x * 16384 + (i * kStreamThreads + t) * 16;
          const u32x4_t v = {acc, acc + 1u, acc + 2u, acc + 3u + static_cast<unsigned>(i)};
          if (in_launch) __builtin_amdgcn_raw_buffer_store_b128(v, part, off, 0, 16);
          else *reinterpret_cast<u32x4_t*>(p.part + off) = v;
        }
      }
      if (in_launch) {
        boundary();
        combine_phase();
        if (s + 1 < p.n_seg) boundary();
      }
      continue;
    }
    if (streamer && t < 128 && ((p.handoff == 1) || (in_launch && p.act_mode))) {
      const int off = blockIdx.x * (kActBytes / kGrid) + t * 16;
      const u32x4_t v = {acc, acc + 1u, acc + 2u, acc + 3u};
      if (in_launch) __builtin_amdgcn_raw_buffer_store_b128(v, act, off, 0, 16);               // sc1: write-through
      else *reinterpret_cast<u32x4_t*>(p.act + off) = v;
    }
    if (in_launch && s + 1 < p.n_seg) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                         // the share has left this CU
      if (p.prefetch && streamer) {
        const __amdgpu_buffer_rsrc_t wn = seg_rsrc(p.seg[s + 1]);
        const int vn = lane_offset(p.seg[s + 1], t);
#pragma unroll
        for (int u = 0; u < U; ++u) pf[u] = ldw(wn, vn, u * kStreamThreads * 16);
        have = true;
      }
      __syncthreads();
      if (t == kStreamThreads) {
        if (p.hier) grid_barrier_hier(p.bar);
        else grid_barrier(p.bar);
      }
      __syncthreads();
    }
  }
  p.sink[blockIdx.x * kThreads + t] = acc;
}

struct Timer {
  hipEvent_t a, b;
  Timer() {
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
  }
};

template <int U>
double time_form(hipStream_t st, const std::vector<Seg>& chain, int layers_per_chain, const char* form, unsigned* bar, unsigned char* act,
                 unsigned char* part, unsigned* sink, int reps) {
  Params base{};
  base.bar = bar;
  base.act = act;
  base.part = part;
  base.sink = sink;
  const std::string f = form;
  hipGraph_t graph;
  hipGraphExec_t exec;
  CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  if (f == "launches" || f == "launches_no_handoff" || f == "launches_combine") {
    for (size_t s = 0; s < chain.size(); ++s) {
      Params p = base;
      p.n_seg = 1;
      p.seg[0] = chain[s];
      p.handoff = f == "launches_no_handoff" ? 0 : 1;
      p.combine = f == "launches_combine" ? 1 : 0;
      p.phase = 1;
      hipLaunchKernelGGL(stream_chain_kernel<U>, dim3(kGrid), dim3(kThreads), 0, st, p);
      if (p.combine) {                    // the combine launch: 64 workgroups, as today's combine kernels
        p.phase = 2;
        hipLaunchKernelGGL(stream_chain_kernel<U>, dim3(64), dim3(kThreads), 0, st, p);
      }
    }
  } else {
    // in-launch forms: "floor", or a name built of barrier / prefetch + _act0|_act1|_act2 + optional _hier
    Params p = base;
    p.n_seg = static_cast<int>(chain.size());
    for (size_t s = 0; s < chain.size(); ++s) p.seg[s] = chain[s];
    p.handoff = f == "floor" ? 0 : 2;
    p.prefetch = f.rfind("prefetch", 0) == 0 ? 1 : 0;
    p.act_mode = f.find("_act1") != std::string::npos ? 1 : f.find("_act2") != std::string::npos ? 2 : 0;
    p.hier = f.find("_hier") != std::string::npos ? 1 : 0;
    p.combine = f.find("_combine") != std::string::npos ? 1 : 0;
    hipLaunchKernelGGL(stream_chain_kernel<U>, dim3(kGrid), dim3(kThreads), 0, st, p);
  }
  CHECK(hipStreamEndCapture(st, &graph));
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CHECK(hipGraphLaunch(exec, st));
  CHECK(hipStreamSynchronize(st));
  Timer tm;
  CHECK(hipEventRecord(tm.a, st));
  for (int i = 0; i < reps; ++i) CHECK(hipGraphLaunch(exec, st));
  CHECK(hipEventRecord(tm.b, st));
  CHECK(hipStreamSynchronize(st));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, tm.a, tm.b));
  CHECK(hipGraphExecDestroy(exec));
  CHECK(hipGraphDestroy(graph));
  return ms * 1e3 / (reps * layers_per_chain);      // us per layer
}

}  // namespace

int main(int argc, char** argv) {
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  // Llama-3-8B layer: rows x K of qkv / o / gate_up / down, bf16; sizes rounded to whole (grid x 512 lanes x 16 loads x 16 B) = 32 MiB
  // would be too coarse: round the per-lane count to 16 instead (grid x 512 x 16 B = 2 MiB per load round)
  const long bytes[4] = {6144L * 4096 * 2, 4096L * 4096 * 2, 28672L * 4096 * 2, 4096L * 14336 * 2};
  const long round_bytes = static_cast<long>(kGrid) * kStreamThreads * 16;
  const int layers = 3;                    // three distinct layers per chain: 1.3 GB of weights, past the 256 MiB memory-side cache
  std::vector<Seg> chain;
  double layer_bytes = 0;
  for (int l = 0; l < layers; ++l)
    for (int m = 0; m < 4; ++m) {
      long iters = (bytes[m] + round_bytes - 1) / round_bytes;
      iters = (iters + 15) / 16 * 16;
      void* w;
      CHECK(hipMalloc(&w, iters * round_bytes));
      CHECK(hipMemsetAsync(w, 0x11 * (m + 1) + l, iters * round_bytes, st));
      chain.push_back(Seg{static_cast<const uint4*>(w), iters});
      if (l == 0) layer_bytes += static_cast<double>(iters * round_bytes);
    }
  unsigned* bar;
  unsigned char* act;
  unsigned* sink;
  CHECK(hipMalloc(&bar, 4096));
  CHECK(hipMemsetAsync(bar, 0, 4096, st));
  CHECK(hipMalloc(&act, kActBytes));
  CHECK(hipMemsetAsync(act, 0, kActBytes, st));
  unsigned char* part;
  CHECK(hipMalloc(&part, kPartBytes));
  CHECK(hipMemsetAsync(part, 0, kPartBytes, st));
  CHECK(hipMalloc(&sink, sizeof(unsigned) * kGrid * kThreads));
  CHECK(hipStreamSynchronize(st));

  const char* forms[] = {"launches_no_handoff", "launches", "floor", "barrier_act0", "barrier_act0_hier", "barrier_act1", "barrier_act1_hier",
                         "barrier_act2_hier", "prefetch_act0_hier", "prefetch_act1_hier", "prefetch_act2_hier", "launches_combine", "barrier_act1_hier_combine"};
  const int n_forms = sizeof(forms) / sizeof(forms[0]);
  std::string out = "{\"what\": \"us per layer of four streamed weight matrices (Llama-3-8B shapes), 256 workgroups x 8 streaming waves\", ";
  out += "\"layer_bytes\": " + std::to_string(static_cast<long>(layer_bytes)) + ", \"forms\": {";
  for (int f = 0; f < n_forms; ++f) {       // (U = 16 loads in flight per lane measured the same as 8 in every form: dropped)
    const double us = time_form<8>(st, chain, layers, forms[f], bar, act, part, sink, 20);
    char buf[256];
    std::snprintf(buf, sizeof buf, "%s\"%s_u8\": {\"us_per_layer\": %.2f, \"TB_per_s\": %.3f}", f ? ", " : "", forms[f], us, layer_bytes / us * 1e-6);
    out += buf;
    std::fprintf(stderr, "%-28s %8.2f us / layer  %.3f TB/s\n", forms[f], us, layer_bytes / us * 1e-6);
  }
  unsigned flags[3] = {0, 0, 0};
  CHECK(hipMemcpy(flags, bar, sizeof flags, hipMemcpyDeviceToHost));
  out += "}, \"barrier_timed_out\": " + std::string(flags[2] ? "true" : "false") + "}";
  std::printf("%s\n", out.c_str());
  if (argc > 1) {
    FILE* fp = std::fopen(argv[1], "w");
    if (fp) {
      std::fprintf(fp, "%s\n", out.c_str());
      std::fclose(fp);
    }
  }
  return 0;
}

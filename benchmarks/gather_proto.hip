// How fast can KV rows be gathered, as a function of the contiguous piece a workgroup reads per token?
//
// Prototype for the shared-prefix decode kernel's unit shape (the chunk kernel moves its bytes at ~4 TB/s, "the
// rate the plain decode kernel's 256-byte row gathers reach").  Not part of the library: a standalone gfx950 program.
//
//   hipcc -O3 --offload-arch=gfx950 benchmarks/gather_proto.hip -o /tmp/gather && /tmp/gather [json path]
//
// Pool layout of a token-major bf16 pool, Llama-3-8B: row(slot) = 8 kv heads x 128 dims x 2 B = 2 KiB for K and the same for V.
// One launch reads the rows of `chunks` 128-token chunks (the bench batch: ~127 chunks = 65 MB of K + V per layer) through a slot-id
// table (one dependent hop, as in the kernel).  Every workgroup reads 64 KiB -- 16 x 16-byte loads per lane, all in flight -- cut as
//   seg 256 B : 128 tokens x 1 kv head   (the chunk kernel's unit today)
//   seg 512 B :  64 tokens x 2 kv heads
//   seg 1 KiB :  32 tokens x 4 kv heads
//   seg 2 KiB :  16 tokens x 8 kv heads  (whole rows)
// so the number of workgroups, the bytes per workgroup and the loads in flight are the same and only the contiguous piece per token
// differs.  `contiguous` = the same bytes as one flat stream (no table, no stride): the ceiling of this load structure.
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

constexpr int kThreads = 256;
constexpr int kRowBytes = 2048;       // 8 kv heads x 128 dims x bf16
constexpr int kChunk = 128;           // tokens per chunk
constexpr int kWgBytes = 65536;       // K + V bytes per workgroup

struct Params {
  const unsigned char* k;
  const unsigned char* v;
  const int* slots;      // [chunks * 128]
  int seg;               // bytes per token piece: 256 / 512 / 1024 / 2048; 0 = flat stream
  unsigned* sink;
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kThreads) void gather_kernel(Params p) {
  const int t = threadIdx.x;
  u32x4_t kv[16];
  if (p.seg == 0) {
    const unsigned char* base = (blockIdx.x & 1 ? p.v : p.k) + static_cast<long>(blockIdx.x >> 1) * kWgBytes;
#pragma unroll
    for (int i = 0; i < 16; ++i) kv[i] = *reinterpret_cast<const u32x4_t*>(base + (i * kThreads + t) * 16);
  } else {
    // workgroup -> (chunk, token group, head group): pieces of `seg` bytes, lanes_per_piece lanes each
    const int lanes = p.seg / 16;                         // 16 .. 128
    const int tokens_per_wg = (kWgBytes / 2) / p.seg;     // 128 .. 16 (K half, V half)
    const int head_groups = kRowBytes / p.seg;            // 8 .. 1
    const int groups_per_chunk = kChunk / tokens_per_wg;  // 1 .. 8
    const int unit = blockIdx.x;
    const int hg = unit % head_groups;
    const int tg = (unit / head_groups) % groups_per_chunk;
    const int chunk = unit / (head_groups * groups_per_chunk);
    const int piece_lane = t % lanes, first_tok = t / lanes;     // kThreads / lanes tokens per pass
    const int tok_step = kThreads / lanes;
    int ids[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int tok = tg * tokens_per_wg + first_tok + i * tok_step;
      ids[i] = p.slots[chunk * kChunk + tok];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long off = static_cast<long>(ids[i]) * kRowBytes + hg * p.seg + piece_lane * 16;
      kv[i] = *reinterpret_cast<const u32x4_t*>(p.k + off);
      kv[8 + i] = *reinterpret_cast<const u32x4_t*>(p.v + off);
    }
  }
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc ^= kv[i].x ^ kv[i].y ^ kv[i].z ^ kv[i].w;
  p.sink[blockIdx.x * kThreads + t] = acc;
}

}  // namespace

int main(int argc, char** argv) {
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  const int chunks = 127;                       // 16 256 tokens: the bench batch's unique K + V rows of one layer (65 MB)
  const int pool_slots = 74 * 1024;             // a layer's pool: 64 x 1152 tokens
  const int layers = 8;                         // rotated: 8 x 2 x 151 MB, past the memory-side cache
  std::vector<unsigned char*> kp(layers), vp(layers);
  for (int l = 0; l < layers; ++l) {
    CHECK(hipMalloc(&kp[l], static_cast<size_t>(pool_slots) * kRowBytes));
    CHECK(hipMalloc(&vp[l], static_cast<size_t>(pool_slots) * kRowBytes));
    CHECK(hipMemsetAsync(kp[l], 0x21 + l, static_cast<size_t>(pool_slots) * kRowBytes, st));
    CHECK(hipMemsetAsync(vp[l], 0x42 + l, static_cast<size_t>(pool_slots) * kRowBytes, st));
  }
  // slot pattern of the bench batch: every chunk is 128 consecutive slots (requests allocate their tokens in one piece), the chunks
  // spread over the pool (a request's private tail sits ~1152 slots after the previous request's)
  std::vector<int> slots(chunks * kChunk);
  for (int c = 0; c < chunks; ++c)
    for (int i = 0; i < kChunk; ++i) slots[c * kChunk + i] = (c * 577) % (pool_slots - kChunk) + i;
  std::vector<int> scattered(slots);            // the same rows when nothing is contiguous (a fragmented pool): a permutation of whole rows
  for (size_t i = 0; i < scattered.size(); ++i) scattered[i] = static_cast<int>((static_cast<long>(i) * 7919 + 13) % pool_slots);
  int *d_slots, *d_scattered;
  CHECK(hipMalloc(&d_slots, slots.size() * 4));
  CHECK(hipMalloc(&d_scattered, slots.size() * 4));
  CHECK(hipMemcpy(d_slots, slots.data(), slots.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_scattered, scattered.data(), slots.size() * 4, hipMemcpyHostToDevice));
  const int wgs = chunks * kChunk * kRowBytes * 2 / kWgBytes;      // 1016
  unsigned* sink;
  CHECK(hipMalloc(&sink, sizeof(unsigned) * wgs * kThreads));
  CHECK(hipStreamSynchronize(st));
  const double bytes = static_cast<double>(wgs) * kWgBytes;

  std::string out = "{\"what\": \"gather of the bench batch's unique K + V rows of one layer (127 chunks x 128 tokens x 2 KiB x 2), 1016 workgroups x 64 KiB, "
                    "16 x 16 B loads per lane in flight; us per launch, eight layers' pools rotated\", \"bytes_per_launch\": " +
                    std::to_string(static_cast<long>(bytes)) + ", \"forms\": {";
  const int segs[5] = {0, 256, 512, 1024, 2048};
  bool first = true;
  for (int pattern = 0; pattern < 2; ++pattern) {
    for (int s = 0; s < 5; ++s) {
      if (pattern == 1 && segs[s] == 0) continue;
      hipGraph_t graph;
      hipGraphExec_t exec;
      CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int l = 0; l < layers; ++l) {
        Params p{kp[l], vp[l], pattern ? d_scattered : d_slots, segs[s], sink};
        hipLaunchKernelGGL(gather_kernel, dim3(wgs), dim3(kThreads), 0, st, p);
      }
      CHECK(hipStreamEndCapture(st, &graph));
      CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      for (int i = 0; i < 3; ++i) CHECK(hipGraphLaunch(exec, st));
      CHECK(hipStreamSynchronize(st));
      hipEvent_t a, b;
      CHECK(hipEventCreate(&a));
      CHECK(hipEventCreate(&b));
      const int reps = 20;
      CHECK(hipEventRecord(a, st));
      for (int i = 0; i < reps; ++i) CHECK(hipGraphLaunch(exec, st));
      CHECK(hipEventRecord(b, st));
      CHECK(hipStreamSynchronize(st));
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, a, b));
      const double us = ms * 1e3 / (reps * layers);
      char name[64], buf[256];
      if (segs[s] == 0) std::snprintf(name, sizeof name, "contiguous");
      else std::snprintf(name, sizeof name, "%s_seg%d", pattern ? "scattered_rows" : "chunked_rows", segs[s]);
      std::snprintf(buf, sizeof buf, "%s\"%s\": {\"us_per_launch\": %.2f, \"TB_per_s\": %.3f}", first ? "" : ", ", name, us, bytes / us * 1e-6);
      first = false;
      out += buf;
      std::fprintf(stderr, "%-24s %7.2f us  %.3f TB/s\n", name, us, bytes / us * 1e-6);
      CHECK(hipGraphExecDestroy(exec));
      CHECK(hipGraphDestroy(graph));
    }
  }
  // ---- the floor of a SHORT flat stream: what a launch that reads a weight matrix once can reach at best, by size --------------
  out += "}, \"flat_stream_by_size\": {";
  unsigned char* big;
  const size_t big_bytes = size_t(2200) << 20;
  CHECK(hipMalloc(&big, big_bytes));
  CHECK(hipMemsetAsync(big, 0x5a, big_bytes, st));
  CHECK(hipFree(sink));
  CHECK(hipMalloc(&sink, sizeof(unsigned) * 17000 * kThreads));
  const char* names[5] = {"o_proj_33.6MB", "qkv_proj_50.3MB", "down_proj_117.4MB", "gate_up_proj_234.9MB", "lm_head_1050.7MB"};
  const long sizes[5] = {4096L * 4096 * 2, 6144L * 4096 * 2, 4096L * 14336 * 2, 28672L * 4096 * 2, 128256L * 4096 * 2};
  for (int m = 0; m < 5; ++m) {
    const long sz = sizes[m] / (2 * kWgBytes) * (2 * kWgBytes);
    const int n_wg = static_cast<int>(sz / kWgBytes);
    hipGraph_t graph;
    hipGraphExec_t exec;
    const int launches = 8;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < launches; ++l) {
      const size_t off = (static_cast<size_t>(l) * 271 * (1 << 20)) % (big_bytes - sz) / 65536 * 65536;      // a different 271 MiB-stepped window per launch
      Params p{big + off, big + off + sz / 2, nullptr, 0, sink};
      hipLaunchKernelGGL(gather_kernel, dim3(n_wg), dim3(kThreads), 0, st, p);
    }
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 2; ++i) CHECK(hipGraphLaunch(exec, st));
    CHECK(hipStreamSynchronize(st));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const int reps = 10;
    CHECK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) CHECK(hipGraphLaunch(exec, st));
    CHECK(hipEventRecord(b, st));
    CHECK(hipStreamSynchronize(st));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / (reps * launches);
    char buf[256];
    std::snprintf(buf, sizeof buf, "%s\"%s\": {\"us_per_launch\": %.2f, \"TB_per_s\": %.3f}", m ? ", " : "", names[m], us, sz / us * 1e-6);
    out += buf;
    std::fprintf(stderr, "flat %-22s %8.2f us  %.3f TB/s\n", names[m], us, sz / us * 1e-6);
    CHECK(hipGraphExecDestroy(exec));
    CHECK(hipGraphDestroy(graph));
  }
  out += "}}";
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

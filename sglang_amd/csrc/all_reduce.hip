// One-shot all-reduce over xGMI peer mappings for decode-sized messages, gfx950.
//
// Replaces (reference, /root/reference/python/sglang):
//   kernels/aot/csrc/allreduce/custom_all_reduce_hip.cuh:347-420 (one-shot kernel), :150-236 (flag barriers),
//   :448-590 (IPC handle exchange / buffer registration), srt/distributed/device_communicators/
//   custom_all_reduce.py:182-307 (registration, dispatch, capture), called from
//   srt/distributed/parallel_state.py:648-758,965-1008 (GroupCoordinator.all_reduce).
// Oracle: torch.sum over the ranks' inputs in fp32, rounded once (tests/test_xgmi_all_reduce_gpu.py).
//
// MI355X has no switch: every GPU has a direct xGMI link to each of its 7 peers, so for a message of a few
// hundred KiB the cheapest all-reduce is the one where each rank PULLS the other ranks' copies over those links at
// once and adds them up locally -- one hop, no ring steps, (world-1)/world of the message per link direction:
//   phase 0  copy my input into my registered (IPC-exported, uncached) buffer
//   phase 1  flag barrier: tell every peer "my copy is complete", wait for all of theirs
//   phase 2  every rank reads all copies, adds them in RANK ORDER in fp32 (so all ranks produce identical bits),
//            rounds once to bf16 and -- optionally -- applies the operator that follows a row-parallel projection in
//            the decoder layer: residual add + RMSNorm (layernorm.py:777-826), saving a launch and a round trip
//   phase 3  end barrier: nobody overwrites its buffer for the next call while a peer may still read it
// The flags are monotonically increasing counters kept in device memory (the kernel increments its own), so a
// launch has no host-side state: it can be captured into the decode hipGraph and replayed.
// Flags are written with system-scope stores straight into the PEER's signal block and polled locally.  Everything a
// peer reads is written with SYSTEM-SCOPE (sc0 sc1) stores -- write-through past this device's L2 whatever MTYPE the pages
// carry -- and read with system-scope loads, so that a release is the completion of the data stores (s_waitcnt vmcnt(0)),
// not a write-back of the whole L2, and does not rest on the workspace pages being mapped uncached on both sides
// (they are allocated hipDeviceMallocUncached as well); see flag_barrier.  sgl_amd_xgmi_set_release_fence(1) puts a full
// system-scope release fence in front of every flag as a fallback.
#include <cstddef>
#include <cstring>
#include "common.hpp"
#include "sglang_amd.h"

using namespace sgl_amd;

namespace {

constexpr int kMaxWorld = 8;
constexpr int kMaxBlocks = 256;            // one workgroup per CU at most: a phase is a few dependent round trips per chunk, width hides them
constexpr int kArThreads = 512;
constexpr int64_t kDataOffset = 32768;         // data area starts here (signal block padded to 32 KiB)

struct Signal {
  uint32_t start[kMaxBlocks][kMaxWorld];
  uint32_t end[kMaxBlocks][kMaxWorld];
  uint32_t mid[kMaxBlocks][kMaxWorld];       // two-stage kernel: "my slice of the sum is published"
  uint32_t flag[kMaxBlocks];
  uint32_t timed_out;                        // set when a flag wait gave up (a peer never arrived)
  uint32_t trap_on_timeout;                  // armed by the host once the start-up self-test passed (sgl_amd_xgmi_arm)
  uint32_t release_fence;                    // 1: a system-scope release fence precedes every flag THIS rank sends (the fallback
                                             // protocol; per communicator, read when a launch RUNS: sgl_amd_xgmi_set_release_fence)
};
static_assert(sizeof(Signal) <= kDataOffset, "signal block");

struct Peers {
  unsigned char* base[kMaxWorld];            // every rank's workspace as mapped into THIS process (own = local pointer)
};

struct ArParams {
  Peers peers;
  const uint16_t* inp;          // [rows, hidden] bf16 (contiguous)
  uint16_t* out;                // [rows, hidden]
  uint16_t* residual;           // epilogue 1: [rows, hidden], updated in place
  const uint16_t* norm_w;       // epilogue 1: [hidden]
  int64_t numel;                // rows * hidden, multiple of 8
  int rows, hidden;
  int rank, world;
  int epilogue;                 // 0 none, 1 residual add + RMSNorm
  float eps;
  int ws_bytes;                 // size of every rank's workspace: the buffer range of the system-scope accesses (buffer
                                // descriptors carry 32-bit ranges and offsets: the entry points refuse workspaces of 2 GiB and more)
};

// System-scope 16-byte accesses to a workspace (own or a peer's): buffer instructions with sc0 sc1 -- stores write
// through to memory, loads bypass this device's caches -- at byte offset `off` of the workspace `base` points to.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rsrc(unsigned char* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ U4 ld16_sys(__amdgpu_buffer_rsrc_t r, int64_t off) {
  return __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(off), 0, 17));
}
__device__ __forceinline__ void st16_sys(__amdgpu_buffer_rsrc_t r, int64_t off, const U4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, static_cast<int>(off), 0, 17);
}

// two-stage kernel: the rank that sums (and publishes) the k-th unit -- kUnroll chunks, or a row -- of workgroup `block`
__host__ __device__ __forceinline__ int two_stage_owner(int64_t block, int64_t k, int world) { return static_cast<int>((block + k) % world); }

template <bool ACQUIRE = true>
__device__ __forceinline__ void flag_barrier(const ArParams& p, uint32_t (Signal::*arr)[kMaxBlocks][kMaxWorld], uint32_t flag, bool release_fence) {
  // RELEASE.  What a peer reads from this rank lives in this rank's OWN workspace and was written with system-scope
  // (sc0 sc1) stores: they write through L2 to memory, so publishing them needs no cache write-back -- only their
  // completion.  Every wave waits for its own stores (s_waitcnt vmcnt(0): the wait the memory model prescribes ahead of a
  // system-scope release; inline asm, so the compiler cannot drop it), the workgroup barrier collects the waves, then the
  // flag goes out.  (A system-scope release fence here also writes back every dirty L2 line of the device -- the
  // projection's output, the residual stream: 24 us of a 33 us two-stage launch at 256 rows, 1.5 ms of a TP 4 rank's
  // 5.8 ms decode step.  It stays available as the fallback: release_fence.)
  // (`release_fence`: this rank's own setting, read ONCE per launch at kernel entry into a scalar register -- not by every thread
  // ahead of every barrier)
  if (release_fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int t = threadIdx.x;
  if (t < p.world) {
    Signal* peer = reinterpret_cast<Signal*>(p.peers.base[t]);
    Signal* self = reinterpret_cast<Signal*>(p.peers.base[p.rank]);
    __hip_atomic_store(&(peer->*arr)[blockIdx.x][p.rank], flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // bounded: a peer that never launches (a crashed rank) must not wedge the GPU -- give up after ~seconds and leave
    // a mark the host can read (sgl_amd_xgmi_timed_out).  During the start-up self-test the kernel then runs on (the
    // host compares the result and drops the communicator on all ranks); once armed, an unreduced sum must never
    // pass for a result: the kernel traps, the stream fails, every later call on this rank raises.
    int spins = 0;
    while (__hip_atomic_load(&(self->*arr)[blockIdx.x][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < flag) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1 << 25)) {          // about half a minute: host-side skew between ranks (a GC pause, a page fault storm) is not a crash
        self->timed_out = 1u;
        if (self->trap_on_timeout) __builtin_trap();
        break;
      }
    }
    // ACQUIRE, once per workgroup and barrier (not once per poll): whatever this CU's L1 or this XCD's L2 may hold of
    // the peers' workspaces is dropped before anybody reads them (the reads are system-scope loads as well, so this is
    // belt and braces -- but it is the half of the fence pair that costs little).  The END barrier of a launch is
    // followed by no read of a peer's memory -- it only keeps this rank from overwriting its workspace in the next
    // call while a peer still reads it -- and skips it.
    if constexpr (ACQUIRE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  __syncthreads();
}

// sum of the `world` copies of 8 bf16 at element offset e, in rank order, fp32
template <int WORLD>
__device__ __forceinline__ void gather_sum(const ArParams& p, int64_t e, float (&acc)[8]) {
  U4 v[WORLD];
#pragma unroll
  for (int r = 0; r < WORLD; ++r) v[r] = ld16_sys(ws_rsrc(p.peers.base[r], p.ws_bytes), kDataOffset + e * 2);
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
  for (int r = 0; r < WORLD; ++r) {
    const uint32_t w[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[2 * j] += bf_lo(w[j]);
      acc[2 * j + 1] += bf_hi(w[j]);
    }
  }
}

template <int WORLD>
__global__ __launch_bounds__(kArThreads) void xgmi_one_shot_all_reduce_kernel(ArParams p) {
  __shared__ float scratch[16];
  Signal* self = reinterpret_cast<Signal*>(p.peers.base[p.rank]);
  const uint32_t flag = self->flag[blockIdx.x] + 1;
  const bool rel = __builtin_amdgcn_readfirstlane(static_cast<int>(self->release_fence)) != 0;
  const int64_t nvec = p.numel / 8;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kArThreads;
  // phase 0: my copy
  const __amdgpu_buffer_rsrc_t mine = ws_rsrc(p.peers.base[p.rank], p.ws_bytes);
  if (p.epilogue == 0) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kArThreads + threadIdx.x; i < nvec; i += stride)
      st16_sys(mine, kDataOffset + i * 16, ld16(p.inp + i * 8));
  } else {
    // row-wise ownership: the rows a workgroup will finish are the rows it publishes
    for (int r = blockIdx.x; r < p.rows; r += gridDim.x)
      for (int c = threadIdx.x * 8; c < p.hidden; c += kArThreads * 8)
        st16_sys(mine, kDataOffset + (static_cast<int64_t>(r) * p.hidden + c) * 2, ld16(p.inp + static_cast<int64_t>(r) * p.hidden + c));
  }
  flag_barrier(p, &Signal::start, flag, rel);

  if (p.epilogue == 0) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kArThreads + threadIdx.x; i < nvec; i += stride) {
      float acc[8];
      gather_sum<WORLD>(p, i * 8, acc);
      U4 o;
      o.x = pack_bf2(acc[0], acc[1]); o.y = pack_bf2(acc[2], acc[3]);
      o.z = pack_bf2(acc[4], acc[5]); o.w = pack_bf2(acc[6], acc[7]);
      st16(p.out + i * 8, o);
    }
  } else {
    // h = bf16(sum);  t = h + residual (fp32);  residual <- bf16(t);  out = bf16(t * rsqrt(mean(t^2) + eps) * w)
    constexpr int kMaxVec = 4;                    // hidden <= 512 threads * 8 * 4 = 16384
    for (int r = blockIdx.x; r < p.rows; r += gridDim.x) {
      float t[kMaxVec][8];
      float sq = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxVec; ++k) {
        const int c = (k * kArThreads + threadIdx.x) * 8;
        if (c < p.hidden) {
          const int64_t e = static_cast<int64_t>(r) * p.hidden + c;
          float acc[8];
          gather_sum<WORLD>(p, e, acc);
          const U4 rv = ld16(p.residual + e);
          const uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            t[k][2 * j] = rbf(acc[2 * j]) + bf_lo(rw[j]);
            t[k][2 * j + 1] = rbf(acc[2 * j + 1]) + bf_hi(rw[j]);
          }
          U4 o;
          o.x = pack_bf2(t[k][0], t[k][1]); o.y = pack_bf2(t[k][2], t[k][3]);
          o.z = pack_bf2(t[k][4], t[k][5]); o.w = pack_bf2(t[k][6], t[k][7]);
          st16(p.residual + e, o);
#pragma unroll
          for (int j = 0; j < 8; ++j) sq += t[k][j] * t[k][j];
        }
      }
      sq = block_sum(sq, scratch);
      const float rs = 1.0f / sqrtf(sq / static_cast<float>(p.hidden) + p.eps);
#pragma unroll
      for (int k = 0; k < kMaxVec; ++k) {
        const int c = (k * kArThreads + threadIdx.x) * 8;
        if (c < p.hidden) {
          const U4 wv = ld16(p.norm_w + c);
          const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
          U4 o;
          o.x = pack_bf2((t[k][0] * rs) * bf_lo(ww[0]), (t[k][1] * rs) * bf_hi(ww[0]));
          o.y = pack_bf2((t[k][2] * rs) * bf_lo(ww[1]), (t[k][3] * rs) * bf_hi(ww[1]));
          o.z = pack_bf2((t[k][4] * rs) * bf_lo(ww[2]), (t[k][5] * rs) * bf_hi(ww[2]));
          o.w = pack_bf2((t[k][6] * rs) * bf_lo(ww[3]), (t[k][7] * rs) * bf_hi(ww[3]));
          st16(p.out + static_cast<int64_t>(r) * p.hidden + c, o);
        }
      }
    }
  }
  flag_barrier<false>(p, &Signal::end, flag, rel);
  if (threadIdx.x == 0) self->flag[blockIdx.x] = flag;
}


// ---------------------------------------------------------------------------------------------
// Two-stage all-reduce for prefill-sized messages (custom_all_reduce_hip.cuh:595-652, selected by
// custom_all_reduce.py:260-307 above the one-shot sizes): reduce-scatter, then all-gather, both by PULLING over the
// direct links.  Rank r owns every WORLD-th 8 KiB chunk of a workgroup's share of the message: it reads that chunk
// from all copies, adds in rank order (fp32, one rounding) and publishes the sum in the second half of its workspace;
// after the middle barrier everybody reads every chunk from its owner.  Per link direction 2/world of the message
// instead of the one-shot kernel's whole message; same bits on every rank (each sum is computed once, by its owner).
// Chunk c of kArThreads 16-byte vectors belongs to workgroup c % grid in ALL phases on ALL ranks, so the per-workgroup
// flag rows pair the same data on both sides of a link; its owner is rank (c % grid + c / grid) % world.
// `kUnroll` chunks of a workgroup are in flight together in every phase (a phase is a chain of dependent memory round
// trips per chunk otherwise: 4 MiB took 39 us on local memory before, loopback).
// epilogue 1 (rows x hidden messages, hidden <= kArThreads * 8 * 4): a chunk is a ROW, so that the workgroup that
// gathers a row in the last phase holds all of it and finishes the operator that follows a row-parallel projection --
// residual add + RMSNorm (layernorm.py:777-826), as in the one-shot kernel -- instead of a second launch reading the sum.
template <int WORLD>
__global__ __launch_bounds__(kArThreads) void xgmi_two_stage_all_reduce_kernel(ArParams p, int64_t sums_offset) {
  constexpr int kUnroll = 4;
  __shared__ float scratch[16];
  Signal* self = reinterpret_cast<Signal*>(p.peers.base[p.rank]);
  const uint32_t flag = self->flag[blockIdx.x] + 1;
  const bool rel = __builtin_amdgcn_readfirstlane(static_cast<int>(self->release_fence)) != 0;
  const __amdgpu_buffer_rsrc_t mine = ws_rsrc(p.peers.base[p.rank], p.ws_bytes);   // copies at kDataOffset, sums at sums_offset
  if (p.epilogue == 0) {
    const int64_t nvec = p.numel / 8;
    const int64_t nchunks = (nvec + kArThreads - 1) / kArThreads;
    const int64_t step = static_cast<int64_t>(gridDim.x) * kUnroll;
    // chunk c belongs to workgroup b = (c / kUnroll) % grid; its owner is rank (b + c / (kUnroll * grid)) % world -- the
    // workgroup index is part of it, or a message of one iteration per workgroup (anything up to 8 MiB at the automatic
    // grid) would be summed by rank 0 alone
    for (int64_t c0 = static_cast<int64_t>(blockIdx.x) * kUnroll; c0 < nchunks; c0 += step) {
      U4 v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int64_t i = (c0 + u) * kArThreads + threadIdx.x;
        if (c0 + u < nchunks && i < nvec) v[u] = ld16(p.inp + i * 8);
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int64_t i = (c0 + u) * kArThreads + threadIdx.x;
        if (c0 + u < nchunks && i < nvec) st16_sys(mine, kDataOffset + i * 16, v[u]);
      }
    }
    flag_barrier(p, &Signal::start, flag, rel);
    int64_t k = 0;
    for (int64_t c0 = static_cast<int64_t>(blockIdx.x) * kUnroll; c0 < nchunks; c0 += step, ++k) {
      if (two_stage_owner(blockIdx.x, k, WORLD) != p.rank) continue;
      float acc[kUnroll][8];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int64_t i = (c0 + u) * kArThreads + threadIdx.x;
        if (c0 + u < nchunks && i < nvec) gather_sum<WORLD>(p, i * 8, acc[u]);
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int64_t i = (c0 + u) * kArThreads + threadIdx.x;
        if (c0 + u < nchunks && i < nvec) {
          U4 o;
          o.x = pack_bf2(acc[u][0], acc[u][1]); o.y = pack_bf2(acc[u][2], acc[u][3]);
          o.z = pack_bf2(acc[u][4], acc[u][5]); o.w = pack_bf2(acc[u][6], acc[u][7]);
          st16_sys(mine, sums_offset + i * 16, o);
        }
      }
    }
    flag_barrier(p, &Signal::mid, flag, rel);
    k = 0;
    for (int64_t c0 = static_cast<int64_t>(blockIdx.x) * kUnroll; c0 < nchunks; c0 += step, ++k) {
      const __amdgpu_buffer_rsrc_t src = ws_rsrc(p.peers.base[two_stage_owner(blockIdx.x, k, WORLD)], p.ws_bytes);   // the chunks' owner
      U4 v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int64_t i = (c0 + u) * kArThreads + threadIdx.x;
        if (c0 + u < nchunks && i < nvec) v[u] = ld16_sys(src, sums_offset + i * 16);
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int64_t i = (c0 + u) * kArThreads + threadIdx.x;
        if (c0 + u < nchunks && i < nvec) st16(p.out + i * 8, v[u]);
      }
    }
  } else {
    // rows: workgroup b owns rows b, b + grid, ...; the j-th of them is summed by rank (b + j) % world
    constexpr int kMaxVec = 4;
    for (int r = blockIdx.x; r < p.rows; r += gridDim.x)
      for (int c = threadIdx.x * 8; c < p.hidden; c += kArThreads * 8)
        st16_sys(mine, kDataOffset + (static_cast<int64_t>(r) * p.hidden + c) * 2, ld16(p.inp + static_cast<int64_t>(r) * p.hidden + c));
    flag_barrier(p, &Signal::start, flag, rel);
    int j = 0;
    for (int r = blockIdx.x; r < p.rows; r += gridDim.x, ++j) {
      if (two_stage_owner(blockIdx.x, j, WORLD) != p.rank) continue;
#pragma unroll
      for (int kv = 0; kv < kMaxVec; ++kv) {
        const int c = (kv * kArThreads + threadIdx.x) * 8;
        if (c < p.hidden) {
          const int64_t e = static_cast<int64_t>(r) * p.hidden + c;
          float acc[8];
          gather_sum<WORLD>(p, e, acc);
          U4 o;
          o.x = pack_bf2(acc[0], acc[1]); o.y = pack_bf2(acc[2], acc[3]);
          o.z = pack_bf2(acc[4], acc[5]); o.w = pack_bf2(acc[6], acc[7]);
          st16_sys(mine, sums_offset + e * 2, o);
        }
      }
    }
    flag_barrier(p, &Signal::mid, flag, rel);
    j = 0;
    for (int r = blockIdx.x; r < p.rows; r += gridDim.x, ++j) {
      const __amdgpu_buffer_rsrc_t src = ws_rsrc(p.peers.base[two_stage_owner(blockIdx.x, j, WORLD)], p.ws_bytes);   // the row's owner
      // h = the owner's bf16(sum);  t = h + residual (fp32);  residual <- bf16(t);  out = bf16(t * rsqrt(mean(t^2) + eps) * w)
      float t[kMaxVec][8];
      float sq = 0.f;
#pragma unroll
      for (int kv = 0; kv < kMaxVec; ++kv) {
        const int c = (kv * kArThreads + threadIdx.x) * 8;
        if (c < p.hidden) {
          const int64_t e = static_cast<int64_t>(r) * p.hidden + c;
          const U4 hv = ld16_sys(src, sums_offset + e * 2);
          const U4 rv = ld16(p.residual + e);
          const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w}, rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            t[kv][2 * q] = bf_lo(hw[q]) + bf_lo(rw[q]);
            t[kv][2 * q + 1] = bf_hi(hw[q]) + bf_hi(rw[q]);
          }
          U4 o;
          o.x = pack_bf2(t[kv][0], t[kv][1]); o.y = pack_bf2(t[kv][2], t[kv][3]);
          o.z = pack_bf2(t[kv][4], t[kv][5]); o.w = pack_bf2(t[kv][6], t[kv][7]);
          st16(p.residual + e, o);
#pragma unroll
          for (int q = 0; q < 8; ++q) sq += t[kv][q] * t[kv][q];
        }
      }
      sq = block_sum(sq, scratch);
      const float rs = 1.0f / sqrtf(sq / static_cast<float>(p.hidden) + p.eps);
#pragma unroll
      for (int kv = 0; kv < kMaxVec; ++kv) {
        const int c = (kv * kArThreads + threadIdx.x) * 8;
        if (c < p.hidden) {
          const U4 wv = ld16(p.norm_w + c);
          const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
          U4 o;
          o.x = pack_bf2((t[kv][0] * rs) * bf_lo(ww[0]), (t[kv][1] * rs) * bf_hi(ww[0]));
          o.y = pack_bf2((t[kv][2] * rs) * bf_lo(ww[1]), (t[kv][3] * rs) * bf_hi(ww[1]));
          o.z = pack_bf2((t[kv][4] * rs) * bf_lo(ww[2]), (t[kv][5] * rs) * bf_hi(ww[2]));
          o.w = pack_bf2((t[kv][6] * rs) * bf_lo(ww[3]), (t[kv][7] * rs) * bf_hi(ww[3]));
          st16(p.out + static_cast<int64_t>(r) * p.hidden + c, o);
        }
      }
    }
  }
  flag_barrier<false>(p, &Signal::end, flag, rel);
  if (threadIdx.x == 0) self->flag[blockIdx.x] = flag;
}

// All-gather along the last dimension (the vocab-parallel logits of logits_processor.py:676 inside the decode graph):
// out[row, r * cols + c] = rank r's inp[row, c].  One launch, same flag protocol, same chunk-to-workgroup map.
template <int WORLD>
__global__ __launch_bounds__(kArThreads) void xgmi_all_gather_kernel(ArParams p) {
  Signal* self = reinterpret_cast<Signal*>(p.peers.base[p.rank]);
  const uint32_t flag = self->flag[blockIdx.x] + 1;
  const bool rel = __builtin_amdgcn_readfirstlane(static_cast<int>(self->release_fence)) != 0;
  const int64_t nvec = p.numel / 8;                 // of one rank's shard [rows, hidden]
  const int64_t nchunks = (nvec + kArThreads - 1) / kArThreads;
  const int vpr = p.hidden / 8;                     // vectors per shard row
  const __amdgpu_buffer_rsrc_t mine = ws_rsrc(p.peers.base[p.rank], p.ws_bytes);
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int64_t i = c * kArThreads + threadIdx.x;
    if (i < nvec) st16_sys(mine, kDataOffset + i * 16, ld16(p.inp + i * 8));
  }
  flag_barrier(p, &Signal::start, flag, rel);
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int64_t i = c * kArThreads + threadIdx.x;
    if (i < nvec) {
      const int64_t row = i / vpr;
      const int col = static_cast<int>(i - row * vpr) * 8;
      U4 v[WORLD];
#pragma unroll
      for (int r = 0; r < WORLD; ++r) v[r] = ld16_sys(ws_rsrc(p.peers.base[r], p.ws_bytes), kDataOffset + i * 16);
#pragma unroll
      for (int r = 0; r < WORLD; ++r) st16(p.out + row * (static_cast<int64_t>(p.hidden) * WORLD) + static_cast<int64_t>(r) * p.hidden + col, v[r]);
    }
  }
  flag_barrier<false>(p, &Signal::end, flag, rel);
  if (threadIdx.x == 0) self->flag[blockIdx.x] = flag;
}

}  // namespace

// cap on the automatic workgroup counts (tests with several ranks on ONE GPU: all their spinning workgroups must be
// resident together); explicit num_blocks arguments are not touched
static int g_xgmi_auto_blocks_cap = kMaxBlocks;

static int two_stage_auto_blocks(int64_t rows, int64_t numel, int epilogue) {
  const int64_t units = epilogue ? rows : ((numel / 8 + kArThreads - 1) / kArThreads + 3) / 4;
  int blocks = static_cast<int>(units < kMaxBlocks ? (units < 1 ? 1 : units) : kMaxBlocks);
  if (blocks > g_xgmi_auto_blocks_cap) blocks = g_xgmi_auto_blocks_cap;
  return blocks;
}

extern "C" {

int sgl_amd_xgmi_debug_auto_blocks_cap(int cap) {
  if (cap < 1 || cap > kMaxBlocks) {
    set_last_error("xgmi_debug_auto_blocks_cap: 1..%d", kMaxBlocks);
    return -1;
  }
  g_xgmi_auto_blocks_cap = cap;
  return 0;
}

int sgl_amd_xgmi_set_release_fence(void* workspace, int on) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(workspace, "xgmi_set_release_fence: null workspace");
  const uint32_t v = on ? 1u : 0u;
  hipError_t e = hipMemcpy(static_cast<unsigned char*>(workspace) + offsetof(Signal, release_fence), &v, 4, hipMemcpyHostToDevice);
  SGL_CHECK_ARG(e == hipSuccess, "xgmi_set_release_fence: %s", hipGetErrorString(e));
  return 0;
}

int64_t sgl_amd_xgmi_data_offset(void) { return kDataOffset; }

int sgl_amd_xgmi_two_stage_owner_units(int64_t rows, int hidden, int world, int epilogue, int num_blocks, int64_t* units_per_rank) {
  SGL_CHECK_ARG(world >= 1 && world <= kMaxWorld && units_per_rank && rows >= 0 && hidden > 0 && hidden % 8 == 0,
                "xgmi_two_stage_owner_units: bad argument");
  const int64_t numel = rows * hidden;
  const int blocks = num_blocks > 0 ? num_blocks : two_stage_auto_blocks(rows, numel, epilogue);
  for (int r = 0; r < world; ++r) units_per_rank[r] = 0;
  if (epilogue) {                                    // unit = a row
    for (int64_t b = 0; b < blocks; ++b) {
      int64_t j = 0;
      for (int64_t r = b; r < rows; r += blocks, ++j) units_per_rank[two_stage_owner(b, j, world)] += 1;
    }
  } else {                                           // unit = a chunk of kArThreads 16-byte vectors
    const int64_t nvec = numel / 8, nchunks = (nvec + kArThreads - 1) / kArThreads;
    for (int64_t b = 0; b < blocks; ++b) {
      int64_t k = 0;
      for (int64_t c0 = b * 4; c0 < nchunks; c0 += static_cast<int64_t>(blocks) * 4, ++k)
        for (int u = 0; u < 4 && c0 + u < nchunks; ++u) units_per_rank[two_stage_owner(b, k, world)] += 1;
    }
  }
  return blocks;
}

int64_t sgl_amd_xgmi_workspace_bytes(int64_t max_message_bytes) { return kDataOffset + ((max_message_bytes + 255) / 256) * 256; }

int sgl_amd_xgmi_max_world(void) { return kMaxWorld; }

int sgl_amd_xgmi_alloc(int64_t bytes, void** out_ptr) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(bytes >= kDataOffset && out_ptr, "xgmi_alloc: need at least %lld bytes", (long long)kDataOffset);
  void* ptr = nullptr;
  hipError_t e = hipExtMallocWithFlags(&ptr, static_cast<size_t>(bytes), hipDeviceMallocUncached);
  SGL_CHECK_ARG(e == hipSuccess, "xgmi_alloc: hipExtMallocWithFlags(%lld, uncached): %s", (long long)bytes, hipGetErrorString(e));
  e = hipMemset(ptr, 0, static_cast<size_t>(bytes));
  SGL_CHECK_ARG(e == hipSuccess, "xgmi_alloc: hipMemset: %s", hipGetErrorString(e));
  e = hipDeviceSynchronize();
  SGL_CHECK_ARG(e == hipSuccess, "xgmi_alloc: sync: %s", hipGetErrorString(e));
  *out_ptr = ptr;
  return 0;
}

int sgl_amd_xgmi_free(void* ptr) {
  SGL_CLEAR_STALE_ERROR();
  if (!ptr) return 0;
  hipError_t e = hipFree(ptr);
  SGL_CHECK_ARG(e == hipSuccess, "xgmi_free: %s", hipGetErrorString(e));
  return 0;
}

int sgl_amd_xgmi_timed_out(const void* workspace) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(workspace, "xgmi_timed_out: null workspace");
  uint32_t v = 0;
  hipError_t e = hipMemcpy(&v, static_cast<const unsigned char*>(workspace) + offsetof(Signal, timed_out), 4, hipMemcpyDeviceToHost);
  SGL_CHECK_ARG(e == hipSuccess, "xgmi_timed_out: %s", hipGetErrorString(e));
  return static_cast<int>(v);
}

int sgl_amd_xgmi_ipc_handle_bytes(void) { return static_cast<int>(sizeof(hipIpcMemHandle_t)); }

int sgl_amd_xgmi_ipc_get_handle(void* ptr, void* out_handle) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(ptr && out_handle, "xgmi_ipc_get_handle: null argument");
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, ptr);
  SGL_CHECK_ARG(e == hipSuccess, "xgmi_ipc_get_handle: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)", hipGetErrorString(e));
  std::memcpy(out_handle, &h, sizeof(h));
  return 0;
}

int sgl_amd_xgmi_ipc_open_handle(const void* handle, void** out_ptr) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(handle && out_ptr, "xgmi_ipc_open_handle: null argument");
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle, sizeof(h));
  void* ptr = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
  SGL_CHECK_ARG(e == hipSuccess, "xgmi_ipc_open_handle: %s", hipGetErrorString(e));
  *out_ptr = ptr;
  return 0;
}

int sgl_amd_xgmi_ipc_close_handle(void* ptr) {
  SGL_CLEAR_STALE_ERROR();
  if (!ptr) return 0;
  hipError_t e = hipIpcCloseMemHandle(ptr);
  SGL_CHECK_ARG(e == hipSuccess, "xgmi_ipc_close_handle: %s", hipGetErrorString(e));
  return 0;
}

int sgl_amd_xgmi_one_shot_all_reduce(const void* inp, void* out, int64_t rows, int hidden, int rank, int world,
                                     const void* const* peer_workspaces_host, int64_t workspace_bytes, int epilogue,
                                     void* residual, const void* norm_weight, float eps, int num_blocks, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  // world 1 = loopback: the rank-shape benchmarks run one rank of a TP job on one GPU with the collectives' launches
  // (copy, flag barriers, sum of one copy, epilogue) in place and only the wire missing
  SGL_CHECK_ARG(world == 1 || world == 2 || world == 4 || world == 8, "xgmi_one_shot_all_reduce: world=%d (supported: 1, 2, 4, 8)", world);
  SGL_CHECK_ARG(rank >= 0 && rank < world && peer_workspaces_host, "xgmi_one_shot_all_reduce: bad rank / peers");
  SGL_CHECK_ARG(rows >= 0 && hidden > 0 && hidden % 8 == 0, "xgmi_one_shot_all_reduce: hidden=%d must be a multiple of 8", hidden);
  if (rows == 0) return 0;
  const int64_t numel = rows * hidden;
  SGL_CHECK_ARG(kDataOffset + numel * 2 <= workspace_bytes, "xgmi_one_shot_all_reduce: message of %lld bytes does not fit the %lld-byte workspace",
                (long long)(numel * 2), (long long)workspace_bytes);
  SGL_CHECK_ARG(epilogue == 0 || epilogue == 1, "xgmi_one_shot_all_reduce: epilogue must be 0 (none) or 1 (add_rmsnorm)");
  SGL_CHECK_ARG(epilogue == 0 || (residual && norm_weight && hidden <= kArThreads * 8 * 4), "xgmi_one_shot_all_reduce: add_rmsnorm needs residual, norm_weight and hidden <= %d", kArThreads * 8 * 4);
  SGL_CHECK_ARG(inp && out, "xgmi_one_shot_all_reduce: null tensor");
  ArParams p{};
  for (int r = 0; r < world; ++r) {
    SGL_CHECK_ARG(peer_workspaces_host[r], "xgmi_one_shot_all_reduce: peer %d is not mapped", r);
    p.peers.base[r] = static_cast<unsigned char*>(const_cast<void*>(peer_workspaces_host[r]));
  }
  p.inp = static_cast<const uint16_t*>(inp); p.out = static_cast<uint16_t*>(out);
  p.residual = static_cast<uint16_t*>(residual); p.norm_w = static_cast<const uint16_t*>(norm_weight);
  p.numel = numel; p.rows = static_cast<int>(rows); p.hidden = hidden; p.rank = rank; p.world = world; p.epilogue = epilogue; p.eps = eps;
  SGL_CHECK_ARG(workspace_bytes > 0 && workspace_bytes < (int64_t{1} << 31), "xgmi: workspace_bytes=%lld (buffer descriptors address < 2 GiB)", (long long)workspace_bytes);
  p.ws_bytes = static_cast<int>(workspace_bytes);
  int blocks = num_blocks;
  if (blocks <= 0) {
    // enough workgroups to keep ~7 links busy, never more than the flag table has rows
    const int64_t want = epilogue ? rows : (numel / 8 + kArThreads * 2 - 1) / (kArThreads * 2);
    blocks = static_cast<int>(want < 1 ? 1 : (want > 64 ? 64 : want));
    if (blocks > g_xgmi_auto_blocks_cap) blocks = g_xgmi_auto_blocks_cap;
  }
  SGL_CHECK_ARG(blocks >= 1 && blocks <= kMaxBlocks, "xgmi_one_shot_all_reduce: num_blocks=%d (1..%d)", blocks, kMaxBlocks);
  hipStream_t st = as_stream(stream);
  switch (world) {
    case 1: hipLaunchKernelGGL(xgmi_one_shot_all_reduce_kernel<1>, dim3(blocks), dim3(kArThreads), 0, st, p); break;
    case 2: hipLaunchKernelGGL(xgmi_one_shot_all_reduce_kernel<2>, dim3(blocks), dim3(kArThreads), 0, st, p); break;
    case 4: hipLaunchKernelGGL(xgmi_one_shot_all_reduce_kernel<4>, dim3(blocks), dim3(kArThreads), 0, st, p); break;
    default: hipLaunchKernelGGL(xgmi_one_shot_all_reduce_kernel<8>, dim3(blocks), dim3(kArThreads), 0, st, p); break;
  }
  SGL_CHECK_LAUNCH("xgmi_one_shot_all_reduce");
  return 0;
}

int sgl_amd_xgmi_arm(void* workspace, int trap_on_timeout) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(workspace, "xgmi_arm: null workspace");
  const uint32_t v = trap_on_timeout ? 1u : 0u;
  hipError_t e = hipMemcpy(static_cast<unsigned char*>(workspace) + offsetof(Signal, trap_on_timeout), &v, 4, hipMemcpyHostToDevice);
  SGL_CHECK_ARG(e == hipSuccess, "xgmi_arm: %s", hipGetErrorString(e));
  return 0;
}

// shared argument checks of the two collectives below
static int fill_peers(const char* who, ArParams* p, int rank, int world, const void* const* peer_workspaces_host) {
  SGL_CHECK_ARG(world == 1 || world == 2 || world == 4 || world == 8, "%s: world=%d (supported: 1, 2, 4, 8)", who, world);
  SGL_CHECK_ARG(rank >= 0 && rank < world && peer_workspaces_host, "%s: bad rank / peers", who);
  for (int r = 0; r < world; ++r) {
    SGL_CHECK_ARG(peer_workspaces_host[r], "%s: peer %d is not mapped", who, r);
    p->peers.base[r] = static_cast<unsigned char*>(const_cast<void*>(peer_workspaces_host[r]));
  }
  p->rank = rank; p->world = world;
  return 0;
}

int sgl_amd_xgmi_two_stage_all_reduce(const void* inp, void* out, int64_t rows, int hidden, int rank, int world,
                                      const void* const* peer_workspaces_host, int64_t workspace_bytes, int epilogue,
                                      void* residual, const void* norm_weight, float eps, int num_blocks, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(rows >= 0 && hidden > 0 && hidden % 8 == 0 && inp && out, "xgmi_two_stage_all_reduce: hidden=%d must be a multiple of 8", hidden);
  if (rows == 0) return 0;
  const int64_t numel = rows * hidden;
  SGL_CHECK_ARG(epilogue == 0 || epilogue == 1, "xgmi_two_stage_all_reduce: epilogue must be 0 (none) or 1 (add_rmsnorm)");
  SGL_CHECK_ARG(epilogue == 0 || (residual && norm_weight && hidden <= kArThreads * 8 * 4),
                "xgmi_two_stage_all_reduce: add_rmsnorm needs residual, norm_weight and hidden <= %d", kArThreads * 8 * 4);
  // the workspace's data area is cut in two: copies of the inputs, then the published sums
  const int64_t half = ((workspace_bytes - kDataOffset) / 2) / 256 * 256;
  SGL_CHECK_ARG(numel * 2 <= half, "xgmi_two_stage_all_reduce: message of %lld bytes needs a workspace of %lld (have %lld)",
                (long long)(numel * 2), (long long)(kDataOffset + 4 * numel + 512), (long long)workspace_bytes);
  ArParams p{};
  if (int rc = fill_peers("xgmi_two_stage_all_reduce", &p, rank, world, peer_workspaces_host)) return rc;
  p.inp = static_cast<const uint16_t*>(inp); p.out = static_cast<uint16_t*>(out); p.numel = numel;
  p.rows = static_cast<int>(rows); p.hidden = hidden; p.epilogue = epilogue; p.eps = eps;
  SGL_CHECK_ARG(workspace_bytes > 0 && workspace_bytes < (int64_t{1} << 31), "xgmi: workspace_bytes=%lld (buffer descriptors address < 2 GiB)", (long long)workspace_bytes);
  p.ws_bytes = static_cast<int>(workspace_bytes);
  p.residual = static_cast<uint16_t*>(residual); p.norm_w = static_cast<const uint16_t*>(norm_weight);
  int blocks = num_blocks;
  if (blocks <= 0) {
    blocks = two_stage_auto_blocks(rows, numel, epilogue);
  }
  SGL_CHECK_ARG(blocks >= 1 && blocks <= kMaxBlocks, "xgmi_two_stage_all_reduce: num_blocks=%d (1..%d)", blocks, kMaxBlocks);
  hipStream_t st = as_stream(stream);
  const int64_t off = kDataOffset + half;
  switch (world) {
    case 1: hipLaunchKernelGGL(xgmi_two_stage_all_reduce_kernel<1>, dim3(blocks), dim3(kArThreads), 0, st, p, off); break;
    case 2: hipLaunchKernelGGL(xgmi_two_stage_all_reduce_kernel<2>, dim3(blocks), dim3(kArThreads), 0, st, p, off); break;
    case 4: hipLaunchKernelGGL(xgmi_two_stage_all_reduce_kernel<4>, dim3(blocks), dim3(kArThreads), 0, st, p, off); break;
    default: hipLaunchKernelGGL(xgmi_two_stage_all_reduce_kernel<8>, dim3(blocks), dim3(kArThreads), 0, st, p, off); break;
  }
  SGL_CHECK_LAUNCH("xgmi_two_stage_all_reduce");
  return 0;
}

int sgl_amd_xgmi_all_gather(const void* inp, void* out, int64_t rows, int cols_per_rank, int rank, int world,
                            const void* const* peer_workspaces_host, int64_t workspace_bytes, int num_blocks, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(rows >= 0 && cols_per_rank > 0 && cols_per_rank % 8 == 0 && inp && out,
                "xgmi_all_gather: cols_per_rank=%d must be a multiple of 8", cols_per_rank);
  if (rows == 0) return 0;
  const int64_t numel = rows * cols_per_rank;
  SGL_CHECK_ARG(kDataOffset + numel * 2 <= workspace_bytes, "xgmi_all_gather: shard of %lld bytes does not fit the %lld-byte workspace",
                (long long)(numel * 2), (long long)workspace_bytes);
  ArParams p{};
  if (int rc = fill_peers("xgmi_all_gather", &p, rank, world, peer_workspaces_host)) return rc;
  p.inp = static_cast<const uint16_t*>(inp); p.out = static_cast<uint16_t*>(out); p.numel = numel;
  p.rows = static_cast<int>(rows); p.hidden = cols_per_rank;
  SGL_CHECK_ARG(workspace_bytes > 0 && workspace_bytes < (int64_t{1} << 31), "xgmi: workspace_bytes=%lld (buffer descriptors address < 2 GiB)", (long long)workspace_bytes);
  p.ws_bytes = static_cast<int>(workspace_bytes);
  int blocks = num_blocks;
  if (blocks <= 0) {
    const int64_t chunks = (numel / 8 + kArThreads - 1) / kArThreads;
    blocks = static_cast<int>(chunks < kMaxBlocks ? (chunks < 1 ? 1 : chunks) : kMaxBlocks);
    if (blocks > g_xgmi_auto_blocks_cap) blocks = g_xgmi_auto_blocks_cap;
  }
  SGL_CHECK_ARG(blocks >= 1 && blocks <= kMaxBlocks, "xgmi_all_gather: num_blocks=%d (1..%d)", blocks, kMaxBlocks);
  hipStream_t st = as_stream(stream);
  switch (world) {
    case 1: hipLaunchKernelGGL(xgmi_all_gather_kernel<1>, dim3(blocks), dim3(kArThreads), 0, st, p); break;
    case 2: hipLaunchKernelGGL(xgmi_all_gather_kernel<2>, dim3(blocks), dim3(kArThreads), 0, st, p); break;
    case 4: hipLaunchKernelGGL(xgmi_all_gather_kernel<4>, dim3(blocks), dim3(kArThreads), 0, st, p); break;
    default: hipLaunchKernelGGL(xgmi_all_gather_kernel<8>, dim3(blocks), dim3(kArThreads), 0, st, p); break;
  }
  SGL_CHECK_LAUNCH("xgmi_all_gather");
  return 0;
}

}  // extern "C"

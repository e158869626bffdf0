// Shared-prefix (cascade) decode attention for gfx950, one launch for the whole step + an LSE merge.
//
// Replaces (reference, /root/reference/python/sglang):
//   srt/layers/attention/triton_backend.py:136-1012 forward_decode -> kernels/ops/attention/
//   decode_attention.py:1163 decode_attention_fwd, which re-reads a radix-shared prefix once per
//   request.  Oracle: srt/layers/attention/torch_native_backend.py:176-277 (oracle/ops.py).
//
// RadixAttention batches share KV rows: requests whose req_to_token rows start with the same slots
// attend to the SAME rows of the pool.  The device plan (sgl_amd_cascade_plan) groups them; this
// kernel then works on "chunk items", each one workgroup:
//   shared item  = (group, 128-token chunk of the shared prefix, <= 64/G members): the chunk's K/V
//                  rows are read ONCE and multiplied with the decode queries of all members
//                  (members x G query heads = up to 64 rows) on the matrix cores;
//   private item = (request, 128-token chunk of its unshared suffix): the same code with one member.
// Every item is a single shot -- no KV loop: 128 slot ids -> 128 K rows + 128 V rows gathered in one
// burst (16 x 16 B loads in flight per lane), S^T = K Q^T and O^T = V^T P^T with MFMA 16x16x32 bf16,
// softmax state lane-local (transposed products).  That keeps the dependent-load chain at three
// hops (plan -> slot ids -> rows) whatever the context length; parallelism comes from the number of
// items.  Each item writes an unnormalised (acc, max, sum) partial into its slot of every member;
// the merge kernel combines the slots of a (request, head) in slot order.
#include <type_traits>

#include "common.hpp"
#include "cascade_plan.hpp"
#include "kv_format.hpp"
#include "sglang_amd.h"

using namespace sgl_amd;

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kChunk = 128;       // kv tokens per item
constexpr int kRowsPerItem = 64;  // (member, q head) rows per item: 4 waves x one 16-row tile
constexpr float kNegBig = -1.0e30f;
// launch-form policy of the chunk kernel (see cascade_chunk_kernel); process-wide tuning knobs behind
// sgl_amd_debug_cascade_launch_form, like the extend kernel's shape switch
constexpr int64_t kSingleShotHardCap = 16384;    // the largest value the switch below may take: what the plan clears for is decided against THIS, not the switch
int64_t g_cascade_single_shot_units = 10240;   // worst-case workgroups up to which the one-workgroup-per-unit form is launched
int64_t g_cascade_loop_grid = 256 * 5;         // grid of the looping form: 1280 beat 1024 (= what is resident at four per CU) and equalled 2048

__device__ __forceinline__ bf16x8_t as_frag(const U4& v) { return __builtin_bit_cast(bf16x8_t, v); }

#ifdef CASC_TRACE
// timeline probe (benchmarks/r02_exp8_casc_trace.py): per workgroup, the chip-wide 100 MHz clock at entry, record read,
// K image staged, scores done, V image staged, partials stored
__device__ uint64_t* g_casc_trace = nullptr;
#define CASC_STAMP(i)                                                                                     \
  do {                                                                                                    \
    if (g_casc_trace && threadIdx.x == 0) g_casc_trace[static_cast<int64_t>(blockIdx.x) * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define CASC_STAMP(i)
#endif

struct ChunkParams {
  const uint16_t* q;            // [B, Hq, D]
  const uint16_t* k_cache;      // [slots, Hkv, D]
  const uint16_t* v_cache;
  const int32_t* req_to_token;
  const int64_t* req_pool_indices;
  const int32_t* seq_lens;
  const int32_t* plan;
  float* ws_acc;                // [B, Hq, slots_total, D]
  float* ws_ml;                 // [B, Hq, slots_total, 2]
  int64_t q_stride, kc_stride, vc_stride, r2t_stride;
  KvFormat fmt;                 // pool layout (token-major or paged head-major) + element format (bf16 / e4m3 rows)
  int batch, max_items, num_q_heads, group, members_per_item;
  int num_kv_heads;
  int slots_total;
  float scale_log2;
};

// The K rows, then (after S^T) the transposed V rows, pass through ONE 16 KiB LDS image; at D = 128 it is filled twice
// per operand: K by token halves (64 tokens x 128 dims: each score tile is complete inside its half), V^T by head-dim
// halves (64 dims x 128 tokens: each output tile is complete inside its half).  With <= 96 VGPRs that makes five
// workgroups per CU resident -- 1280 on the chip: the bench batch's 1248 units run as one round
// (at four per CU the last 224 started when the first finished, 7 us of a 20 us kernel, benchmarks/r02_exp8).
// Two forms of the launch.
//   cascade_chunk_kernel: one workgroup per unit; the grid covers the worst-case item count of the batch and the workgroups behind
//     the end of the device-built list leave after one load.  96 registers: five workgroups per CU.  Chosen while the worst
//     case stays below kSingleShotUnits workgroups (a request table of a few thousand tokens: the bench's 1.2 k-token table
//     asks for 9.5 k).
//   cascade_chunk_loop_kernel: a grid of kLoopGrid resident workgroups walks the list -- workgroup w takes units w, 2 G - 1 - w, 2 G + w,
//     ... (a snake over the passes: the shared items at the head of the list are the heavy ones, the workgroups that got
//     them in one pass get the far end of the next) below the list's end, which every workgroup knows after ONE scalar load
//     (header[0]) that travels with its first item record.  Chosen for large request tables: the worst case of a 64-request
//     batch is 58 k workgroups at an 8 k-token table (Llama-3-8B's own context length under the reference's scheduler), ~1 M at
//     128 k, and a workgroup behind the list's end costs ~0.25 ns of dispatch -- 20 -> 33.7 us per layer at 8 k, measured
//     under the reference's scheduler (profiles/r05_sched_step_timeline_hooks.txt).  The loop keeps the thread index and the
//     kernel arguments opaque per pass (nothing lane-derived or argument-derived is loop-invariant to the compiler: round 2's
//     persistent loop died of hoisted LDS addresses), and it is given 128 registers -- four workgroups per CU, no spills --
//     because at the 96 budget hipcc parks ~30 registers in scratch once there is a loop around the body: that instance was
//     built and measured (profiles/r05_exp1b_cascade_forms.json): 34.1 us per layer against 26.7 at four per CU and 25.6 for
//     the one-workgroup-per-unit form on the bench's own table; grids of 1280 / 2048 equal, 1024 (= what is resident) 28.0.
template <int D, bool FP8, bool HND>
__global__ __launch_bounds__(kThreads, 5) void cascade_chunk_kernel(ChunkParams p) {
  __shared__ U4 sm[(kChunk / (D > 64 ? D / 64 : 1)) * (D / 8)];   // K: [token of the part][piece ^ swz];  V^T: [d of the part][8-token chunk ^ swz]
  __shared__ float ml_s[64];                                      // token-split units: (max, sum) of [wave][row < 8]
  __shared__ U4 side_s[D > 64 ? 4 * 8 * (D - 64) / 4 : 1];        // ... and their O^T partials of every V^T part but the last
  const int unit = blockIdx.x;
  const int tid = threadIdx.x;
#include "cascade_chunk_body.inc"
}

template <int D, bool FP8, bool HND>
__device__ __forceinline__ void cascade_chunk_unit(const ChunkParams& p, U4* sm, float* ml_s, U4* side_s, const int unit, const int tid) {
#include "cascade_chunk_body.inc"
}

template <int D, bool FP8, bool HND>
__global__ __launch_bounds__(kThreads, 4) void cascade_chunk_loop_kernel(ChunkParams p) {
  __shared__ U4 sm[(kChunk / (D > 64 ? D / 64 : 1)) * (D / 8)];
  __shared__ float ml_s[64];
  __shared__ U4 side_s[D > 64 ? 4 * 8 * (D - 64) / 4 : 1];
  int n_items = __builtin_amdgcn_readfirstlane(p.plan[0]);                   // header[0]: the list's length
  if (n_items > p.max_items) n_items = p.max_items;
  const int n_units = n_items * p.num_kv_heads;
  const int G = static_cast<int>(gridDim.x), w0 = static_cast<int>(blockIdx.x);
  // The kernel's one argument sits at the start of the kernarg segment; every pass re-reads it through an address the
  // compiler cannot see through, exactly as a fresh launch would (scalar loads, served by the scalar cache), and takes the
  // thread index as an opaque value.
  typedef const __attribute__((address_space(4))) uint64_t KernargWord;
  const uint64_t kp = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  for (int base = 0, pass = 0; base < n_units; base += G, ++pass) {
    const int unit = base + ((pass & 1) ? G - 1 - w0 : w0);
    if (unit < n_units) {                                                     // (workgroup-uniform; false only in the last pass)
      uint64_t ka = kp;
      asm volatile("" : "+s"(ka));
      KernargWord* kw = reinterpret_cast<KernargWord*>(ka);
      static_assert(sizeof(ChunkParams) % 8 == 0 && std::is_trivially_copyable<ChunkParams>::value, "kernarg image copied as 8-byte words");
      struct Raw { uint64_t w[sizeof(ChunkParams) / 8]; } raw;
#pragma unroll
      for (unsigned i = 0; i < sizeof(ChunkParams) / 8; ++i) raw.w[i] = kw[i];
      const ChunkParams pl = __builtin_bit_cast(ChunkParams, raw);
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      cascade_chunk_unit<D, FP8, HND>(pl, sm, ml_s, side_s, unit, tid);
      __syncthreads();                                                        // the image is reused by the next unit
    }
  }
}

// ---- cascade plan: group the requests of a decode batch that share a KV prefix ----------------
// One workgroup, once per decode step.  Requests that share ANY cached prefix share their first slot
// (page_size-agnostic: identical slot ids <=> the same cached tokens), so the leader of request b is the
// lowest batch index with the same first slot; the shared length with the leader is the first mismatch
// of the two req_to_token rows.  A group's shared part is the minimum over its members, rounded down
// to kv_tile; groups with < 2 members or a short shared part are dropped.  The output is ONE compact
// list of self-contained chunk items: the shared chunks of every group (x member tiles), then the
// private chunks of every request.
constexpr int kPlanThreads = 1024;
constexpr int kPlanMaxBatch = 1024;

// First launch of the plan: every request's leader (the lowest batch index with the same first slot) and the length of
// the prefix it shares with that leader.  One WAVE per request, sixteen requests per workgroup: inside ONE workgroup the
// sixteen waves walked their requests one after the other, a memory round trip each -- 26 / 183 us at 64 / 512 requests
// (benchmarks/r03_exp7_plan_batch.py).  Results go through the plan buffer's `compare` scratch to the second launch.
__global__ __launch_bounds__(kPlanThreads) void cascade_plan_compare_kernel(
    const int32_t* __restrict__ req_to_token, int64_t r2t_stride, const int64_t* __restrict__ req_pool_indices,
    const int32_t* __restrict__ seq_lens, int batch, int32_t* __restrict__ compare) {
  __shared__ int first_slot[kPlanMaxBatch];
  __shared__ int pool_row[kPlanMaxBatch], len_s[kPlanMaxBatch];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int b = tid; b < batch; b += kPlanThreads) {
    const int len = seq_lens[b];
    const int row = static_cast<int>(req_pool_indices[b]);
    pool_row[b] = row;
    len_s[b] = len;
    first_slot[b] = len > 1 ? req_to_token[static_cast<int64_t>(row) * r2t_stride] : -1 - b;   // unique when too short
  }
  __syncthreads();
  const int b = static_cast<int>(blockIdx.x) * (kPlanThreads / 64) + wid;
  if (b >= batch) return;
  int l = b;
  const int fs = first_slot[b];
  for (int c0 = 0; c0 < b; c0 += 64) {
    const int c = c0 + lane;
    const unsigned long long m = __ballot(c < b && first_slot[c] == fs);
    if (m != 0ull) { l = c0 + __ffsll(static_cast<long long>(m)) - 1; break; }
  }
  // 2048 positions per round (32 loads in flight per lane and row)
  constexpr int kCmpLoads = 32;
  int common = 0;
  if (l != b) {
    const int32_t* ra = req_to_token + static_cast<int64_t>(pool_row[b]) * r2t_stride;
    const int32_t* rb = req_to_token + static_cast<int64_t>(pool_row[l]) * r2t_stride;
    int lim = len_s[b] - 1;                    // the newest token's slot is never shared
    const int ll = len_s[l] - 1;
    if (ll < lim) lim = ll;
    common = lim;
    for (int t0 = 0; t0 < lim; t0 += 64 * kCmpLoads) {
      int va[kCmpLoads], vb[kCmpLoads];
#pragma unroll
      for (int u = 0; u < kCmpLoads; ++u) {
        const int t = t0 + 64 * u + lane;
        const int tc = t < lim ? t : lim - 1;
        va[u] = ra[tc];
        vb[u] = rb[tc];
      }
      int first = 0x7fffffff;
#pragma unroll
      for (int u = kCmpLoads - 1; u >= 0; --u) {
        const int t = t0 + 64 * u + lane;
        const unsigned long long mm = __ballot(t < lim && va[u] != vb[u]);
        if (mm != 0ull) first = t0 + 64 * u + __ffsll(static_cast<long long>(mm)) - 1;
      }
      if (first != 0x7fffffff) { common = first; break; }
    }
  }
  if (lane == 0) {
    compare[b] = l;
    compare[batch + b] = common;
  }
}

__global__ __launch_bounds__(kPlanThreads) void cascade_plan_kernel(
    const int32_t* __restrict__ req_to_token, int64_t r2t_stride, const int64_t* __restrict__ req_pool_indices,
    const int32_t* __restrict__ seq_lens, int batch, int min_shared, int chunk_tokens, int tokens_per_tile,
    int kv_tile, int max_shared, int32_t* __restrict__ plan, int max_items, int zero_items) {
  __shared__ int first_slot[kPlanMaxBatch];
  __shared__ int leader[kPlanMaxBatch];
  __shared__ int grp_min[kPlanMaxBatch];
  __shared__ int grp_cnt[kPlanMaxBatch];
  __shared__ int grp_id[kPlanMaxBatch];
  __shared__ int scan[kPlanMaxBatch];
  __shared__ int order[kPlanMaxBatch];
  __shared__ int g_leader[kPlanMaxBatch / 2], g_kv[kPlanMaxBatch / 2], g_first[kPlanMaxBatch / 2], g_tiles[kPlanMaxBatch / 2],
      g_rows0[kPlanMaxBatch / 2], g_members[kPlanMaxBatch / 2];
  __shared__ int n_shared_items, n_groups, n_rows;
  __shared__ int pool_row[kPlanMaxBatch], len_s[kPlanMaxBatch];   // read once: every later phase would pay the hop again
  __shared__ int wave_tot[kPlanThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const CascadePlanView pv = cascade_plan_view(plan, batch, max_items);
  // members == 0 marks the end of the list for the one-workgroup-per-unit chunk kernel, whose grid never reaches beyond
  // zero_items entries (the looping form reads the list's length from the header): 13.8 -> 98 us per step at a 128 k-token
  // request table when all max_items entries were cleared (profiles/r05_exp1_cascade_table_width.json)
  for (int i = tid; i < zero_items; i += kPlanThreads) pv.items[8 * i + 3] = 0;
  for (int b = tid; b < batch; b += kPlanThreads) {
    pool_row[b] = static_cast<int>(req_pool_indices[b]);
    len_s[b] = seq_lens[b];
    leader[b] = pv.compare[b];                         // first launch: leader and common prefix of every request
    grp_min[b] = 0x7fffffff;
    grp_cnt[b] = 0;
  }
  __syncthreads();
  for (int b = tid; b < batch; b += kPlanThreads) {
    const int l = leader[b];
    if (l != b) {
      atomicMin(&grp_min[l], pv.compare[batch + b]);
      atomicAdd(&grp_cnt[l], 1);
    }
  }
  __syncthreads();
  // ---- grouping decisions.  (One thread walking the batch through LDS took ~0.3 us per request -- every step of it a
  // dependent LDS round trip: 25 us of a 39 us kernel at 64 requests.)  Three parallel passes instead:
  // (1) every request works out whether it LEADS a candidate group and what the group would cost;
  for (int b = tid; b < batch; b += kPlanThreads) {
    grp_id[b] = -1;
    int cost = 0;
    if (leader[b] == b && grp_cnt[b] >= 1) {
      int kv = grp_min[b] / kv_tile * kv_tile;
      if (kv > max_shared) kv = max_shared;
      if (kv >= min_shared) {
        const int members = grp_cnt[b] + 1;
        const int tiles = (members + tokens_per_tile - 1) / tokens_per_tile;
        const int chunks = (kv + chunk_tokens - 1) / chunk_tokens;
        cost = tiles * chunks;
        scan[b] = kv;                                  // parked for pass (2): kv, tiles (members = grp_cnt + 1)
        order[b] = tiles;
      }
    }
    first_slot[b] = cost;                              // first_slot is dead after the leader search: reused as "cost"
  }
  __syncthreads();
  // (2) wave 0 accepts the candidates in batch order -- the running item count decides (plan full: the rest stay
  // ungrouped), so this is sequential, but over wave-uniform scalars read with readlane, not over LDS;
  if (wid == 0) {
    int ng = 0, rows = 0, n_items = 0;
    for (int base = 0; base < batch; base += 64) {
      const int bb = base + lane;
      const int cost = bb < batch ? first_slot[bb] : 0;
      const int members = bb < batch ? grp_cnt[bb] + 1 : 0;
      unsigned long long todo = __ballot(cost > 0);
      int my_gid = -1, my_first = 0, my_rows0 = 0;
      while (todo != 0ull) {
        const int j = __ffsll(static_cast<long long>(todo)) - 1;
        todo &= todo - 1;
        const int cj = __builtin_amdgcn_readlane(cost, j);
        const int mj = __builtin_amdgcn_readlane(members, j);
        // plan full (leave room for the private chunks of every request): the rest stay ungrouped
        if (n_items + cj > max_items / 2) continue;
        if (lane == j) { my_gid = ng; my_first = n_items; my_rows0 = rows; }
        n_items += cj;
        rows += mj;
        ++ng;
      }
      if (my_gid >= 0) {
        grp_id[bb] = my_gid;
        g_leader[my_gid] = bb; g_kv[my_gid] = scan[bb]; g_first[my_gid] = my_first; g_tiles[my_gid] = order[bb];
        g_rows0[my_gid] = my_rows0; g_members[my_gid] = members;
      }
    }
    if (lane == 0) { n_groups = ng; n_rows = rows; n_shared_items = n_items; }
  }
  __syncthreads();
  // (3) members in batch order inside each group, then the ungrouped requests: a request's place is the number of
  // earlier requests of its kind.
  {
    const int rows_total = n_rows;
    int my_pos[ (kPlanMaxBatch + kPlanThreads - 1) / kPlanThreads ];
    int k = 0;
    for (int b = tid; b < batch; b += kPlanThreads, ++k) {
      const int gi = grp_id[leader[b]];
      int rank = 0;
      for (int c = 0; c < b; ++c) rank += (grp_id[leader[c]] == gi) ? 1 : 0;   // gi < 0: all ungrouped count together
      my_pos[k] = gi >= 0 ? g_rows0[gi] + rank : rows_total + rank;
    }
    __syncthreads();                                   // `order` was scratch of pass (1)/(2) until here
    k = 0;
    for (int b = tid; b < batch; b += kPlanThreads, ++k) order[my_pos[k]] = b;
  }
  __syncthreads();
  // ---- write-out, all threads ----
  const int ng = n_groups, rows = n_rows;
  for (int b = tid; b < batch; b += kPlanThreads) {
    const int gi = grp_id[leader[b]];
    pv.req_shared[b] = gi >= 0 ? g_kv[gi] : 0;
    pv.batch_order[b] = order[b];
    if (b < rows) pv.member_rows[b] = order[b];
  }
  for (int gi = tid; gi < ng; gi += kPlanThreads) {
    pv.group_pool_row[gi] = pool_row[g_leader[gi]];
    pv.group_kvlen[gi] = g_kv[gi];
    pv.group_qo[gi + 1] = g_rows0[gi] + g_members[gi];
  }
  for (int i = tid; i < n_shared_items; i += kPlanThreads) {
    int gi = 0;
    while (gi + 1 < ng && g_first[gi + 1] <= i) ++gi;
    const int rel = i - g_first[gi];
    const int c = rel / g_tiles[gi], t = rel - c * g_tiles[gi];
    const int left = g_members[gi] - t * tokens_per_tile;
    const int kvn = g_kv[gi] - c * chunk_tokens;
    int32_t* it = pv.items + 8 * i;
    it[0] = c;                                         // partial slot
    it[1] = c * chunk_tokens;                          // first kv token
    it[2] = kvn < chunk_tokens ? kvn : chunk_tokens;   // kv tokens
    it[3] = left < tokens_per_tile ? left : tokens_per_tile;   // members
    it[4] = g_rows0[gi] + t * tokens_per_tile;         // first entry of member_rows
    it[5] = 0;                                         // shared item
    it[6] = pool_row[g_leader[gi]];
    it[7] = gi;
  }
  if (tid == 0) {
    pv.group_qo[0] = 0;
    pv.header[1] = ng;
    pv.header[2] = rows;
    pv.header[3] = n_shared_items;
  }
  // ---- private chunks of every request, appended behind the shared items (block-wide exclusive scan) ----
  int mine = 0, sh = 0, len = 0;
  if (tid < batch) {
    const int gi = grp_id[leader[tid]];
    sh = gi >= 0 ? g_kv[gi] : 0;
    len = len_s[tid];
    mine = len > sh ? (len - sh + chunk_tokens - 1) / chunk_tokens : 0;
  }
  // inclusive scan: inside the wave by shuffles, across the 16 waves through LDS (three barriers, not twenty)
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wave_tot[wid] = incl;
  __syncthreads();
  int wave_base = 0;
  for (int w = 0; w < wid; ++w) wave_base += wave_tot[w];
  incl += wave_base;
  scan[tid] = incl;
  __syncthreads();
  const int base = n_shared_items + scan[tid] - mine;
  const int first_slot_p = (sh + chunk_tokens - 1) / chunk_tokens;
  for (int j = 0; j < mine; ++j) {
    if (base + j >= max_items) break;                // cannot happen when max_items is sized by the caller
    int32_t* it = pv.items + 8 * (base + j);
    const int kvb = sh + j * chunk_tokens;
    it[0] = first_slot_p + j;
    it[1] = kvb;
    it[2] = len - kvb < chunk_tokens ? len - kvb : chunk_tokens;
    it[4] = tid;                                     // the request itself
    it[5] = 1;                                       // private item
    it[6] = pool_row[tid];
    it[7] = -1;
    it[3] = 1;
  }
  if (tid == kPlanThreads - 1) {
    const int total = n_shared_items + scan[tid];
    pv.header[0] = total < max_items ? total : max_items;
  }
}

// ---- merge: slots [0, n_slots(b)) of every (request, head), in slot order ----------------------
// n_slots = shared chunks + private chunks of the request; every one of them was written this step.
// Block = 256 / (D/4) heads x (D/4) lanes; a thread owns 4 consecutive output elements of one head.
constexpr int kMergeBlock = 16;  // slots loaded per round (all loads of a round are in flight together)

__global__ __launch_bounds__(256) void cascade_merge2_kernel(const float* __restrict__ ws_acc, const float* __restrict__ ws_ml,
                                                             const int32_t* __restrict__ plan, const int32_t* __restrict__ seq_lens,
                                                             uint16_t* __restrict__ out, int64_t out_stride, int batch,
                                                             int max_items, int num_q_heads, int head_dim, int slots_total,
                                                             float out_scale) {
  const int tpd = head_dim >> 2;
  const int b = blockIdx.x;
  const int hq = blockIdx.y * (256 / tpd) + threadIdx.x / tpd;
  const int d = (threadIdx.x % tpd) * 4;
  if (hq >= num_q_heads) return;
  const CascadePlanView pv = cascade_plan_view(plan, batch, max_items);
  const int sh = pv.req_shared[b];
  const int len = seq_lens[b];
  int n = (sh + kChunk - 1) / kChunk + (len > sh ? (len - sh + kChunk - 1) / kChunk : 0);
  if (n > slots_total) n = slots_total;
  const int64_t base = (static_cast<int64_t>(b) * num_q_heads + hq) * slots_total;
  const float2* ml = reinterpret_cast<const float2*>(ws_ml) + base;
  const float* accp = ws_acc + base * head_dim + d;
  float m_run = kNegBig, l = 0.f;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = 0; s0 < n; s0 += kMergeBlock) {
    float2 mlv[kMergeBlock];
    float4 av[kMergeBlock];
#pragma unroll
    for (int i = 0; i < kMergeBlock; ++i) {
      const int s = s0 + i < n ? s0 + i : n - 1;
      mlv[i] = ml[s];
      av[i] = *reinterpret_cast<const float4*>(accp + static_cast<int64_t>(s) * head_dim);
    }
    float m_new = m_run;
#pragma unroll
    for (int i = 0; i < kMergeBlock; ++i)
      if (s0 + i < n && mlv[i].y > 0.f) m_new = fmaxf(m_new, mlv[i].x);
    const float alpha = exp2f(m_run - m_new);
    l *= alpha; o.x *= alpha; o.y *= alpha; o.z *= alpha; o.w *= alpha;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < kMergeBlock; ++i) {
      if (s0 + i < n && mlv[i].y > 0.f) {
        const float sc = exp2f(mlv[i].x - m_run);
        l += mlv[i].y * sc;
        o.x += av[i].x * sc; o.y += av[i].y * sc; o.z += av[i].z * sc; o.w += av[i].w * sc;
      }
    }
  }
  const float inv = (l > 0.f) ? out_scale / l : 0.f;       // out_scale: v_scale of an fp8 pool, else 1
  uint2 w;
  w.x = pack_bf2(o.x * inv, o.y * inv);
  w.y = pack_bf2(o.z * inv, o.w * inv);
  *reinterpret_cast<uint2*>(out + static_cast<int64_t>(b) * out_stride + static_cast<int64_t>(hq) * head_dim + d) = w;
}

}  // namespace

extern "C" {

#ifdef CASC_TRACE
int sgl_amd_debug_casc_trace(void* buf) {
  uint64_t* p = static_cast<uint64_t*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_casc_trace), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

int sgl_amd_cascade_chunk_tokens(void) { return kChunk; }

int64_t sgl_amd_cascade_plan_ints(int64_t batch, int64_t max_items) { return cascade_plan_ints(batch, max_items); }

int sgl_amd_cascade_plan(const int32_t* req_to_token, int64_t req_to_token_stride, const int64_t* req_pool_indices,
                         const int32_t* seq_lens, int64_t batch, int num_q_heads, int num_kv_heads,
                         int min_shared_len, int64_t max_context_len, int32_t* plan, int64_t max_items, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(batch >= 1 && batch <= kPlanMaxBatch, "cascade_plan: batch=%lld (supported: 1..%d)", (long long)batch, kPlanMaxBatch);
  SGL_CHECK_ARG(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0 && num_q_heads / num_kv_heads <= kRowsPerItem, "cascade_plan: bad head counts");
  SGL_CHECK_ARG(max_items >= 1 && plan && max_context_len >= 1, "cascade_plan: bad plan arguments");
  const int members_per_item = kRowsPerItem / (num_q_heads / num_kv_heads);
  const int chunks = static_cast<int>((max_context_len + kChunk - 1) / kChunk);
  // every request's private chunks must fit behind the shared items (which may use max_items / 2): a list that
  // silently dropped a chunk would drop its tokens from the attention
  SGL_CHECK_ARG(max_items >= 2 * batch * (chunks + 1),
                "cascade_plan: max_items=%lld too small for batch=%lld x %d chunks (need >= %lld)", (long long)max_items,
                (long long)batch, chunks + 1, (long long)(2 * batch * (chunks + 1)));
  const CascadePlanView pv = cascade_plan_view(plan, batch, max_items);
  hipLaunchKernelGGL(cascade_plan_compare_kernel, dim3(static_cast<unsigned>((batch + kPlanThreads / 64 - 1) / (kPlanThreads / 64))),
                     dim3(kPlanThreads), 0, as_stream(stream), req_to_token, req_to_token_stride, req_pool_indices, seq_lens,
                     static_cast<int>(batch), pv.compare);
  hipLaunchKernelGGL(cascade_plan_kernel, dim3(1), dim3(kPlanThreads), 0, as_stream(stream), req_to_token,
                     req_to_token_stride, req_pool_indices, seq_lens, static_cast<int>(batch), min_shared_len,
                     kChunk, members_per_item, 64, chunks * kChunk, plan, static_cast<int>(max_items),
                     // end-of-list markers for every item a one-workgroup-per-unit launch of THIS batch can reach (the attention entry's
                     // own bound, batch x (chunks + 1)) whenever such a launch is possible at ANY setting of the launch-form switch --
                     // the switch may change between the plan and the attention launches, or after graphs were captured
                     static_cast<int>(batch * (chunks + 1) * num_kv_heads <= kSingleShotHardCap
                                          ? (batch * (chunks + 1) < max_items ? batch * (chunks + 1) : max_items) : 0));
  SGL_CHECK_LAUNCH("cascade_plan");
  return 0;
}

int sgl_amd_cascade_members_per_item(int num_q_heads, int num_kv_heads) {
  const int group = num_q_heads / (num_kv_heads > 0 ? num_kv_heads : 1);
  return group > 0 && group <= kRowsPerItem ? kRowsPerItem / group : 0;
}

int sgl_amd_cascade_decode_attention(const void* q, const void* k_cache, const void* v_cache, void* out,
                                     const int32_t* req_to_token, int64_t req_to_token_stride,
                                     const int64_t* req_pool_indices, const int32_t* seq_lens, const int32_t* plan,
                                     int64_t batch, int64_t max_items, int num_q_heads, int num_kv_heads, int head_dim,
                                     int64_t q_token_stride, int64_t out_token_stride, int64_t k_cache_row_stride,
                                     int64_t v_cache_row_stride, float sm_scale, int64_t max_context_len,
                                     int slots_total, void* ws_acc, void* ws_ml, void* stream) {
  return sgl_amd_cascade_decode_attention_ex(q, k_cache, v_cache, out, req_to_token, req_to_token_stride, req_pool_indices,
                                             seq_lens, plan, batch, max_items, num_q_heads, num_kv_heads, head_dim,
                                             q_token_stride, out_token_stride, k_cache_row_stride, v_cache_row_stride, sm_scale,
                                             max_context_len, slots_total, ws_acc, ws_ml, 0, 1.0f, 1.0f, 1, 0, stream);
}

int sgl_amd_cascade_decode_attention_ex(const void* q, const void* k_cache, const void* v_cache, void* out,
                                        const int32_t* req_to_token, int64_t req_to_token_stride,
                                        const int64_t* req_pool_indices, const int32_t* seq_lens, const int32_t* plan,
                                        int64_t batch, int64_t max_items, int num_q_heads, int num_kv_heads, int head_dim,
                                        int64_t q_token_stride, int64_t out_token_stride, int64_t k_cache_row_stride,
                                        int64_t v_cache_row_stride, float sm_scale, int64_t max_context_len,
                                        int slots_total, void* ws_acc, void* ws_ml, int kv_fp8, float k_scale, float v_scale,
                                        int page_size, int kv_layout_hnd, void* stream) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(head_dim == 64 || head_dim == 128, "cascade_decode_attention: head_dim=%d not supported (64/128)", head_dim);
  SGL_CHECK_ARG(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0 && num_q_heads / num_kv_heads <= kRowsPerItem,
                "cascade_decode_attention: bad head counts (%d / %d)", num_q_heads, num_kv_heads);
  SGL_CHECK_ARG(q_token_stride % 8 == 0 && k_cache_row_stride % 8 == 0 && v_cache_row_stride % 8 == 0 && out_token_stride % 4 == 0,
                "cascade_decode_attention: strides must keep 16-byte (q, k, v) / 8-byte (out) alignment");
  SGL_CHECK_ARG(k_cache_row_stride > 0 && k_cache_row_stride == v_cache_row_stride,
                "cascade_decode_attention: K and V pools must share a positive row stride");
  SGL_CHECK_ARG(!kv_fp8 || (k_scale > 0.f && v_scale > 0.f), "cascade_decode_attention: fp8 KV needs positive k_scale / v_scale");
  SGL_CHECK_ARG(batch >= 1 && batch <= 1024 && max_items >= 1 && plan && ws_acc && ws_ml, "cascade_decode_attention: bad batch / workspace");
  const int chunks = static_cast<int>((max_context_len + kChunk - 1) / kChunk);
  SGL_CHECK_ARG(slots_total >= chunks + 1, "cascade_decode_attention: slots_total=%d < %d (context chunks + 1)", slots_total, chunks + 1);
  hipStream_t st = as_stream(stream);
  ChunkParams p{};
  p.q = static_cast<const uint16_t*>(q);
  p.k_cache = static_cast<const uint16_t*>(k_cache);
  p.v_cache = static_cast<const uint16_t*>(v_cache);
  p.req_to_token = req_to_token; p.req_pool_indices = req_pool_indices; p.seq_lens = seq_lens; p.plan = plan;
  p.ws_acc = static_cast<float*>(ws_acc); p.ws_ml = static_cast<float*>(ws_ml);
  p.q_stride = q_token_stride; p.kc_stride = k_cache_row_stride; p.vc_stride = v_cache_row_stride; p.r2t_stride = req_to_token_stride;
  p.batch = static_cast<int>(batch); p.max_items = static_cast<int>(max_items); p.num_q_heads = num_q_heads;
  p.group = num_q_heads / num_kv_heads; p.members_per_item = kRowsPerItem / p.group;
  p.num_kv_heads = num_kv_heads;
  p.slots_total = slots_total;
  p.scale_log2 = sm_scale * 1.4426950408889634f * (kv_fp8 ? k_scale : 1.0f);
  SGL_CHECK_ARG(make_kv_format(&p.fmt, k_cache_row_stride, num_kv_heads, head_dim, page_size, kv_layout_hnd, kv_fp8),
                "cascade_decode_attention: HND pools need a power-of-two page_size (got %d); row / page strides below 4 GiB", page_size);
  // worst-case item count of this batch.  A request of len tokens whose first s are its group's shared part owns
  // ceil((len - s) / 128) private items and, as one of m members, 1 / m of the group's ceil(s / 128) x ceil(m / members_per_item)
  // shared items -- at most ceil(s / 128) of them: never more than chunks + 1 items per request (tests/test_cascade_plan_host.py
  // checks the bound on the host restatement of the plan).  Round 4 added member_tiles x chunks on top: 9552 instead of 5632
  // workgroups for the bench batch, i.e. 3.9 k more workgroups that leave after one load (~0.25 ns of dispatch each).
  int64_t units = batch * (chunks + 1);
  if (units > max_items) units = max_items;
  const int64_t worst = units * num_kv_heads;
  const bool loop = worst > g_cascade_single_shot_units;
  dim3 grid(static_cast<unsigned>(loop ? (worst < g_cascade_loop_grid ? worst : g_cascade_loop_grid) : worst));
#define SGL_LAUNCH_CASC2(D_, KERNEL_)                                                                                 \
  do {                                                                                                       \
    if (kv_fp8 && kv_layout_hnd) hipLaunchKernelGGL((KERNEL_<D_, true, true>), grid, dim3(kThreads), 0, st, p);   \
    else if (kv_fp8) hipLaunchKernelGGL((KERNEL_<D_, true, false>), grid, dim3(kThreads), 0, st, p);             \
    else if (kv_layout_hnd) hipLaunchKernelGGL((KERNEL_<D_, false, true>), grid, dim3(kThreads), 0, st, p);      \
    else hipLaunchKernelGGL((KERNEL_<D_, false, false>), grid, dim3(kThreads), 0, st, p);                        \
  } while (0)
#define SGL_LAUNCH_CASC(D_)                                       \
  do {                                                            \
    if (loop) SGL_LAUNCH_CASC2(D_, cascade_chunk_loop_kernel);    \
    else SGL_LAUNCH_CASC2(D_, cascade_chunk_kernel);              \
  } while (0)
  if (head_dim == 128) SGL_LAUNCH_CASC(128);
  else SGL_LAUNCH_CASC(64);
#undef SGL_LAUNCH_CASC2
#undef SGL_LAUNCH_CASC
  const int hpb = 256 / (head_dim / 4);
  hipLaunchKernelGGL(cascade_merge2_kernel, dim3(static_cast<unsigned>(batch), (num_q_heads + hpb - 1) / hpb), dim3(256), 0, st,
                     p.ws_acc, p.ws_ml, plan, seq_lens, static_cast<uint16_t*>(out), out_token_stride, p.batch, p.max_items,
                     num_q_heads, head_dim, slots_total, kv_fp8 ? v_scale : 1.0f);
  SGL_CHECK_LAUNCH("cascade_decode_attention");
  return 0;
}

int sgl_amd_debug_cascade_launch_form(int64_t single_shot_units, int64_t loop_grid) {
  SGL_CLEAR_STALE_ERROR();
  SGL_CHECK_ARG(single_shot_units >= 0 && single_shot_units <= kSingleShotHardCap && loop_grid >= 0 && loop_grid <= (1 << 20),
                "debug_cascade_launch_form: single_shot_units must be 0..%lld, loop_grid 0..2^20", (long long)kSingleShotHardCap);
  g_cascade_single_shot_units = single_shot_units > 0 ? single_shot_units : 10240;
  g_cascade_loop_grid = loop_grid > 0 ? loop_grid : 256 * 5;
  return 0;
}

}  // extern "C"

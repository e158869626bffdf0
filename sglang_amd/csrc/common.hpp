// Shared device/host helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// Everything in csrc/ is written for one target only: wave64, 256 CUs in
// 8 XCDs, 160 KiB LDS per CU, MFMA 16x16x32 bf16.  There is no CUDA path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

namespace sgl_amd {

constexpr int kWave = 64;

// ---- error plumbing (C-ABI: int status + sgl_amd_last_error()) ------------
void set_last_error(const char* fmt, ...);

#define SGL_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      ::sgl_amd::set_last_error(__VA_ARGS__);    \
      return -1;                                 \
    }                                            \
  } while (0)

// hipGetLastError() is per-thread and sticky: another HIP user in the process (torch)
// may have left a benign error behind.  Every entry point clears it first so that
// SGL_CHECK_LAUNCH only reports errors of our own launches.
#define SGL_CLEAR_STALE_ERROR() (void)hipGetLastError()

#define SGL_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      ::sgl_amd::set_last_error("%s: launch failed: %s", name,              \
                                hipGetErrorString(e__));                    \
      return -2;                                                            \
    }                                                                       \
  } while (0)

// ---- bf16 <-> f32 ---------------------------------------------------------
// bf16 values travel as raw uint16_t; conversions are the same
// round-to-nearest-even torch uses (c10::BFloat16), NaN -> 0x7FC0.
__host__ __device__ __forceinline__ float bf2f(uint16_t v) {
  union { uint32_t u; float f; } c;
  c.u = static_cast<uint32_t>(v) << 16;
  return c.f;
}

__host__ __device__ __forceinline__ uint16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  // gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even): one instruction instead of ~8
  return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f));
#endif
  union { uint32_t u; float f; } c;
  c.f = f;
  uint32_t u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}

// round an f32 to the nearest bf16 and come back (models a torch bf16 op
// boundary inside an f32 pipeline).
__host__ __device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// low / high bf16 halves of a packed dword.
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));   // one v_cvt_pk_bf16_f32
}

// 2^x for softmax arguments (x <= 0, possibly hugely negative): the bare v_exp_f32, no denormal fix-up.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---- wave64 reductions -----------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Block-wide sum for blocks of up to 1024 threads. `scratch` = 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float t = (lane < nw) ? scratch[lane] : 0.f;
  t = wave_sum(t);
  __syncthreads();
  return t;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float t = (lane < nw) ? scratch[lane] : -INFINITY;
  t = wave_max(t);
  __syncthreads();
  return t;
}

struct __attribute__((aligned(16))) U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 ld16(const void* p) { return *reinterpret_cast<const U4*>(p); }
__device__ __forceinline__ void st16(void* p, const U4& v) { *reinterpret_cast<U4*>(p) = v; }

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace sgl_amd

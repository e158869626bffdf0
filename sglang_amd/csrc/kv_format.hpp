// KV pool addressing and element formats shared by the attention / store kernels.
//
// Reference layouts (/root/reference/python/sglang/srt/mem_cache/memory_pool.py):
//   NHD  k_buffer[layer] [slots, H_kv, D]                      (:2049-2060, the default)
//   HND  k_buffer[layer] [pages, H_kv, page_size, D]           (:2061-2117, `use_hnd`; slot = page * page_size + off)
// Element formats: bf16, or OCP e4m3 ("fp8_e4m3", :2364-2374: rows hold K / k_scale, the attention multiplies the
// scale back).  One formula addresses both layouts:
//   row(slot, h) = base + (slot >> page_shift) * page_stride + (slot & page_mask) * tok_stride + h * head_stride
// (NHD: page_shift = 0, so the middle term vanishes).  All strides are in BYTES.
#pragma once
#include "common.hpp"

namespace sgl_amd {

// The strides are 32-bit on purpose: row() is then one or two v_mad_u64_u32 (32 x 32 + 64) per gathered row instead of
// 64 x 64-bit multiplies (six quarter-rate integer multiplies each, ~600 clocks per KV tile in the extend kernel).
struct KvFormat {
  uint32_t page_stride, tok_stride, head_stride;
  int page_shift, page_mask;
  int fp8;
};

inline bool make_kv_format(KvFormat* f, int64_t slot_stride_elems, int num_kv_heads, int head_dim, int page_size, int hnd, int fp8) {
  const int es = fp8 ? 1 : 2;
  f->fp8 = fp8;
  if (!hnd) {
    if (slot_stride_elems * es >= (int64_t{1} << 32)) return false;
    f->page_shift = 0; f->page_mask = 0; f->tok_stride = 0;
    f->page_stride = static_cast<uint32_t>(slot_stride_elems * es);
    f->head_stride = static_cast<uint32_t>(head_dim * es);
    return true;
  }
  if (page_size < 1 || (page_size & (page_size - 1)) != 0) return false;
  if (static_cast<int64_t>(num_kv_heads) * page_size * head_dim * es >= (int64_t{1} << 32)) return false;
  int sh = 0;
  while ((1 << sh) < page_size) ++sh;
  f->page_shift = sh; f->page_mask = page_size - 1;
  f->tok_stride = static_cast<uint32_t>(head_dim * es);
  f->head_stride = static_cast<uint32_t>(page_size * head_dim * es);
  f->page_stride = static_cast<uint32_t>(num_kv_heads * page_size * head_dim * es);
  return true;
}

__device__ __forceinline__ const unsigned char* kv_row(const void* base, const KvFormat& f, int slot, int kvh) {
  // slot ids are non-negative: unsigned 32 x 32 -> 64 products
  const uint64_t off = static_cast<uint64_t>(static_cast<uint32_t>(slot) >> f.page_shift) * f.page_stride +
                       static_cast<uint64_t>(static_cast<uint32_t>(slot & f.page_mask)) * f.tok_stride +
                       static_cast<uint64_t>(static_cast<uint32_t>(kvh)) * f.head_stride;
  return static_cast<const unsigned char*>(base) + off;
}

// 8 consecutive e4m3 values -> 8 bf16 (exact: 3 significand bits fit), packed like a 16-byte bf16 load
__device__ __forceinline__ U4 fp8x8_to_bf16x8(uint2 w) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(w.x, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(w.x, true);
  const f32x2_t c = __builtin_amdgcn_cvt_pk_f32_fp8(w.y, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(w.y, true);
  U4 o;
  o.x = pack_bf2(a[0], a[1]); o.y = pack_bf2(b[0], b[1]); o.z = pack_bf2(c[0], c[1]); o.w = pack_bf2(d[0], d[1]);
  return o;
}

// 8 elements of a KV row starting at element e8 * 8, as bf16x8
template <bool FP8>
__device__ __forceinline__ U4 ld_kv8(const unsigned char* row, int e8) {
  if constexpr (FP8) return fp8x8_to_bf16x8(*reinterpret_cast<const uint2*>(row + e8 * 8));
  else return ld16(row + e8 * 16);
}

// 8 fp32 -> 8 e4m3 (round to nearest even, saturating at +-448 like a clamp before torch's .to(float8_e4m3fn))
__device__ __forceinline__ uint2 f32x8_to_fp8x8(const float* v, float inv_scale) {
  float c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = fminf(fmaxf(v[j] * inv_scale, -448.f), 448.f);
  uint2 o;
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w, true);
  o.x = static_cast<uint32_t>(w);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], w, true);
  o.y = static_cast<uint32_t>(w);
  return o;
}

// logit soft cap in the log2 domain: x = s * log2(e) with s the scaled score; returns cap * tanh(s / cap) * log2(e)
__device__ __forceinline__ float soft_cap_log2(float x, float cap_log2, float inv_cap_log2) {
  const float t = x * inv_cap_log2;                      // s / cap
  const float e = __expf(2.f * t);
  const float th = 1.f - 2.f / (e + 1.f);                // tanh(t), saturates cleanly for |t| large
  return cap_log2 * th;
}

}  // namespace sgl_amd

// Activation storage formats and the device helpers every kernel uses to touch them.
//   TA_FMT_F32   : NHWC float32.
//   TA_FMT_SPLIT : per pixel and 32-channel block, 32 bf16 `hi` (64 B) followed by 32 bf16 `lo` (64 B), x = hi + lo
//                  (hi = bf16_rne(x), lo = bf16_rne(x - hi)).  Same 4 bytes per element and the same block
//                  addressing as float32 (32 channels = 128 B), so tensor geometry, halos and DMA walks do not change;
//                  the bf16 MFMA operand fragments of the pipelined conv come straight out of LDS with no VALU.
#pragma once
#include <hip/hip_runtime.h>

enum { TA_FMT_F32 = 0, TA_FMT_SPLIT = 1 };

typedef float ta_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned ta_bf16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }

// 4 consecutive channels ch..ch+3 (ch % 4 == 0) of the pixel whose channel-0 address is `pix`
__device__ __forceinline__ ta_f32x4 ta_ld4(const float* pix, int ch, int fmt) {
  if (fmt == TA_FMT_F32) return *(const ta_f32x4*)(pix + ch);
  const char* b = (const char*)pix + ((ch >> 5) << 7) + ((ch & 31) << 1);
  const uint2 h = *(const uint2*)b, l = *(const uint2*)(b + 64);
  ta_f32x4 r;
  r[0] = __uint_as_float(h.x << 16) + __uint_as_float(l.x << 16);
  r[1] = __uint_as_float(h.x & 0xFFFF0000u) + __uint_as_float(l.x & 0xFFFF0000u);
  r[2] = __uint_as_float(h.y << 16) + __uint_as_float(l.y << 16);
  r[3] = __uint_as_float(h.y & 0xFFFF0000u) + __uint_as_float(l.y & 0xFFFF0000u);
  return r;
}

__device__ __forceinline__ void ta_st4(float* pix, int ch, int fmt, ta_f32x4 v) {
  if (fmt == TA_FMT_F32) {
    *(ta_f32x4*)(pix + ch) = v;
    return;
  }
  unsigned hi[4], lo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = ta_bf16_bits(v[e]);
    lo[e] = ta_bf16_bits(v[e] - __uint_as_float(hi[e] << 16));
  }
  char* b = (char*)pix + ((ch >> 5) << 7) + ((ch & 31) << 1);
  *(uint2*)b = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
  *(uint2*)(b + 64) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
}

__device__ __forceinline__ float ta_ld1(const float* pix, int ch, int fmt) {
  if (fmt == TA_FMT_F32) return pix[ch];
  const char* b = (const char*)pix + ((ch >> 5) << 7) + ((ch & 31) << 1);
  const unsigned h = *(const unsigned short*)b, l = *(const unsigned short*)(b + 64);
  return __uint_as_float(h << 16) + __uint_as_float(l << 16);
}

// Activation storage formats and the device helpers every kernel uses to touch them.
//   TA_FMT_F32     : NHWC float32.
//   TA_FMT_SPLIT   : per pixel and 32-channel block, 32 bf16 `hi` (64 B) followed by 32 bf16 `lo` (64 B), x = hi + lo
//                    (hi = bf16_rne(x), lo = bf16_rne(x - hi): 16 significant bits).  Same 4 bytes per element and the
//                    same block addressing as float32 (32 channels = 128 B), so tensor geometry, halos and DMA walks do
//                    not change; the MFMA operand fragments of the pipelined conv come straight out of LDS with no VALU.
//   TA_FMT_SPLIT16 : the same image with IEEE half words: hi = f16_rne(x), lo = f16_rne(x - hi): 22 significant bits
//                    for |x| in [2^-3, 65504] (lo goes subnormal below that: the absolute error stays <= 2^-25).
//                    |x| > 65504 does not fit: the conv epilogues that write this format raise the context's range
//                    flag (ta_conv_launch::range_flag) and the call fails with TA_E_RANGE instead of returning numbers.
//   TA_FMT_F16     : NHWC IEEE half, 2 bytes per element (the single-half arithmetic mode PREC_F16: operands carry 11 bits
//                    anyway, so there is no `lo` to keep).  A pixel is c halfs = c / 2 float slots: every stride the kernels
//                    take in floats is HALF the float32 tensor's, a 64-channel block is the 128 bytes a K slab row holds.
//                    Same range rule as TA_FMT_SPLIT16.  c % 64 == 0.
#pragma once
#include <hip/hip_runtime.h>

enum { TA_FMT_F32 = 0, TA_FMT_SPLIT = 1, TA_FMT_SPLIT16 = 2, TA_FMT_F16 = 3 };
#define TA_F16_MAX 65504.0f

typedef float ta_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned ta_bf16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ unsigned ta_f16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x); }
__device__ __forceinline__ float ta_f16_val(unsigned bits) { return (float)__builtin_bit_cast(_Float16, (unsigned short)bits); }

// two packed 16-bit words (element 0 in the low half) of a `hi` or `lo` plane -> float32
template <bool F16>
__device__ __forceinline__ void ta_unpack2(unsigned w, float& a, float& b) {
  if constexpr (F16) {
    a = ta_f16_val(w & 0xFFFFu);
    b = ta_f16_val(w >> 16);
  } else {
    a = __uint_as_float(w << 16);
    b = __uint_as_float(w & 0xFFFF0000u);
  }
}
// (x0, x1) -> packed hi word and packed lo word
template <bool F16>
__device__ __forceinline__ void ta_pack2(float x0, float x1, unsigned& hw, unsigned& lw) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  if constexpr (F16) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 h = __builtin_convertvector((f32x2){x0, x1}, f16x2);
    hw = __builtin_bit_cast(unsigned, h);
    const f32x2 r = {x0 - (float)h[0], x1 - (float)h[1]};
    lw = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
  } else {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    // two floats per v_cvt_pk_bf16_f32 (round to nearest even, same as the scalar conversion)
    hw = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x0, x1}, bf16x2));
    const f32x2 r = {x0 - __uint_as_float(hw << 16), x1 - __uint_as_float(hw & 0xFFFF0000u)};
    lw = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  }
}

// (x0, x1) -> two packed half floats (round to nearest even)
__device__ __forceinline__ unsigned ta_pack_half2(float x0, float x1) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x0, x1}, f16x2));
}

// byte offset of channel `ch`'s hi word inside a pixel of a split-format tensor (the lo word sits 64 B further)
__device__ __forceinline__ unsigned ta_split_chan(int ch) { return (unsigned)(((ch >> 5) << 7) + ((ch & 31) << 1)); }

// 4 consecutive channels ch..ch+3 (ch % 4 == 0) of the pixel whose channel-0 address is `pix`
__device__ __forceinline__ ta_f32x4 ta_ld4(const float* pix, int ch, int fmt) {
  if (fmt == TA_FMT_F32) return *(const ta_f32x4*)(pix + ch);
  if (fmt == TA_FMT_F16) {
    const uint2 w = *(const uint2*)((const char*)pix + 2 * ch);
    float v[4];
    ta_unpack2<true>(w.x, v[0], v[1]);
    ta_unpack2<true>(w.y, v[2], v[3]);
    return ta_f32x4{v[0], v[1], v[2], v[3]};
  }
  const char* b = (const char*)pix + ta_split_chan(ch);
  const uint2 h = *(const uint2*)b, l = *(const uint2*)(b + 64);
  float hv[4], lv[4];
  if (fmt == TA_FMT_SPLIT16) {
    ta_unpack2<true>(h.x, hv[0], hv[1]);
    ta_unpack2<true>(h.y, hv[2], hv[3]);
    ta_unpack2<true>(l.x, lv[0], lv[1]);
    ta_unpack2<true>(l.y, lv[2], lv[3]);
  } else {
    ta_unpack2<false>(h.x, hv[0], hv[1]);
    ta_unpack2<false>(h.y, hv[2], hv[3]);
    ta_unpack2<false>(l.x, lv[0], lv[1]);
    ta_unpack2<false>(l.y, lv[2], lv[3]);
  }
  return ta_f32x4{hv[0] + lv[0], hv[1] + lv[1], hv[2] + lv[2], hv[3] + lv[3]};
}

__device__ __forceinline__ void ta_st4(float* pix, int ch, int fmt, ta_f32x4 v) {
  if (fmt == TA_FMT_F32) {
    *(ta_f32x4*)(pix + ch) = v;
    return;
  }
  if (fmt == TA_FMT_F16) {
    *(uint2*)((char*)pix + 2 * ch) = make_uint2(ta_pack_half2(v[0], v[1]), ta_pack_half2(v[2], v[3]));
    return;
  }
  unsigned h0, h1, l0, l1;
  if (fmt == TA_FMT_SPLIT16) {
    ta_pack2<true>(v[0], v[1], h0, l0);
    ta_pack2<true>(v[2], v[3], h1, l1);
  } else {
    ta_pack2<false>(v[0], v[1], h0, l0);
    ta_pack2<false>(v[2], v[3], h1, l1);
  }
  char* b = (char*)pix + ta_split_chan(ch);
  *(uint2*)b = make_uint2(h0, h1);
  *(uint2*)(b + 64) = make_uint2(l0, l1);
}

__device__ __forceinline__ float ta_ld1(const float* pix, int ch, int fmt) {
  if (fmt == TA_FMT_F32) return pix[ch];
  if (fmt == TA_FMT_F16) return ta_f16_val(*(const unsigned short*)((const char*)pix + 2 * ch));
  const char* b = (const char*)pix + ta_split_chan(ch);
  const unsigned h = *(const unsigned short*)b, l = *(const unsigned short*)(b + 64);
  if (fmt == TA_FMT_SPLIT16) return ta_f16_val(h) + ta_f16_val(l);
  return __uint_as_float(h << 16) + __uint_as_float(l << 16);
}

// HBM-bound layer kernels (NHWC float32, 16-byte vector accesses, channel-fastest threads):
// depthwise 3x3 (+folded BN bias, ReLU), 2x2 max-pool, channel-slice copy, and the uint8 ->
// float pre-processing of the three wrappers.
#include <string.h>

#include "act_format.h"
#include "ta_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- depthwise 3x3 (retinaface/model.py:32-39,65-67) ---------------------------------------
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const ta_dw_launch p) {
  const int c4 = p.C >> 2;
  const size_t total = (size_t)p.N * p.Ho * p.Wo * c4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c4);
    size_t pix = i / c4;
    const int x = (int)(pix % p.Wo);
    pix /= p.Wo;
    const int y = (int)(pix % p.Ho);
    const int img = (int)(pix / p.Ho);
    const float* src = p.in + (size_t)img * p.in_img + (size_t)(y * p.stride) * p.in_row +
                       (size_t)(x * p.stride) * p.in_pix + p.in_off0;
    f32x4 acc = *(const f32x4*)(p.bias + cg * 4);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const f32x4 v = ta_ld4(src + (size_t)ky * p.in_row + (size_t)kx * p.in_pix, cg * 4, p.in_fmt);
        const f32x4 w = *(const f32x4*)(p.w + (ky * 3 + kx) * p.C + cg * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(v[e], w[e], acc[e]);
      }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = acc[e] > 0.f ? acc[e] : 0.f;
    }
    ta_st4(p.out + (size_t)img * p.out_img + (size_t)y * p.out_row + (size_t)x * p.out_pix + p.out_off0, cg * 4,
           p.out_fmt, acc);
  }
}

static int grid_for(size_t total, int block = 256) {
  size_t g = (total + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

int ta_launch_dwconv(ta_ctx* ctx, const ta_dw_launch& p) {
  const size_t total = (size_t)p.N * p.Ho * p.Wo * (p.C / 4);
  if (!total) return TA_OK;
  ta_prof_scope scope(ctx, 1, (double)total * 4 * 4 * 2);
  hipLaunchKernelGGL(dwconv3x3_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, p);
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

// ---- 2x2/2 max-pool, floor (openpose/model.py:8-13) ----------------------------------------
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* in, float* out, int in_fmt, int out_fmt, int N, int Ho, int Wo, int C,
                                                        int in_img, int in_row, int in_pix, int in_off0,
                                                        int out_img, int out_row, int out_pix, int out_off0) {
  const int c4 = C >> 2;
  const size_t total = (size_t)N * Ho * Wo * c4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c4);
    size_t pix = i / c4;
    const int x = (int)(pix % Wo);
    pix /= Wo;
    const int y = (int)(pix % Ho);
    const int img = (int)(pix / Ho);
    const float* s = in + (size_t)img * in_img + (size_t)(2 * y) * in_row + (size_t)(2 * x) * in_pix + in_off0;
    const f32x4 a = ta_ld4(s, cg * 4, in_fmt), b = ta_ld4(s + in_pix, cg * 4, in_fmt);
    const f32x4 c = ta_ld4(s + in_row, cg * 4, in_fmt), d = ta_ld4(s + in_row + in_pix, cg * 4, in_fmt);
    f32x4 m;
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
    ta_st4(out + (size_t)img * out_img + (size_t)y * out_row + (size_t)x * out_pix + out_off0, cg * 4, out_fmt, m);
  }
}

int ta_launch_maxpool(ta_ctx* ctx, const ta_tensor& in, const ta_tensor& out, int n) {
  const size_t total = (size_t)n * out.h * out.w * (out.c / 4);
  if (!total) return TA_OK;
  ta_prof_scope scope(ctx, 1, (double)total * 16 * 5);
  hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, in.dev, out.dev, in.fmt, out.fmt, n, out.h,
                     out.w, out.c, (int)((size_t)in.hp() * in.wp() * in.c), in.wp() * in.c, in.c,
                     (int)in.off(0, 0, 0), (int)((size_t)out.hp() * out.wp() * out.c), out.wp() * out.c, out.c,
                     (int)out.off(0, 0, 0));
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

// ---- channel-slice copy (concat by slices, openpose/model.py:120-136) ------------------------
__global__ __launch_bounds__(256) void copych_kernel(const float* in, float* out, int N, int H, int W, int ch,
                                                      int in_img, int in_row, int in_pix, int in_off0,
                                                      int out_img, int out_row, int out_pix, int out_off0) {
  const int c4 = ch >> 2;
  const size_t total = (size_t)N * H * W * c4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c4);
    size_t pix = i / c4;
    const int x = (int)(pix % W);
    pix /= W;
    const int y = (int)(pix % H);
    const int img = (int)(pix / H);
    *(f32x4*)(out + (size_t)img * out_img + (size_t)y * out_row + (size_t)x * out_pix + out_off0 + cg * 4) =
        *(const f32x4*)(in + (size_t)img * in_img + (size_t)y * in_row + (size_t)x * in_pix + in_off0 + cg * 4);
  }
}

int ta_launch_copych(ta_ctx* ctx, const ta_tensor& in, int in_ch, const ta_tensor& out, int out_ch, int ch, int n) {
  // raw 16-byte copies: with pre-split tensors the slice must be whole 32-channel blocks in the same format
  if (in.fmt != out.fmt || (in.fmt != TA_FMT_F32 && ((in_ch | out_ch | ch) & 31)))
    return ta_fail(ctx, TA_E_INVALID, "copych: incompatible tensor formats");
  const size_t total = (size_t)n * in.h * in.w * (ch / 4);
  if (!total) return TA_OK;
  ta_prof_scope scope(ctx, 1, (double)total * 16 * 2);
  hipLaunchKernelGGL(copych_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, in.dev, out.dev, n, in.h, in.w,
                     ch, (int)((size_t)in.hp() * in.wp() * in.c), in.wp() * in.c, in.c, (int)in.off(0, 0, 0) + in_ch,
                     (int)((size_t)out.hp() * out.wp() * out.c), out.wp() * out.c, out.c,
                     (int)out.off(0, 0, 0) + out_ch);
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

// ---- pre-processing --------------------------------------------------------------------------
// RETINAFACE : NHWC RGB uint8 -> BGR float 0..255            (retinaface/wrapper.py:144-146)
// OPENPOSE   : NHWC RGB uint8 -> RGB float x/255 - 0.5        (openpose/wrapper.py:116-122)
// ARCFACE    : NCHW BGR uint8 crops -> BGR float (x-127.5)*0.0078125 (arcface/model.py:88)
// Output: NHWC with 4 channels (4th = 0) in the interior of the halo-padded input tensor.
__global__ __launch_bounds__(256) void preprocess_kernel(int mode, const uint8_t* src, int N, int H, int W,
                                                          float* dst, int d_img, int d_row, int d_pix, int d_off0) {
  const size_t total = (size_t)N * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    size_t pix = i;
    const int x = (int)(pix % W);
    pix /= W;
    const int y = (int)(pix % H);
    const int img = (int)(pix / H);
    f32x4 v;
    if (mode == TA_PRE_ARCFACE_CROPS) {
      const size_t plane = (size_t)H * W;
      const uint8_t* s = src + (size_t)img * 3 * plane + (size_t)y * W + x;
      v[0] = ((float)s[0] - 127.5f) * 0.0078125f;
      v[1] = ((float)s[plane] - 127.5f) * 0.0078125f;
      v[2] = ((float)s[2 * plane] - 127.5f) * 0.0078125f;
    } else {
      const uint8_t* s = src + i * 3;
      if (mode == TA_PRE_RETINAFACE) {
        v[0] = (float)s[2];
        v[1] = (float)s[1];
        v[2] = (float)s[0];
      } else {
        v[0] = __fdiv_rn((float)s[0], 255.0f) - 0.5f;
        v[1] = __fdiv_rn((float)s[1], 255.0f) - 0.5f;
        v[2] = __fdiv_rn((float)s[2], 255.0f) - 0.5f;
      }
    }
    v[3] = 0.f;
    *(f32x4*)(dst + (size_t)img * d_img + (size_t)y * d_row + (size_t)x * d_pix + d_off0) = v;
  }
}

int ta_launch_preprocess(ta_ctx* ctx, int mode, const uint8_t* src_dev, int n, int h, int w, const ta_tensor& dst) {
  if (dst.c != 4 || dst.n < n || dst.h != h || dst.w != w)
    return ta_fail(ctx, TA_E_INVALID, "preprocess: destination tensor mismatch");
  const size_t total = (size_t)n * h * w;
  if (!total) return TA_OK;
  ta_prof_scope scope(ctx, 2, (double)total * (3 + 16));
  hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, mode, src_dev, n, h, w,
                     dst.dev, (int)((size_t)dst.hp() * dst.wp() * dst.c), dst.wp() * dst.c, dst.c,
                     (int)dst.off(0, 0, 0));
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

// ---- RetinaFace front (retinaface/wrapper.py:144-146 + model.py:60-67 + scales.0.0.conv_block) ------------------------
// uint8 RGB frame -> BGR float 0..255 -> conv3x3 s2 p1 (3 -> 8) + BN + ReLU -> depthwise 3x3 p1 (8) + BN + ReLU ->
// 1x1 (8 -> 16) + BN + ReLU, written once as a 16-channel float tensor at half resolution.  Unfused this is four
// launches that stream a 4-channel float copy of the frames and two 8-channel maps through HBM (0.37 ms per 32 frames at
// 416 x 739); here a workgroup builds an 18 x 18 x 8 tile of the stride-2 map in LDS (zero outside the map: the
// depthwise conv's own padding) and every thread finishes one output pixel.  All float32 FMAs, taps in (ky, kx, c) order.
// weights: [stem W 8x27 (o, c_bgr, ky, kx)] [stem b 8] [dw W 9x8 (tap, c)] [dw b 8] [pw W 16x8 (o, c)] [pw b 16] = 448 floats
#define RFS_T 16
#define RFS_IN (2 * (RFS_T + 2) + 1)          // input rows / columns under an (RFS_T + 2)^2 tile of the stride-2 map
struct rf_stem_weights {
  float v[448];                               // passed BY VALUE: wave-uniform reads become scalar loads, FMAs take SGPR operands
};
__global__ __launch_bounds__(256) void rf_stem_kernel(const uint8_t* frames, int H, int W, const rf_stem_weights wt, float* out, int Ho,
                                                       int Wo, int o_img, int o_row, int o_pix, int o_off0) {
  __shared__ uint8_t pix[RFS_IN * RFS_IN * 3 + 3];
  __shared__ float tile[(RFS_T + 2) * (RFS_T + 2) * 8];
  const int tid = threadIdx.x;
  const int img = blockIdx.z;
  const int ty0 = blockIdx.y * RFS_T, tx0 = blockIdx.x * RFS_T;
  const uint8_t* fr = frames + (size_t)img * H * W * 3;
  // input window: rows 2 (ty0 - 1) - 1 ..., zero outside the frame (the conv's padding)
  const int iy0 = 2 * (ty0 - 1) - 1, ix0 = 2 * (tx0 - 1) - 1;
  for (int i = tid; i < RFS_IN * RFS_IN * 3; i += 256) {
    const int r = i / (RFS_IN * 3), cb = i - r * (RFS_IN * 3);
    const int iy = iy0 + r, ix = ix0 + cb / 3;
    pix[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? fr[((size_t)iy * W + ix) * 3 + cb % 3] : (uint8_t)0;
  }
  __syncthreads();
  const float* sW = wt.v;          // [8][27]
  const float* sB = wt.v + 216;
  for (int idx = tid; idx < (RFS_T + 2) * (RFS_T + 2); idx += 256) {
    const int ty = idx / (RFS_T + 2), tx = idx - ty * (RFS_T + 2);
    const int sy = ty0 - 1 + ty, sx = tx0 - 1 + tx;
    float acc[8];
    if (sy < 0 || sy >= Ho || sx < 0 || sx >= Wo) {
#pragma unroll
      for (int o = 0; o < 8; ++o) acc[o] = 0.f;
    } else {
#pragma unroll
      for (int o = 0; o < 8; ++o) acc[o] = sB[o];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const uint8_t* px = pix + ((2 * ty + ky) * RFS_IN + 2 * tx + kx) * 3;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float v = (float)px[2 - c];                 // network channel c of BGR = frame channel 2 - c
#pragma unroll
            for (int o = 0; o < 8; ++o) acc[o] = __builtin_fmaf(v, sW[o * 27 + c * 9 + ky * 3 + kx], acc[o]);
          }
        }
#pragma unroll
      for (int o = 0; o < 8; ++o) acc[o] = acc[o] > 0.f ? acc[o] : 0.f;
    }
    *(f32x4*)(tile + idx * 8) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *(f32x4*)(tile + idx * 8 + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
  }
  __syncthreads();
  const int py = tid / RFS_T, px_ = tid - py * RFS_T;
  const int oy = ty0 + py, ox = tx0 + px_;
  if (oy >= Ho || ox >= Wo) return;
  const float* dW = wt.v + 224;    // [9][8]
  const float* dB = wt.v + 296;
  const float* pW = wt.v + 304;    // [16][8]
  const float* pB = wt.v + 432;
  float d[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) d[c] = dB[c];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float* t = tile + ((py + ky) * (RFS_T + 2) + px_ + kx) * 8;
      const f32x4 t0 = *(const f32x4*)t, t1 = *(const f32x4*)(t + 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        d[c] = __builtin_fmaf(t0[c], dW[(ky * 3 + kx) * 8 + c], d[c]);
        d[4 + c] = __builtin_fmaf(t1[c], dW[(ky * 3 + kx) * 8 + 4 + c], d[4 + c]);
      }
    }
#pragma unroll
  for (int c = 0; c < 8; ++c) d[c] = d[c] > 0.f ? d[c] : 0.f;
  float* o = out + (size_t)img * o_img + (size_t)oy * o_row + (size_t)ox * o_pix + o_off0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int oc = q * 4 + e;
      float a = pB[oc];
#pragma unroll
      for (int c = 0; c < 8; ++c) a = __builtin_fmaf(d[c], pW[oc * 8 + c], a);
      r[e] = a > 0.f ? a : 0.f;
    }
    *(f32x4*)(o + q * 4) = r;
  }
}

int ta_launch_rfstem(ta_ctx* ctx, const uint8_t* frames_dev, int n, int h, int w, const float* weights_host, const ta_tensor& out) {
  if (!frames_dev || !weights_host || out.c != 16 || out.fmt != TA_FMT_F32 || out.h != (h + 1) / 2 || out.w != (w + 1) / 2 || out.n < n)
    return ta_fail(ctx, TA_E_INVALID, "rfstem: destination tensor mismatch");
  if (n <= 0) return TA_OK;
  ta_prof_scope scope(ctx, 0, 2.0 * (216.0 + 128.0) * (double)n * out.h * out.w);   // the two dense convs (depthwise MACs are not counted anywhere)
  ctx->cur_flops = 2.0 * (216.0 + 128.0) * (double)n * out.h * out.w;
  ctx->note_kernel("rf_stem_kernel");
  rf_stem_weights wt;                                                   // 1.8 KB of kernel arguments
  memcpy(wt.v, weights_host, sizeof(wt.v));
  hipLaunchKernelGGL(rf_stem_kernel, dim3((out.w + RFS_T - 1) / RFS_T, (out.h + RFS_T - 1) / RFS_T, n), dim3(256), 0, ctx->stream,
                     frames_dev, h, w, wt, out.dev, out.h, out.w, (int)((size_t)out.hp() * out.wp() * out.c), out.wp() * out.c,
                     out.c, (int)out.off(0, 0, 0));
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

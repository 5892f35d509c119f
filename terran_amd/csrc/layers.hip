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
// 416 x 739).  All float32 FMAs, taps in (ky, kx, c) order.
//
// A workgroup finishes a 14 x 62 tile of the output.  The 448 weights are kernel arguments, i.e. scalar loads, and it is
// their latency that bounded the first version of this kernel (one pixel per thread, ~110 waits on a scalar load for 830
// FMAs, bytes addressed with an integer division each: 154 us per 32 frames at 416 x 739 with the vector ALU 10 % busy).
// Here every thread works on FOUR horizontally adjacent pixels per weight it fetches, the weights arrive in blocks of
// 32 .. 80 (real loops keep the compiler from fetching all 448 first and spilling them into lanes) and every FMA is one
// half of a v_pk_fma_f32: 73 us (PMC: 1.9 k vector instructions per wave, vector ALU busy half of the time; with two
// workgroups = 8 waves per CU -- 56 KB of LDS each -- the rest is latency the two cannot hide for each other):
//   1. the 33-row x 129-pixel window of the frame goes to LDS as the BYTES they are, copied with aligned dword loads
//      (bytes outside the frame = the conv's zero padding are masked to 0; each row keeps its own misalignment 0..3);
//   2. thread (ty, j) builds pixels 4j..4j+3 of row ty of the 16 x 64 x 8 tile of the stride-2 map from 3 x 27 window bytes
//      (4 x ds_read_b64 + 7 x v_alignbyte per row; zero outside the map: the depthwise conv's own padding);
//   3. thread (py, g) finishes output pixels 4g..4g+3 of row py: depthwise 3x3 from 3 x 6 tile pixels, then the 1x1;
//   4. the 14 x 62 x 16 results cross LDS once more so that a store instruction writes 1 KB of consecutive bytes (from the
//      registers a lane's 16 bytes are 256 bytes from its neighbour's: measured 33 us of the kernel's 98).
// The tile is kept as two planes of 4 channels with a 16-byte pad after every 4 pixels: a thread's 16-byte accesses are
// 80 bytes from its neighbour's -- conflict-free for the 16-lane groups LDS serves a b128 access in; the output staging
// XORs a thread's chunk index with its number for the same reason.
// weights as passed: [stem W 27x8 (tap = ky, kx, c_bgr; o)] [stem b 8] [dw W 9x8 (tap, c)] [dw b 8] [pw W 8x16 (c, o)] [pw b 16]
// = 448 floats (the blob holds the stem conv as (o, c, ky, kx) and the 1x1 as (o, c): the launcher transposes them).
// `frames` is 4-byte aligned and readable up to the next multiple of 4 beyond the last frame (frames_alloc rounds up).
typedef float rfs_f32x2 __attribute__((ext_vector_type(2)));
#define RFS_TY 14
#define RFS_TX 62
#define RFS_CY (RFS_TY + 2)                   // rows / columns of the stride-2 map under an output tile
#define RFS_CX (RFS_TX + 2)
#define RFS_IY (2 * RFS_CY + 1)               // frame rows under those
#define RFS_IXB ((2 * RFS_CX + 1) * 3)        // ... and bytes per row (387)
#define RFS_DW ((RFS_IXB + 3 + 3) / 4)        // aligned dwords that cover them at any misalignment (98)
#define RFS_PITCH 100                         // dwords per window row in LDS
#define RFS_TPITCH 84                         // 16-byte slots per tile row and plane: pixel x lives in slot x + x / 4
#define RFS_PIX_SLOTS (RFS_IY * RFS_PITCH / 4)
#define RFS_LDS_SLOTS (RFS_PIX_SLOTS + 2 * RFS_CY * RFS_TPITCH)
#define RFS_OROW (RFS_TX * 4)                 // 16-byte chunks per output tile row (248)
#define RFS_OPITCH 256                        // ... and slots per row of the output staging (16 threads x 16 chunks)
#define RFS_SLOTS (RFS_TY * RFS_OPITCH > RFS_LDS_SLOTS ? RFS_TY * RFS_OPITCH : RFS_LDS_SLOTS)
static_assert(RFS_IY * RFS_PITCH % 4 == 0, "tile stays 16-byte aligned");
struct rf_stem_weights {
  float v[448];                               // passed BY VALUE: wave-uniform reads become scalar loads, FMAs take SGPR operands
};
// FUSE (round 6): the NEXT block of the base -- depthwise 3x3 stride 2 (16) + ReLU -> 1x1 (16 -> 32) + ReLU (model.py:26-39,
// scales.0.0.sep_block + scales.0.1.conv_block) -- runs on the staged tile as well, so the 16-channel half-resolution map (210 MB
// per 32 frames at 640 x 640, written once and read once) never reaches HBM: the kernel writes the 32-channel QUARTER-resolution
// map.  A stride-2 output pixel (oy, ox) reads rows 2 oy - 1 .. 2 oy + 1 of the 16-channel map: the 14 x 62 tile starts at the ODD
// position (2 oy0 - 1, 2 ox0 - 1) and yields 6 x 30 outputs from 13 x 61 of its pixels (tiles step 12 x 60: 1.2 x the front's work
// for that map).  Pixels of the 16-channel map outside the map are staged as ZEROS (this conv's padding).  5. thread = output pixel:
// 9 taps x 16 channels from the staging (conflict-free: consecutive pixels are two staged pixels apart), the 1x1 in trips of 4
// output channels with the weights as scalar loads from `w2` ([9][16] dw, [16] bias, [16][32] 1x1 (c, oc), [32] bias);  6. the 6 x 30
// x 32 results cross LDS so that a store instruction writes 3.75 KB of consecutive bytes.
#define RFS_OY 6
#define RFS_OX 30
#define RFS_O2ROW (RFS_OX * 8)                // 16-byte chunks per row of the fused output tile
template <bool FUSE>
__global__ __launch_bounds__(256) void rf_stem_kernel(const uint8_t* frames, int H, int W, const rf_stem_weights wt, const float* __restrict__ w2,
                                                       float* out, int Ho, int Wo, int Ho2, int Wo2, int o_img, int o_row, int o_pix, int o_off0) {
  __shared__ __attribute__((aligned(16))) f32x4 lds4[RFS_SLOTS];       // 56 KB: two workgroups per CU
  unsigned* pix = (unsigned*)lds4;
  f32x4* tile = lds4 + RFS_PIX_SLOTS;
  const int tid = threadIdx.x;
  const int img = blockIdx.z;
  const int ty0 = FUSE ? 2 * RFS_OY * (int)blockIdx.y - 1 : (int)blockIdx.y * RFS_TY;
  const int tx0 = FUSE ? 2 * RFS_OX * (int)blockIdx.x - 1 : (int)blockIdx.x * RFS_TX;
  const int iy0 = 2 * (ty0 - 1) - 1, ix0 = 2 * (tx0 - 1) - 1;
  const int Wb = W * 3;
  const uint8_t* imgp = frames + (size_t)img * H * Wb;
  // misalignment of frame row 0's first window byte (ix0 >= -5: 24 + 3 ix0 >= 0)
  const int mis_img = (int)(((size_t)imgp + 24 + ix0 * 3) & 3);
  // 1. window rows as bytes: two rows per pass (waves 0-1 / 2-3), thread = one aligned dword.  All of a thread's loads
  //    are issued before the first one is waited for.  Only tiles on the frame's left / right edge mask bytes.
  const int wb3 = Wb & 3;
  {
    const int half = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int k = tid & 127;
    constexpr int NR = (RFS_IY + 1) / 2;
    unsigned v[NR];
    const bool x_inside = ix0 >= 0 && ix0 * 3 + RFS_IXB + 3 <= Wb;      // every dword of every row lies inside its frame row
    if (k < RFS_DW) {
      const int fbk = ix0 * 3 + 4 * k;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int r = half + 2 * i;
        const int iy = iy0 + r;
        const bool row_ok = r < RFS_IY && iy >= 0 && iy < H;          // wave-uniform
        const int iyc = row_ok ? iy : 0;
        const int mis = (mis_img + (iyc & 3) * wb3) & 3;               // of this row's first window byte
        const int fb0 = fbk - mis;                                     // frame-row byte index of this dword's byte 0
        const uint8_t* rowp = imgp + (size_t)iyc * Wb;                 // uniform; rowp + fb0 is 4-byte aligned
        v[i] = 0;
        if (row_ok && (x_inside || (fb0 > -4 && fb0 < Wb))) v[i] = *(const unsigned*)(rowp + fb0);
      }
      if (x_inside) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int r = half + 2 * i;
          if (r < RFS_IY) pix[r * RFS_PITCH + k] = v[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int r = half + 2 * i;
          if (r >= RFS_IY) continue;
          const int iy = iy0 + r;
          const bool row_ok = iy >= 0 && iy < H;
          const int mis = (mis_img + ((row_ok ? iy : 0) & 3) * wb3) & 3;
          const int fb0 = fbk - mis;
          int lo = -fb0, hi = Wb - fb0;                                // bytes [lo, hi) of the dword are inside the frame row
          lo = lo < 0 ? 0 : (lo > 4 ? 4 : lo);
          hi = hi > 4 ? 4 : (hi < 0 ? 0 : hi);
          const unsigned mlo = lo >= 4 ? 0u : 0xFFFFFFFFu << (8 * lo);
          const unsigned mhi = hi >= 4 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu << (8 * hi));
          pix[r * RFS_PITCH + k] = v[i] & mlo & mhi;
        }
      }
    }
  }
  __syncthreads();
  // 2. conv3x3 s2 (3 -> 8) + ReLU: thread = (row ty, pixels 4j .. 4j+3)
  {
    const float* sW = wt.v;          // [27][8]
    const float* sB = wt.v + 216;
    const int ty = tid >> 4, j = tid & 15;
    const int sy = ty0 - 1 + ty;
    const bool srow_ok = sy >= 0 && sy < Ho;
    rfs_f32x2 acc[4][4];                          // [pixel][pair of output channels]: v_pk_fma_f32 throughout
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[p][o] = rfs_f32x2{sB[2 * o], sB[2 * o + 1]};
#pragma nounroll
    for (int ky = 0; ky < 3; ++ky) {            // a real loop: 72 weights (scalar registers) per trip, see above
      const int r = 2 * ty + ky;
      const int iy = iy0 + r;
      const int mis = (mis_img + (((iy >= 0 && iy < H) ? iy : 0) & 3) * wb3) & 3;
      const unsigned* row = pix + r * RFS_PITCH + 6 * j;          // window bytes 24 j ... start `mis` bytes into this dword
      unsigned d[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint2 t = *(const uint2*)(row + 2 * q);
        d[2 * q] = t.x;
        d[2 * q + 1] = t.y;
      }
      float f[28];
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const unsigned u = __builtin_amdgcn_alignbyte(d[q + 1], d[q], (unsigned)mis);
        f[4 * q] = (float)(u & 255u);
        f[4 * q + 1] = (float)((u >> 8) & 255u);
        f[4 * q + 2] = (float)((u >> 16) & 255u);
        f[4 * q + 3] = (float)(u >> 24);
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float* w = sW + ((ky * 3 + kx) * 3 + c) * 8;
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const float v = f[(2 * p + kx) * 3 + 2 - c];        // network channel c of BGR = frame channel 2 - c
            const rfs_f32x2 vv = {v, v};
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[p][o] = __builtin_elementwise_fma(vv, rfs_f32x2{w[2 * o], w[2 * o + 1]}, acc[p][o]);
          }
        }
    }
    // zero outside the map (the depthwise conv's padding): only tiles on the map's border have such pixels
    const bool tile_inside = ty0 >= 1 && ty0 + RFS_CY - 1 <= Ho && tx0 >= 1 && tx0 + RFS_CX - 1 <= Wo;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int sx = tx0 - 1 + 4 * j + p;
      const float m = (tile_inside || (srow_ok && sx >= 0 && sx < Wo)) ? 0.f : -1.f;     // max(acc, 0) | min(.., 0) below
      f32x4 a, b;
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        a[2 * o] = fmaxf(acc[p][o][0], 0.f);
        a[2 * o + 1] = fmaxf(acc[p][o][1], 0.f);
        b[2 * o] = fmaxf(acc[p][2 + o][0], 0.f);
        b[2 * o + 1] = fmaxf(acc[p][2 + o][1], 0.f);
      }
      if (!tile_inside && m < 0.f) {
        a = f32x4{0.f, 0.f, 0.f, 0.f};
        b = a;
      }
      tile[ty * RFS_TPITCH + 5 * j + p] = a;
      tile[(RFS_CY + ty) * RFS_TPITCH + 5 * j + p] = b;
    }
  }
  __syncthreads();
  // 3. depthwise 3x3 (8) + ReLU, 1x1 (8 -> 16) + ReLU: thread = (row py, pixels 4g .. 4g+3)
  const int py = tid >> 4, g = tid & 15;
  const bool active = py < RFS_TY;
  f32x4 r[4][4];                                  // [channel quad][pixel]
  if (active) {
    const float* dW = wt.v + 224;    // [9][8]
    const float* dB = wt.v + 296;
    const float* pW = wt.v + 304;    // [16][8]
    const float* pB = wt.v + 432;
    rfs_f32x2 d[4][4];                            // [pixel][pair of channels]
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < 4; ++c) d[p][c] = rfs_f32x2{dB[2 * c], dB[2 * c + 1]};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      f32x4 ta[6], tb[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int x = 4 * g + i;                                   // <= 65: slot x + x / 4 <= 81 (64, 65: never written,
        ta[i] = tile[(py + ky) * RFS_TPITCH + x + (x >> 2)];       //  read only for pixels 62, 63 that are not stored)
        tb[i] = tile[(RFS_CY + py + ky) * RFS_TPITCH + x + (x >> 2)];
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float* w = dW + (ky * 3 + kx) * 8;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const f32x4 a = ta[p + kx], b = tb[p + kx];
          d[p][0] = __builtin_elementwise_fma(rfs_f32x2{a[0], a[1]}, rfs_f32x2{w[0], w[1]}, d[p][0]);
          d[p][1] = __builtin_elementwise_fma(rfs_f32x2{a[2], a[3]}, rfs_f32x2{w[2], w[3]}, d[p][1]);
          d[p][2] = __builtin_elementwise_fma(rfs_f32x2{b[0], b[1]}, rfs_f32x2{w[4], w[5]}, d[p][2]);
          d[p][3] = __builtin_elementwise_fma(rfs_f32x2{b[2], b[3]}, rfs_f32x2{w[6], w[7]}, d[p][3]);
        }
      }
    }
    float dr[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        dr[p][2 * c] = fmaxf(d[p][c][0], 0.f);
        dr[p][2 * c + 1] = fmaxf(d[p][c][1], 0.f);
      }
#pragma nounroll
    for (int q = 0; q < 4; ++q) {                 // output channels 4q .. 4q+3: a real loop, 32 weights per trip; pW is [c][oc] here
      rfs_f32x2 a[4][2];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) a[p][h] = rfs_f32x2{pB[4 * q + 2 * h], pB[4 * q + 2 * h + 1]};
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const rfs_f32x2 vv = {dr[p][c], dr[p][c]};
#pragma unroll
          for (int h = 0; h < 2; ++h)
            a[p][h] = __builtin_elementwise_fma(vv, rfs_f32x2{pW[c * 16 + 4 * q + 2 * h], pW[c * 16 + 4 * q + 2 * h + 1]}, a[p][h]);
        }
      f32x4 o4[4];
#pragma unroll
      for (int p = 0; p < 4; ++p)
        o4[p] = f32x4{fmaxf(a[p][0][0], 0.f), fmaxf(a[p][0][1], 0.f), fmaxf(a[p][1][0], 0.f), fmaxf(a[p][1][1], 0.f)};
      if (q == 0) {                               // wave-uniform: keeps r[][] in registers without unrolling the loop
#pragma unroll
        for (int p = 0; p < 4; ++p) r[0][p] = o4[p];
      } else if (q == 1) {
#pragma unroll
        for (int p = 0; p < 4; ++p) r[1][p] = o4[p];
      } else if (q == 2) {
#pragma unroll
        for (int p = 0; p < 4; ++p) r[2][p] = o4[p];
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) r[3][p] = o4[p];
      }
    }
  }
  __syncthreads();                                // every tile read is done: window + tile become the output staging
  // 4. staging: row py, chunk (16 bytes) c of thread g at 16 g + (c ^ g)
  if (active) {
    const bool row_in = !FUSE || (ty0 + py >= 0 && ty0 + py < Ho);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int mx = tx0 + 4 * g + p;
      const bool in_map = row_in && (!FUSE || (mx >= 0 && mx < Wo));       // FUSE: outside the map = the next conv's zero padding
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = p * 4 + q;
        lds4[py * RFS_OPITCH + 16 * g + (c ^ g)] = in_map ? r[q][p] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  __syncthreads();
  if constexpr (FUSE) {
    // 5. depthwise 3x3 stride 2 (16) + ReLU -> 1x1 (16 -> 32) + ReLU: thread = output pixel (orow, ocol) of the 6 x 30 tile
    const int orow = tid / RFS_OX, ocol = tid - orow * RFS_OX;
    const bool worker = tid < RFS_OY * RFS_OX;
    f32x4 o8[8];                                  // the pixel's 32 outputs
    if (worker) {
      const float* dW = w2;             // [9][16]
      const float* dB = w2 + 144;
      const float* pW = w2 + 160;       // [16][32] (c, oc)
      const float* pB = w2 + 672;
      float d[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) d[c] = dB[c];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int x = 2 * ocol + kx, gx = x >> 2, px = x & 3;
          const f32x4* src = lds4 + (2 * orow + ky) * RFS_OPITCH + 16 * gx;
          const float* w = dW + (ky * 3 + kx) * 16;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = src[(4 * px + q) ^ gx];
#pragma unroll
            for (int e = 0; e < 4; ++e) d[4 * q + e] = __builtin_fmaf(v[e], w[4 * q + e], d[4 * q + e]);
          }
        }
#pragma unroll
      for (int c = 0; c < 16; ++c) d[c] = fmaxf(d[c], 0.f);
#pragma nounroll
      for (int q = 0; q < 8; ++q) {               // output channels 4q .. 4q+3: a real loop, 64 scalar weights per trip
        float a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = pB[4 * q + e];
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = __builtin_fmaf(d[c], pW[c * 32 + 4 * q + e], a[e]);
        const f32x4 o4 = {fmaxf(a[0], 0.f), fmaxf(a[1], 0.f), fmaxf(a[2], 0.f), fmaxf(a[3], 0.f)};
        if (q == 0) o8[0] = o4;                   // wave-uniform: keeps o8[] in registers without unrolling the loop
        else if (q == 1) o8[1] = o4;
        else if (q == 2) o8[2] = o4;
        else if (q == 3) o8[3] = o4;
        else if (q == 4) o8[4] = o4;
        else if (q == 5) o8[5] = o4;
        else if (q == 6) o8[6] = o4;
        else o8[7] = o4;
      }
    }
    __syncthreads();                              // every read of the 16-channel staging is done: it becomes the output staging
    // 6. row orow, pixel ocol, chunk q at 8 ocol + (q ^ (ocol / 2 & 7)): a quarter wave's 16 consecutive pixels hit 16 different banks
    if (worker) {
#pragma unroll
      for (int q = 0; q < 8; ++q) lds4[orow * RFS_O2ROW + 8 * ocol + (q ^ ((ocol >> 1) & 7))] = o8[q];
    }
    __syncthreads();
    const int oy0 = RFS_OY * (int)blockIdx.y, ox0 = RFS_OX * (int)blockIdx.x;
    const int oc = tid >> 3;                      // pixel of the row this thread stores a chunk of
    if (tid < RFS_O2ROW && ox0 + oc < Wo2) {
      const int q = (tid & 7) ^ ((oc >> 1) & 7);  // the logical chunk that sits at this physical position
      float* dst = out + (size_t)img * o_img + o_off0 + (size_t)oy0 * o_row + (size_t)(ox0 + oc) * o_pix + q * 4;
      const int rows = Ho2 - oy0 < RFS_OY ? Ho2 - oy0 : RFS_OY;
#pragma unroll
      for (int row = 0; row < RFS_OY; ++row)
        if (row < rows) *(f32x4*)(dst + (size_t)row * o_row) = lds4[row * RFS_O2ROW + tid];
    }
    return;
  }
  if (tid < RFS_OROW && tx0 + (tid >> 2) < Wo) {          // thread = chunk of a row: constant offsets, one row per trip
    const int gg = tid >> 4;
    const f32x4* src = lds4 + 16 * gg + ((tid & 15) ^ gg);
    float* dst = out + (size_t)img * o_img + o_off0 + (size_t)ty0 * o_row + (size_t)(tx0 + (tid >> 2)) * o_pix + (tid & 3) * 4;
    const int rows = Ho - ty0 < RFS_TY ? Ho - ty0 : RFS_TY;
#pragma unroll
    for (int row = 0; row < RFS_TY; ++row)
      if (row < rows) *(f32x4*)(dst + (size_t)row * o_row) = src[row * RFS_OPITCH];
  }
}

int ta_launch_rfstem(ta_ctx* ctx, const uint8_t* frames_dev, int n, int h, int w, const float* weights_host, const float* w2_dev, const ta_tensor& out) {
  // w2_dev: device pointer to the 704 weights of the fused second block (FUSE: `out` is the 32-channel quarter-resolution map), or nullptr
  const bool fuse = w2_dev != nullptr;
  const int mh = (h + 1) / 2, mw = (w + 1) / 2;                        // the 16-channel half-resolution map
  const int oh = fuse ? (mh + 1) / 2 : mh, ow = fuse ? (mw + 1) / 2 : mw;
  if (!frames_dev || !weights_host || out.c != (fuse ? 32 : 16) || out.fmt != TA_FMT_F32 || out.h != oh || out.w != ow || out.n < n)
    return ta_fail(ctx, TA_E_INVALID, "rfstem: destination tensor mismatch");
  if (n <= 0) return TA_OK;
  if ((uintptr_t)frames_dev & 3)
    return ta_fail(ctx, TA_E_INVALID, "rfstem: the frames must be 4-byte aligned");
  // the dense convs (depthwise MACs are not counted anywhere)
  const double flops = 2.0 * (216.0 + 128.0) * (double)n * mh * mw + (fuse ? 2.0 * 512.0 * (double)n * oh * ow : 0.0);
  ta_prof_scope scope(ctx, 0, flops);
  ctx->cur_flops = flops;
  ctx->note_kernel(fuse ? "rf_stem_kernel<true>" : "rf_stem_kernel");
  rf_stem_weights wt;                                                   // 1.8 KB of kernel arguments
  memcpy(wt.v, weights_host, sizeof(wt.v));
  for (int o = 0; o < 8; ++o)                                           // (o, c, ky, kx) -> (ky, kx, c; o)
    for (int c = 0; c < 3; ++c)
      for (int t = 0; t < 9; ++t) wt.v[(t * 3 + c) * 8 + o] = weights_host[o * 27 + c * 9 + t];
  for (int oc = 0; oc < 16; ++oc)                                       // (oc, c) -> (c, oc)
    for (int c = 0; c < 8; ++c) wt.v[304 + c * 16 + oc] = weights_host[304 + oc * 8 + c];
  const int o_img = (int)((size_t)out.hp() * out.wp() * out.c), o_row = out.wp() * out.c, o_off0 = (int)out.off(0, 0, 0);
  if (fuse)
    hipLaunchKernelGGL(rf_stem_kernel<true>, dim3((ow + RFS_OX - 1) / RFS_OX, (oh + RFS_OY - 1) / RFS_OY, n), dim3(256), 0, ctx->stream,
                       frames_dev, h, w, wt, w2_dev, out.dev, mh, mw, oh, ow, o_img, o_row, out.c, o_off0);
  else
    hipLaunchKernelGGL(rf_stem_kernel<false>, dim3((mw + RFS_TX - 1) / RFS_TX, (mh + RFS_TY - 1) / RFS_TY, n), dim3(256), 0, ctx->stream,
                       frames_dev, h, w, wt, (const float*)nullptr, out.dev, mh, mw, mh, mw, o_img, o_row, out.c, o_off0);
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

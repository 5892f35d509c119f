// ArcFace pre/post-processing on the device:
//  * landmark-aligned 112x112 crop = PIL Image.transform(AFFINE, BILINEAR, fillcolor=0) restated in
//    float64 (arcface/wrapper.py:61-72; Pillow Geometry.c affine_transform + bilinear_filter32RGB:
//    pixel centre +0.5, reject outside [0,W)x[0,H), shift -0.5, floor, clipped taps, double lerp,
//    truncation to uint8), output BGR CHW like the reference crop;
//  * row-wise L2 normalisation (sklearn normalize, wrapper.py:176) as one wavefront per row;
//  * cosine distance matrix (examples/match.py:38).
// Compiled with -ffp-contract=off: the float64 warp must round exactly like Pillow's C code.
#include <string.h>

#include "ta_internal.h"

// face k is cut from the image at src[k].img (H x W x 3 uint8 RGB): the faces of one launch may come from several frame
// batches (ta_arcface_embed_faces_multi), each with its own size
struct ta_warp_src {
  const uint8_t* img;
  int32_t H, W;
};
__global__ __launch_bounds__(256) void warp_kernel(const ta_warp_src* src, const double* inv_affine, int n, uint8_t* crops) {
  const int total = n * 112 * 112;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int k = i / (112 * 112);
    const int rem = i - k * 112 * 112;
    const int oy = rem / 112, ox = rem - oy * 112;
    const double* a = inv_affine + 6 * k;
    const int H = src[k].H, W = src[k].W;
    const double xin = (double)ox + 0.5, yin = (double)oy + 0.5;
    double sx = a[0] * xin + a[1] * yin + a[2];
    double sy = a[3] * xin + a[4] * yin + a[5];
    uint8_t* dst = crops + (size_t)k * 3 * 112 * 112 + oy * 112 + ox;
    if (sx < 0.0 || sx >= (double)W || sy < 0.0 || sy >= (double)H) {
      dst[0] = 0;
      dst[112 * 112] = 0;
      dst[2 * 112 * 112] = 0;
      continue;
    }
    sx -= 0.5;
    sy -= 0.5;
    const int x = (int)floor(sx), y = (int)floor(sy);
    const double dx = sx - (double)x, dy = sy - (double)y;
    const int x0 = x < 0 ? 0 : (x < W ? x : W - 1);
    const int x1 = x + 1 < 0 ? 0 : (x + 1 < W ? x + 1 : W - 1);
    const int yc = y < 0 ? 0 : (y < H ? y : H - 1);
    const bool row2 = (y + 1 >= 0) && (y + 1 < H);
    const uint8_t* img = src[k].img;
    const uint8_t* r0 = img + (size_t)yc * W * 3;
    const uint8_t* r1 = img + (size_t)(row2 ? y + 1 : yc) * W * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double p00 = r0[x0 * 3 + c], p01 = r0[x1 * 3 + c];
      double v1 = p00 + (p01 - p00) * dx;
      double v2 = v1;
      if (row2) {
        const double p10 = r1[x0 * 3 + c], p11 = r1[x1 * 3 + c];
        v2 = p10 + (p11 - p10) * dx;
      }
      const double v = v1 + (v2 - v1) * dy;
      dst[(2 - c) * 112 * 112] = (uint8_t)v;       // RGB -> BGR planes, C truncation
    }
  }
}

// one wavefront per row
__global__ __launch_bounds__(256) void l2norm_kernel(float* x, int n, int dim) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  float* r = x + (size_t)row * dim;
  float s = 0.f;
  for (int i = lane; i < dim; i += 64) s += r[i] * r[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  float nrm = sqrtf(s);
  if (nrm == 0.f) nrm = 1.f;
  for (int i = lane; i < dim; i += 64) r[i] = r[i] / nrm;
}

__global__ __launch_bounds__(256) void cosine_kernel(const float* a, int na, const float* b, int nb, int dim, float* out) {
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (pair >= na * nb) return;
  const int i = pair / nb, j = pair - i * nb;
  const float* u = a + (size_t)i * dim;
  const float* v = b + (size_t)j * dim;
  double uv = 0, uu = 0, vv = 0;
  for (int k = lane; k < dim; k += 64) {
    const double x = u[k], y = v[k];
    uv += x * y;
    uu += x * x;
    vv += y * y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uv += __shfl_xor(uv, o);
    uu += __shfl_xor(uu, o);
    vv += __shfl_xor(vv, o);
  }
  if (lane == 0) out[pair] = (float)(1.0 - uv / (sqrt(uu) * sqrt(vv)));
}

static int embed_tail(ta_model* m, int n, int normalize, float* out) {
  ta_ctx* ctx = m->ctx;
  const ta_tensor& E = m->tensors[m->hdr.outputs[0]];
  if (E.unscale_dev || E.fmt != 0) return ta_fail(ctx, TA_E_INVALID, "arcface: the embedding tensor must be plain float32");
  if (normalize) {
    ta_prof_scope scope(ctx, 3, (double)n * 512 * 8);
    hipLaunchKernelGGL(l2norm_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, E.dev, n, 512);
    TA_HIP(ctx, hipGetLastError());
  }
  TA_HIP(ctx, hipMemcpyAsync(out, E.dev, (size_t)n * 512 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  TA_TRY(ta_range_enqueue(ctx));
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ta_range_check(ctx);                    // f16x3: TA_E_RANGE when an activation left the half-float range
}

extern "C" {

int ta_arcface_embed_crops(ta_model* m, const uint8_t* crops, int n, int normalize, float* out) {
  ta_enter(m ? m->ctx : nullptr);
  if (!m || n < 0 || (n > 0 && (!crops || !out))) return TA_E_INVALID;
  if (m->kind != TA_MODEL_ARCFACE) return ta_fail(m->ctx, TA_E_INVALID, "arcface: wrong model kind");
  if (n == 0) return TA_OK;
  TA_TRY(ta_model_forward_crops(m, crops, n));
  return embed_tail(m, n, normalize, out);
}

int ta_arcface_embed_faces_multi(ta_model* m, const ta_frames* const* frames, int n_sources, const int32_t* source_index,
                                 const int32_t* frame_index, const double* inv_affine, int n, int normalize, float* out,
                                 uint8_t* crops_out) {
  ta_enter(m ? m->ctx : nullptr);
  if (!m || !frames || n_sources <= 0 || n < 0 || (n > 0 && (!frame_index || !inv_affine || !out))) return TA_E_INVALID;
  ta_ctx* ctx = m->ctx;
  if (m->kind != TA_MODEL_ARCFACE) return ta_fail(ctx, TA_E_INVALID, "arcface: wrong model kind");
  if (n == 0) return TA_OK;
  for (int s = 0; s < n_sources; ++s)
    if (!frames[s] || frames[s]->ctx->device != ctx->device) return ta_fail(ctx, TA_E_INVALID, "arcface: frame batch %d is null or lives on another device", s);
  TA_TRY(ta_model_plan(m, n, 112, 112));
  const size_t crop_bytes = (size_t)n * 3 * 112 * 112;
  const size_t aff_off = (crop_bytes + 255) & ~(size_t)255;
  const size_t src_off = aff_off + (((size_t)n * 48 + 255) & ~(size_t)255);
  char* scr = nullptr;
  char* stage = nullptr;                          // pinned: the two tables travel in one async copy
  TA_TRY(ta_scratch(ctx, src_off + (size_t)n * sizeof(ta_warp_src), (void**)&scr));
  TA_TRY(ta_pinned(ctx, (src_off - aff_off) + (size_t)n * sizeof(ta_warp_src), (void**)&stage));
  memcpy(stage, inv_affine, (size_t)n * 48);
  ta_warp_src* srcs = (ta_warp_src*)(stage + (src_off - aff_off));
  for (int k = 0; k < n; ++k) {
    const int s = source_index ? source_index[k] : 0;
    if (s < 0 || s >= n_sources) return ta_fail(ctx, TA_E_INVALID, "arcface: source_index[%d] out of range", k);
    const ta_frames* f = frames[s];
    if (frame_index[k] < 0 || frame_index[k] >= f->n) return ta_fail(ctx, TA_E_INVALID, "arcface: frame_index[%d] out of range", k);
    srcs[k].img = f->dev + (size_t)frame_index[k] * f->h * f->w * 3;
    srcs[k].H = f->h;
    srcs[k].W = f->w;
  }
  TA_HIP(ctx, hipMemcpyAsync(scr + aff_off, stage, (src_off - aff_off) + (size_t)n * sizeof(ta_warp_src), hipMemcpyHostToDevice, ctx->stream));
  {
    ta_prof_scope scope(ctx, 2, (double)crop_bytes * 5);
    int g = (n * 112 * 112 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(warp_kernel, dim3(g), dim3(256), 0, ctx->stream, (const ta_warp_src*)(scr + src_off),
                       (const double*)(scr + aff_off), n, (uint8_t*)scr);
    TA_HIP(ctx, hipGetLastError());
  }
  TA_TRY(ta_launch_preprocess(ctx, TA_PRE_ARCFACE_CROPS, (const uint8_t*)scr, n, 112, 112, m->tensors[m->hdr.input_tensor]));
  TA_TRY(ta_model_run_ops(m));
  if (crops_out) TA_HIP(ctx, hipMemcpyAsync(crops_out, scr, crop_bytes, hipMemcpyDeviceToHost, ctx->stream));
  return embed_tail(m, n, normalize, out);
}

int ta_arcface_embed_faces(ta_model* m, const ta_frames* frames, const int32_t* frame_index, const double* inv_affine,
                           int n, int normalize, float* out, uint8_t* crops_out) {
  if (!frames) return TA_E_INVALID;
  return ta_arcface_embed_faces_multi(m, &frames, 1, nullptr, frame_index, inv_affine, n, normalize, out, crops_out);
}

int ta_cosine_distance(ta_ctx* ctx, const float* a, int na, const float* b, int nb, int dim, float* out) {
  ta_enter(ctx);
  if (!ctx || na < 0 || nb < 0 || dim <= 0) return TA_E_INVALID;
  if (na == 0 || nb == 0) return TA_OK;
  if (!a || !b || !out) return ta_fail(ctx, TA_E_INVALID, "cosine: null pointer");
  const size_t ab = ((size_t)na * dim * 4 + 255) & ~(size_t)255, bb = ((size_t)nb * dim * 4 + 255) & ~(size_t)255;
  char* scr = nullptr;
  TA_TRY(ta_scratch(ctx, ab + bb + (size_t)na * nb * 4, (void**)&scr));
  TA_HIP(ctx, hipMemcpyAsync(scr, a, (size_t)na * dim * 4, hipMemcpyHostToDevice, ctx->stream));
  TA_HIP(ctx, hipMemcpyAsync(scr + ab, b, (size_t)nb * dim * 4, hipMemcpyHostToDevice, ctx->stream));
  {
    ta_prof_scope scope(ctx, 3, (double)na * nb * dim * 8);
    hipLaunchKernelGGL(cosine_kernel, dim3((na * nb + 3) / 4), dim3(256), 0, ctx->stream, (const float*)scr, na,
                       (const float*)(scr + ab), nb, dim, (float*)(scr + ab + bb));
    TA_HIP(ctx, hipGetLastError());
  }
  TA_HIP(ctx, hipMemcpyAsync(out, scr + ab + bb, (size_t)na * nb * 4, hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return TA_OK;
}

}  // extern "C"

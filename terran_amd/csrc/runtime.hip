// Context, error text, scratch, HIP-event profiling, and device-resident uint8 frame batches
// (upload, cv2-style bilinear resize, zero-pad paste).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "ta_internal.h"

int ta_fail(ta_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

int ta_scratch(ta_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->scratch_bytes) {
    // the old block may still be read by queued kernels: drain first
    TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    size_t want = bytes + bytes / 4 + (1 << 20);
    TA_HIP(ctx, hipMalloc(&ctx->scratch, want));
    ctx->scratch_bytes = want;
  }
  *out = ctx->scratch;
  return TA_OK;
}

int ta_pinned(ta_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->pinned_bytes) {
    TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    size_t want = bytes + bytes / 4 + (1 << 16);
    TA_HIP(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
    ctx->pinned_bytes = want;
  }
  *out = ctx->pinned;
  return TA_OK;
}

// ---- f16x3 range flag -----------------------------------------------------------------------------
int ta_range_enqueue(ta_ctx* ctx) {
  TA_HIP(ctx, hipMemcpyAsync(ctx->range_flag_host, ctx->range_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  return TA_OK;
}

int ta_range_check(ta_ctx* ctx) {
  if (!*ctx->range_flag_host) return TA_OK;
  *ctx->range_flag_host = 0;
  TA_HIP(ctx, hipMemsetAsync(ctx->range_flag, 0, sizeof(int), ctx->stream));
  return ta_fail(ctx, TA_E_RANGE, "f16x3: an activation exceeded the half-float range (|x| > 65504); run this input in the f32 or bf16x3 mode");
}

// End of a task entry point whose post-processing returned `rc`: garbage maps of an overflowed network can make the grouping /
// selection fail in their own ways (TA_E_OVERFLOW with > 65535 "peaks", ...); the caller must then hear TA_E_RANGE -- the error it
// can act on (re-run on the exact-f32 twin) -- and the flag must not survive into the next, unrelated call.
int ta_range_finish(ta_ctx* ctx, int rc) {
  if (rc != TA_OK && rc != TA_E_CAPACITY) (void)hipStreamSynchronize(ctx->stream);    // error paths may have returned before their sync
  const int rr = ta_range_check(ctx);
  return rr != TA_OK ? rr : rc;
}

// ---- profiling ---------------------------------------------------------------------------------
static hipEvent_t get_event(ta_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

ta_prof_scope::ta_prof_scope(ta_ctx* c, int k, double w) : ctx(c), klass(k), work(w) {
  ctx->cur_kernel = nullptr;          // whatever launch noted its instance last belongs to an EARLIER scope (also while profiling was off)
  if (!ctx->profiling) return;
  a = get_event(ctx);
  b = get_event(ctx);
  if (a) (void)hipEventRecord(a, ctx->stream);
}

ta_prof_scope::~ta_prof_scope() {
  if (!ctx->profiling || !a || !b) {
    ctx->cur_kernel = nullptr;
    return;
  }
  (void)hipEventRecord(b, ctx->stream);
  ctx->pending.push_back({a, b, klass, ctx->cur_kernel});
  ctx->cur_kernel = nullptr;
  ctx->prof[klass].launches += 1;
  ctx->prof[klass].work += work;
}

void ta_drain_profile(ta_ctx* ctx) {
  if (ctx->pending.empty()) return;
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& pe : ctx->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
      ctx->prof[pe.klass].ms += ms;
      if (pe.kernel) ctx->kernel_ms[*pe.kernel] += ms;
    }
    ctx->event_pool.push_back(pe.a);
    ctx->event_pool.push_back(pe.b);
  }
  ctx->pending.clear();
}

extern "C" {

const char* ta_version(void) { return "terran_amd 0.1.0 (gfx950)"; }

int ta_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int ta_device_pci_bus_id(int device_id, char* out, int capacity) {
  if (!out || capacity < 16) return TA_E_INVALID;
  out[0] = 0;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return TA_E_DEVICE;
  if (hipDeviceGetPCIBusId(out, capacity, device_id) != hipSuccess) return TA_E_DEVICE;
  return TA_OK;
}

int ta_ctx_create(int device_id, ta_ctx** out) {
  if (!out) return TA_E_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return TA_E_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return TA_E_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return TA_E_DEVICE;   // this library is gfx950-only
  if (hipSetDevice(device_id) != hipSuccess) return TA_E_DEVICE;
  // Host threads BLOCK in their stream syncs instead of spinning: every task thread (detect / embed / pose per lane, 12+ per
  // GPU) ends each call in hipStreamSynchronize, and the runtime's default wait polls the completion signal from user space for
  // as long as the stream runs -- ~6 cores busy per rank doing nothing (0.085 CPU-seconds per 13.5 ms step, round 5).  With
  // this flag the runtime polls for ~100 us (short kernels still return at once) and then sleeps on the signal's interrupt.
  // TERRAN_AMD_SPIN_WAIT=1 keeps the spinning wait (latency probes).
  {
    const char* spin = getenv("TERRAN_AMD_SPIN_WAIT");
    if (!(spin && spin[0] && spin[0] != '0')) {
      (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
      (void)hipGetLastError();          // a runtime that refuses the flag on an already active device must not leave an error for the first launch check
    }
  }
  ta_ctx* ctx = new ta_ctx();
  ctx->device = device_id;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return TA_E_DEVICE;
  }
  (void)hipEventCreate(&ctx->t0);
  (void)hipEventCreate(&ctx->t1);
  const size_t flag_bytes = sizeof(int) * (TA_AMAX_SLOT0 + 2 * TA_AMAX_OPS);
  if (hipMalloc((void**)&ctx->range_flag, flag_bytes) != hipSuccess || hipMemset(ctx->range_flag, 0, flag_bytes) != hipSuccess ||
      hipHostMalloc((void**)&ctx->range_flag_host, sizeof(int), hipHostMallocDefault) != hipSuccess) {
    if (ctx->range_flag) (void)hipFree(ctx->range_flag);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return TA_E_DEVICE;
  }
  *ctx->range_flag_host = 0;
  // measurement aids for tools/ (the shipped path leaves both unset)
  if (const char* e = getenv("TA_CONV_PREFER")) ctx->conv_force = atoi(e);
#if defined(TA_TOOLS) || defined(TA_CONV_TRACE)   // timing ablations that give WRONG results (bits 0..7; bits 8.. = the workgroup the trace build stamps): only in a tools build (TA_EXTRA_FLAGS=-DTA_TOOLS), never in the shipped library
  if (const char* e = getenv("TA_CONV_PROBE")) ctx->conv_probe = atoi(e);
#endif
  *out = ctx;
  return TA_OK;
}

void ta_ctx_destroy(ta_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ta_drain_profile(ctx);
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->t0) (void)hipEventDestroy(ctx->t0);
  if (ctx->t1) (void)hipEventDestroy(ctx->t1);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->pose_wphase) (void)hipFree(ctx->pose_wphase);
  ta_pose_free_big(ctx);
  if (ctx->range_flag) (void)hipFree(ctx->range_flag);
  if (ctx->range_flag_host) (void)hipHostFree(ctx->range_flag_host);
  for (auto& e : ctx->frame_cache) (void)hipFree(e.second);
  for (int i = 0; i < 2; ++i) {
    if (ctx->side_stream[i]) {
      (void)hipStreamSynchronize(ctx->side_stream[i]);
      (void)hipStreamDestroy(ctx->side_stream[i]);
    }
    if (ctx->side_fork[i]) (void)hipEventDestroy(ctx->side_fork[i]);
    if (ctx->side_join[i]) (void)hipEventDestroy(ctx->side_join[i]);
  }
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* ta_last_error(const ta_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int ta_debug_range_check(ta_ctx* ctx) {
  ta_enter(ctx);
  if (!ctx) return TA_E_INVALID;
  TA_TRY(ta_range_enqueue(ctx));
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ta_range_check(ctx);
}

int ta_ctx_sync(ta_ctx* ctx) {
  ta_enter(ctx);
  if (!ctx) return TA_E_INVALID;
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return TA_OK;
}

int ta_profile_enable(ta_ctx* ctx, int on) {
  ta_enter(ctx);
  if (!ctx) return TA_E_INVALID;
  if (!on) ta_drain_profile(ctx);
  ctx->profiling = on != 0;
  return TA_OK;
}

int ta_profile_reset(ta_ctx* ctx) {
  ta_enter(ctx);
  if (!ctx) return TA_E_INVALID;
  ta_drain_profile(ctx);
  for (auto& p : ctx->prof) p = ta_prof_class();
  return TA_OK;
}

int ta_profile_read(ta_ctx* ctx, int klass, double* ms, int64_t* launches, double* work) {
  ta_enter(ctx);
  if (!ctx || klass < 0 || klass > 3) return TA_E_INVALID;
  ta_drain_profile(ctx);
  if (ms) *ms = ctx->prof[klass].ms;
  if (launches) *launches = ctx->prof[klass].launches;
  if (work) *work = ctx->prof[klass].work;
  return TA_OK;
}

int ta_timer_start(ta_ctx* ctx) {
  ta_enter(ctx);
  if (!ctx) return TA_E_INVALID;
  TA_HIP(ctx, hipEventRecord(ctx->t0, ctx->stream));
  return TA_OK;
}

int ta_timer_stop(ta_ctx* ctx, double* ms) {
  ta_enter(ctx);
  if (!ctx || !ms) return TA_E_INVALID;
  TA_HIP(ctx, hipEventRecord(ctx->t1, ctx->stream));
  TA_HIP(ctx, hipEventSynchronize(ctx->t1));
  float f = 0.f;
  TA_HIP(ctx, hipEventElapsedTime(&f, ctx->t0, ctx->t1));
  *ms = f;
  return TA_OK;
}

}  // extern "C"

// ---- frames ------------------------------------------------------------------------------------
// cv2.resize INTER_LINEAR for uint8 (OpenCV's classic two-pass fixed-point bilinear: 11-bit
// coefficients; horizontal pass into int, vertical pass ((b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2)>>2).
// Coefficient tables are built on the host exactly as oracle/facade.py does and uploaded.
__global__ __launch_bounds__(256) void resize_linear_kernel(const uint8_t* src, int N, int H, int W, uint8_t* dst,
                                                             int dh, int dw, const int32_t* xtab, const int32_t* ytab) {
  // xtab: [dw][4] = sx0, sx1, a0, a1 ; ytab: [dh][4] = y0, y1, b0, b1
  const size_t total = (size_t)N * dh * dw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    size_t pix = i;
    const int x = (int)(pix % dw);
    pix /= dw;
    const int y = (int)(pix % dh);
    const int img = (int)(pix / dh);
    const int4 xt = *(const int4*)(xtab + 4 * x);
    const int4 yt = *(const int4*)(ytab + 4 * y);
    const uint8_t* r0 = src + ((size_t)img * H + yt.x) * W * 3;
    const uint8_t* r1 = src + ((size_t)img * H + yt.y) * W * 3;
    uint8_t o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int h0 = r0[xt.x * 3 + c] * xt.z + r0[xt.y * 3 + c] * xt.w;
      const int h1 = r1[xt.x * 3 + c] * xt.z + r1[xt.y * 3 + c] * xt.w;
      int v = (((yt.z * (h0 >> 4)) >> 16) + ((yt.w * (h1 >> 4)) >> 16) + 2) >> 2;
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      o[c] = (uint8_t)v;
    }
    uint8_t* d = dst + i * 3;
    d[0] = o[0];
    d[1] = o[1];
    d[2] = o[2];
  }
}

__global__ __launch_bounds__(256) void paste_kernel(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw,
                                                     int top, int left) {
  const size_t total = (size_t)sh * sw * 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % 3);
    size_t pix = i / 3;
    const int x = (int)(pix % sw);
    const int y = (int)(pix / sw);
    dst[((size_t)(y + top) * dw + x + left) * 3 + c] = src[i];
  }
}

// Pillow ImagingResample pass along x: out[y][xx][c] = clip8((2^21 + sum_k src[y][xmin+k][c]*coef[xx][k]) >> 22).
// `transpose` = 1 runs the same pass along y (src/dst indexed [x][y]).
__global__ __launch_bounds__(256) void pil_resample_kernel(const uint8_t* src, int N, int H, int W, uint8_t* dst, int out_size,
                                                            const int32_t* bounds, const int32_t* coef, int ksize, int vertical) {
  const int oh = vertical ? out_size : H, ow = vertical ? W : out_size;
  const size_t total = (size_t)N * oh * ow;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t pix = i;
    const int x = (int)(pix % ow);
    pix /= ow;
    const int y = (int)(pix % oh);
    const int img = (int)(pix / oh);
    const int o = vertical ? y : x;
    const int lo = bounds[2 * o], cnt = bounds[2 * o + 1];
    const int32_t* k = coef + (size_t)o * ksize;
    int acc[3] = {1 << 21, 1 << 21, 1 << 21};
    for (int t = 0; t < cnt; ++t) {
      const uint8_t* s = vertical ? src + (((size_t)img * H + lo + t) * W + x) * 3 : src + (((size_t)img * H + y) * W + lo + t) * 3;
      acc[0] += s[0] * k[t];
      acc[1] += s[1] * k[t];
      acc[2] += s[2] * k[t];
    }
    uint8_t* d = dst + i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int v = acc[c] >> 22;
      d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}

#include <cmath>
static void axis_table(int src, int dst, bool clamp_frac, std::vector<int32_t>& tab) {
  // mirrors oracle/facade.py:_axis_coeffs/_to_short (OpenCV resize.cpp semantics)
  tab.resize((size_t)dst * 4);
  const double scale = 1.0 / ((double)dst / (double)src);
  for (int d = 0; d < dst; ++d) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f = f - (float)s;
    int s0, s1;
    if (clamp_frac) {
      if (s < 0) { f = 0.f; s = 0; }
      if (s >= src - 1) { f = 0.f; s = src - 1; }
      s0 = s;
      s1 = s + 1 < src ? s + 1 : src - 1;
    } else {
      s0 = s < 0 ? 0 : (s > src - 1 ? src - 1 : s);
      s1 = s + 1 < 0 ? 0 : (s + 1 > src - 1 ? src - 1 : s + 1);
    }
    auto to_short = [](float c) {
      float v = nearbyintf(c * 2048.0f);   // round half to even (default rounding mode)
      if (v > 32767.f) v = 32767.f;
      if (v < -32768.f) v = -32768.f;
      return (int32_t)v;
    };
    tab[4 * d + 0] = s0;
    tab[4 * d + 1] = s1;
    tab[4 * d + 2] = to_short(1.0f - f);
    tab[4 * d + 3] = to_short(f);
  }
}

extern "C" {

int ta_host_alloc(ta_ctx* ctx, size_t bytes, void** out) {
  ta_enter(ctx);
  if (!ctx || !out || bytes == 0) return ta_fail(ctx, TA_E_INVALID, "host_alloc: bad args");
  *out = nullptr;
  TA_HIP(ctx, hipHostMalloc(out, bytes, hipHostMallocDefault));
  return TA_OK;
}

void ta_host_free(ta_ctx* ctx, void* ptr) {
  ta_enter(ctx);
  if (ptr) (void)hipHostFree(ptr);
}

// parked frame buffers per context: a handful of recent sizes, bounded in bytes
#define TA_FRAME_CACHE_SLOTS 8
#define TA_FRAME_CACHE_BYTES ((size_t)2 << 30)
static int frames_alloc(ta_ctx* ctx, int n, int h, int w, bool zero, ta_frames** out);

int ta_frames_alloc(ta_ctx* ctx, int n, int h, int w, ta_frames** out) {
  ta_enter(ctx);
  return frames_alloc(ctx, n, h, w, true, out);
}

static int frames_alloc(ta_ctx* ctx, int n, int h, int w, bool zero, ta_frames** out) {
  if (!ctx || !out || n < 0 || h <= 0 || w <= 0) return ta_fail(ctx, TA_E_INVALID, "frames_alloc: bad shape");
  ta_frames* f = new ta_frames{ctx, n, h, w, nullptr};
  size_t bytes = ((size_t)n * h * w * 3 + 15) & ~(size_t)15;      // kernels read the frames in aligned dwords (rf_stem_kernel)
  if (bytes == 0) bytes = 16;
  {
    std::lock_guard<std::mutex> lock(ctx->frame_cache_mu);
    for (size_t i = 0; i < ctx->frame_cache.size(); ++i)
      if (ctx->frame_cache[i].first == bytes) {
        f->dev = (uint8_t*)ctx->frame_cache[i].second;
        ctx->frame_cache_bytes -= bytes;
        ctx->frame_cache.erase(ctx->frame_cache.begin() + i);
        break;
      }
  }
  if (!f->dev) {
    hipError_t e = hipMalloc((void**)&f->dev, bytes);
    if (e != hipSuccess) {
      delete f;
      return ta_fail(ctx, TA_E_DEVICE, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
  }
  f->cap = bytes;
  if (zero) TA_HIP(ctx, hipMemsetAsync(f->dev, 0, bytes, ctx->stream));   // only pad-merge canvases need zeros
  *out = f;
  return TA_OK;
}

int ta_frames_upload(ta_ctx* ctx, const uint8_t* nhwc_rgb, int n, int h, int w, ta_frames** out) {
  if (!ctx || !out) return TA_E_INVALID;
  ta_enter(ctx);            // hipMalloc below must land on the context's GPU, whatever the calling thread used last
  if (!nhwc_rgb && n > 0) return ta_fail(ctx, TA_E_INVALID, "frames_upload: null data");
  TA_TRY(frames_alloc(ctx, n, h, w, false, out));
  const size_t bytes = (size_t)n * h * w * 3;
  if (bytes) {
    TA_HIP(ctx, hipMemcpyAsync((*out)->dev, nhwc_rgb, bytes, hipMemcpyHostToDevice, ctx->stream));
    TA_HIP(ctx, hipStreamSynchronize(ctx->stream));   // caller's buffer may be reused immediately
  }
  return TA_OK;
}

int ta_frames_shape(const ta_frames* f, int* n, int* h, int* w) {
  if (!f) return TA_E_INVALID;
  if (n) *n = f->n;
  if (h) *h = f->h;
  if (w) *w = f->w;
  return TA_OK;
}

int ta_frames_download(const ta_frames* f, uint8_t* nhwc_rgb) {
  ta_enter(f ? f->ctx : nullptr);
  if (!f || !nhwc_rgb) return TA_E_INVALID;
  ta_ctx* ctx = f->ctx;
  const size_t bytes = (size_t)f->n * f->h * f->w * 3;
  if (bytes) {
    TA_HIP(ctx, hipMemcpyAsync(nhwc_rgb, f->dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return TA_OK;
}

void ta_frames_free(ta_frames* f) {
  ta_enter(f ? f->ctx : nullptr);
  if (!f) return;
  ta_ctx* ctx = f->ctx;
  (void)hipStreamSynchronize(ctx->stream);
  if (f->dev) {
    // every entry point is synchronous at its end, so the buffer is idle here: park it for the next batch
    std::vector<void*> evict;
    {
      std::lock_guard<std::mutex> lock(ctx->frame_cache_mu);
      ctx->frame_cache.emplace_back(f->cap, f->dev);
      ctx->frame_cache_bytes += f->cap;
      while (ctx->frame_cache.size() > TA_FRAME_CACHE_SLOTS || ctx->frame_cache_bytes > TA_FRAME_CACHE_BYTES) {
        ctx->frame_cache_bytes -= ctx->frame_cache.front().first;
        evict.push_back(ctx->frame_cache.front().second);
        ctx->frame_cache.erase(ctx->frame_cache.begin());
      }
    }
    for (void* p : evict) (void)hipFree(p);
  }
  delete f;
}

int ta_frames_resize(ta_ctx* ctx, const ta_frames* src, int dst_h, int dst_w, ta_frames** out) {
  ta_enter(ctx);
  if (!ctx || !src || !out || dst_h <= 0 || dst_w <= 0) return ta_fail(ctx, TA_E_INVALID, "frames_resize: bad args");
  TA_TRY(frames_alloc(ctx, src->n, dst_h, dst_w, false, out));
  std::vector<int32_t> xt, yt;
  axis_table(src->w, dst_w, true, xt);
  axis_table(src->h, dst_h, false, yt);
  void* scr = nullptr;
  const size_t tb = (xt.size() + yt.size()) * sizeof(int32_t);
  TA_TRY(ta_scratch(ctx, tb, &scr));
  void* pin = nullptr;
  TA_TRY(ta_pinned(ctx, tb, &pin));
  memcpy(pin, xt.data(), xt.size() * 4);
  memcpy((char*)pin + xt.size() * 4, yt.data(), yt.size() * 4);
  TA_HIP(ctx, hipMemcpyAsync(scr, pin, tb, hipMemcpyHostToDevice, ctx->stream));
  const size_t total = (size_t)src->n * dst_h * dst_w;
  if (total) {
    ta_prof_scope scope(ctx, 2, (double)total * 3 * 5);
    size_t g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(resize_linear_kernel, dim3((int)g), dim3(256), 0, ctx->stream, src->dev, src->n, src->h, src->w,
                       (*out)->dev, dst_h, dst_w, (const int32_t*)scr, (const int32_t*)scr + xt.size());
    TA_HIP(ctx, hipGetLastError());
  }
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));   // pinned/scratch tables are reused by later calls
  return TA_OK;
}

static double pil_bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// mirrors oracle/arcface_pre.py:_resample_coeffs (Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc)
static int pil_coeffs(int in_size, int out_size, std::vector<int32_t>& bounds, std::vector<int32_t>& coef) {
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  bounds.assign((size_t)out_size * 2, 0);
  coef.assign((size_t)out_size * ksize, 0);
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      k[x] = pil_bicubic((x + xmin - center + 0.5) * ss);
      ww += k[x];
    }
    for (int x = 0; x < xmax; ++x) {
      double v = ww != 0.0 ? k[x] / ww : k[x];
      coef[(size_t)xx * ksize + x] = v < 0 ? (int32_t)(-0.5 + v * (double)(1 << 22)) : (int32_t)(0.5 + v * (double)(1 << 22));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  return ksize;
}

static int pil_pass(ta_ctx* ctx, const uint8_t* src, int n, int h, int w, uint8_t* dst, int out_size, int vertical) {
  std::vector<int32_t> bounds, coef;
  const int ksize = pil_coeffs(vertical ? h : w, out_size, bounds, coef);
  const size_t bb = bounds.size() * 4, cb = coef.size() * 4;
  void* scr = nullptr;
  TA_TRY(ta_scratch(ctx, bb + cb, &scr));
  void* pin = nullptr;
  TA_TRY(ta_pinned(ctx, bb + cb, &pin));
  memcpy(pin, bounds.data(), bb);
  memcpy((char*)pin + bb, coef.data(), cb);
  TA_HIP(ctx, hipMemcpyAsync(scr, pin, bb + cb, hipMemcpyHostToDevice, ctx->stream));
  const size_t total = (size_t)n * (vertical ? out_size : h) * (vertical ? w : out_size);
  size_t g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(pil_resample_kernel, dim3((int)g), dim3(256), 0, ctx->stream, src, n, h, w, dst, out_size,
                     (const int32_t*)scr, (const int32_t*)((char*)scr + bb), ksize, vertical);
  TA_HIP(ctx, hipGetLastError());
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return TA_OK;
}

int ta_frames_resize_bicubic(ta_ctx* ctx, const ta_frames* src, int dst_h, int dst_w, ta_frames** out) {
  ta_enter(ctx);
  if (!ctx || !src || !out || dst_h <= 0 || dst_w <= 0) return ta_fail(ctx, TA_E_INVALID, "frames_resize_bicubic: bad args");
  // horizontal pass first, then vertical, each skipped when the size is unchanged (Pillow ImagingResample)
  ta_frames* tmp = nullptr;
  const uint8_t* cur = src->dev;
  int cw = src->w;
  if (dst_w != src->w) {
    TA_TRY(frames_alloc(ctx, src->n, src->h, dst_w, false, &tmp));
    TA_TRY(pil_pass(ctx, cur, src->n, src->h, src->w, tmp->dev, dst_w, 0));
    cur = tmp->dev;
    cw = dst_w;
  }
  int rc = frames_alloc(ctx, src->n, dst_h, dst_w, false, out);
  if (rc == TA_OK) {
    if (dst_h != src->h) {
      rc = pil_pass(ctx, cur, src->n, src->h, cw, (*out)->dev, dst_h, 1);
    } else {
      hipError_t e = hipMemcpyAsync((*out)->dev, cur, (size_t)src->n * dst_h * dst_w * 3, hipMemcpyDeviceToDevice, ctx->stream);
      if (e != hipSuccess) rc = ta_fail(ctx, TA_E_DEVICE, "copy failed: %s", hipGetErrorString(e));
      (void)hipStreamSynchronize(ctx->stream);
    }
  }
  if (tmp) ta_frames_free(tmp);
  return rc;
}

int ta_frames_paste(ta_ctx* ctx, const ta_frames* src, int src_index, ta_frames* dst, int dst_index, int top, int left) {
  ta_enter(ctx);
  if (!ctx || !src || !dst || src_index < 0 || src_index >= src->n || dst_index < 0 || dst_index >= dst->n ||
      top < 0 || left < 0 || top + src->h > dst->h || left + src->w > dst->w)
    return ta_fail(ctx, TA_E_INVALID, "frames_paste: bad args");
  const size_t total = (size_t)src->h * src->w * 3;
  size_t g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(paste_kernel, dim3((int)g), dim3(256), 0, ctx->stream,
                     src->dev + (size_t)src_index * src->h * src->w * 3, src->h, src->w,
                     dst->dev + (size_t)dst_index * dst->h * dst->w * 3, dst->h, dst->w, top, left);
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

}  // extern "C"

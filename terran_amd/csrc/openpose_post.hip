// OpenPose post-processing on the device (pose/openpose/wrapper.py:214-485):
//   1. x8 bicubic upsample of the 38 PAF + 19 heat-map channels (A=-0.75, align_corners=False,
//      no output clamp), tap accumulation in the exact order torch's CPU kernel uses:
//      fma(v3,w3, fma(v2,w2, fma(v0,w0, v1*w1))) horizontally on 4 rows, then vertically;
//   2. peak finding on parts 0..17: interior pixels, >= the 4 neighbours and >= 0.1, row-major
//      ordered compaction (ballot prefix sums), one workgroup per (image, part);
//   3. PAF line-integral scoring of every (src, dst) peak pair of a limb (10 truncated linspace
//      samples, length penalty, >=9 samples > 0.05 and score > 0), ordered compaction of the
//      accepted pairs, bitonic sort by descending score (ties: row-major pair order) and the
//      reference's greedy matching with its single shared `seen` set -- one workgroup per
//      (image, limb);
//   4. person assembly, filtering and keypoint output -- one wavefront per image, the search for
//      matching humans done lane-parallel, scores accumulated in float64 like the reference.
// Compiled with -ffp-contract=off; every float32 step rounds once, as oracle/openpose_post.py does.
#include <math.h>
#include <string.h>

#include "act_format.h"
#include "ta_internal.h"

// Working sizes of the FAST path (lists in LDS).  They are not limits: an image that outgrows one of them -- a saturated
// heat-map plateau does -- is re-run alone on the <BIG = true> instantiations of the same kernels, whose lists live in
// global memory and are sized from the counts the first pass reported (wrapper.py:235-262,335-366 have no caps), and
// network-resolution maps too large for the LDS staging (short sides beyond ~730 on 16:9 input) take those kernels for
// the whole batch, reading a planar float32 copy of the maps instead.
#define OP_MAXP 1024       // peaks per (image, part)
#define OP_MAXC 8192       // accepted pair candidates per (image, limb)
#define OP_MAXH 192        // humans under assembly per image
#define OP_NMID 10

__constant__ int c_map_idx[19][2] = {{31, 32}, {39, 40}, {33, 34}, {35, 36}, {41, 42}, {43, 44}, {19, 20}, {21, 22},
                                      {23, 24}, {25, 26}, {27, 28}, {29, 30}, {47, 48}, {49, 50}, {53, 54}, {51, 52},
                                      {55, 56}, {37, 38}, {45, 46}};
__constant__ int c_limbseq[19][2] = {{2, 3}, {2, 6}, {3, 4}, {4, 5}, {6, 7}, {7, 8}, {2, 9}, {9, 10}, {10, 11}, {2, 12},
                                      {12, 13}, {13, 14}, {2, 1}, {1, 15}, {15, 17}, {1, 16}, {16, 18}, {3, 17}, {6, 18}};

struct op_maps {
  const float* base;      // network-resolution maps, NHWC addressing
  int img, row, pix;      // element strides
  int off0;               // element offset of interior pixel (0,0), channel 0
  int paf_ch, hm_ch;      // first PAF / heat-map channel
  int fmt;                // TA_FMT_F32 or TA_FMT_SPLIT (act_format.h)
  int h, w;
  const float* unscale;   // nullptr, or per channel of the tensor the power of two every read is multiplied with (exact): the network
                          // stores channel c times 2^a[c] (ta_tensor::unscale_dev)
};

// ---- 1. bicubic x8 ------------------------------------------------------------------------------
// up: [N][57][8h][8w] planar; channel c<38 = PAF c, c>=38 = heat-map c-38.
// The source is NHWC (channels fastest), the result planar (x fastest): a workgroup stages the 5 source rows
// j-2..j+2 (border-clamped) x a strip of <= BC_TW columns (+2 halo each side) x 57 channels through LDS -- global
// reads with lanes along channels (coalesced), LDS image [row][channel][column] -- and then produces the 8 output
// rows 8j..8j+7 of that strip for all 57 planes with lanes along x: a task = (plane, phase row, source column q)
// makes the 8 outputs x = 8q..8q+7 from 4 rows x 5 columns of LDS and stores 32 contiguous bytes, so the
// 430 MB/step write stream is fully coalesced and every source value is read from HBM once per workgroup.
// Tap order per output is exactly ATen's: fma(v3,w3, fma(v2,w2, fma(v0,w0, v1*w1))) on each of the 4 rows, then
// the same chain vertically (weights depend only on the phase x%8 / y%8 and are exact in float32).
#define BC_TW 64
__global__ __launch_bounds__(256) void bicubic_kernel(const op_maps m, int N, float* up, const float* ywt,
                                                       const float* xphase /*[8][4]*/) {
  extern __shared__ float bc_sm[];
  const int H8 = m.h * 8, W8 = m.w * 8;
  const int j = blockIdx.x;                                  // source row
  const int q0 = blockIdx.y * BC_TW;                         // first source column of the strip
  const int img = blockIdx.z;
  const int tw = m.w - q0 < BC_TW ? m.w - q0 : BC_TW;        // columns produced
  const int lw = tw + 4;                                     // columns staged (q0-2 .. q0+tw+1, clamped)
  const int WP = lw | 1;                                     // odd pitch: the transposing LDS stores spread over banks
  const float* src = m.base + (size_t)img * m.img + m.off0;
  for (int idx = threadIdx.x; idx < 5 * lw * 57; idx += 256) {
    const int c = idx % 57;
    const int t = idx / 57;
    const int lc = t % lw, sr = t / lw;
    int row = j - 2 + sr;
    row = row < 0 ? 0 : (row > m.h - 1 ? m.h - 1 : row);
    int col = q0 - 2 + lc;
    col = col < 0 ? 0 : (col > m.w - 1 ? m.w - 1 : col);
    const int ch = c < 38 ? m.paf_ch + c : m.hm_ch + (c - 38);
    bc_sm[(sr * 57 + c) * WP + lc] = ta_ld1(src + (size_t)row * m.row + (size_t)col * m.pix, ch, m.fmt) * (m.unscale ? m.unscale[ch] : 1.0f);
  }
  __syncthreads();
  for (int task = threadIdx.x; task < 57 * 8 * tw; task += 256) {
    const int lq = task % tw;
    const int t = task / tw;
    const int ph = t & 7, c = t >> 3;
    const int y = 8 * j + ph;
    const float4 yw = *(const float4*)(ywt + 4 * y);
    const int so = ph < 4 ? 0 : 1;                           // rows j-2..j+1 for phases 0..3, j-1..j+2 for 4..7
    float v[4][5];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int k = 0; k < 5; ++k) v[r][k] = bc_sm[((so + r) * 57 + c) * WP + lq + k];
    float out[8];
#pragma unroll
    for (int px = 0; px < 8; ++px) {
      const float4 xw = *(const float4*)(xphase + 4 * px);
      const int o = px < 4 ? 0 : 1;                          // phases 0..3 start at column q-2, 4..7 at q-1
      float rows[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = v[r][o + 1] * xw.y;
        a = __builtin_fmaf(v[r][o + 0], xw.x, a);
        a = __builtin_fmaf(v[r][o + 2], xw.z, a);
        a = __builtin_fmaf(v[r][o + 3], xw.w, a);
        rows[r] = a;
      }
      float tt = rows[1] * yw.y;
      tt = __builtin_fmaf(rows[0], yw.x, tt);
      tt = __builtin_fmaf(rows[2], yw.z, tt);
      tt = __builtin_fmaf(rows[3], yw.w, tt);
      out[px] = tt;
    }
    float* dst = up + (((size_t)img * 57 + c) * H8 + y) * W8 + 8 * (q0 + lq);
    *(float4*)dst = make_float4(out[0], out[1], out[2], out[3]);
    *(float4*)(dst + 4) = make_float4(out[4], out[5], out[6], out[7]);
  }
}

static void cubic_axis(int in_size, std::vector<int>& idx, std::vector<float>& wts) {
  // mirrors oracle/openpose_post.py:_axis_plan / cubic_coeffs (ATen area_pixel_compute_source_index +
  // get_cubic_upsample_coefficients, A = -0.75); all values are exact in float32.
  const int out = in_size * 8;
  idx.resize((size_t)out * 4);
  wts.resize((size_t)out * 4);
  const float A = -0.75f;
  auto c1 = [&](float x) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; };
  auto c2 = [&](float x) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; };
  for (int d = 0; d < out; ++d) {
    const float real = 0.125f * ((float)d + 0.5f) - 0.5f;
    const float fl = floorf(real);
    const float t = real - fl;
    const int i0 = (int)fl;
    for (int k = 0; k < 4; ++k) {
      int v = i0 - 1 + k;
      v = v < 0 ? 0 : (v > in_size - 1 ? in_size - 1 : v);
      idx[(size_t)d * 4 + k] = v;
    }
    const float x2 = 1.f - t;
    wts[(size_t)d * 4 + 0] = c2(t + 1.f);
    wts[(size_t)d * 4 + 1] = c1(t);
    wts[(size_t)d * 4 + 2] = c1(x2);
    wts[(size_t)d * 4 + 3] = c2(x2 + 1.f);
  }
}

// ---- 2. peaks -------------------------------------------------------------------------------------
// The x8 maps are never materialised for grouping: peaks and PAF samples evaluate the bicubic interpolation on demand
// from the network-resolution maps staged in LDS, with exactly the tap order bicubic_kernel uses (which stays as the
// debug tap ta_bicubic_x8 and is pinned bitwise to F.interpolate): no 430 MB write + re-read per 32 frames at 184 x 327.
struct op_work {
  op_maps m;           // network-resolution maps (NHWC)
  const float* wphase; // [8][4] cubic weights per output phase (same table for x and y: they depend on the phase only)
  int N, H8, W8;
  int img_base;        // work-area image i is image img_base + i of the maps
  int maxp, maxc, maxh; // list capacities (OP_MAXP / OP_MAXC / OP_MAXH on the fast path)
  int* peak_cnt;       // [N][18]   (min(true count, maxp))
  int* peak_true;      // [N][18]   the number of peaks found, capped or not
  int* peak_yx;        // [N][18][maxp][2]
  float* peak_sc;      // [N][18][maxp]
  int* conn_cnt;       // [N][19]   (-1 = limb missing)
  int* pair_true;      // [N][19]   accepted pair candidates of the limb, capped or not
  int* conn_ij;        // [N][19][maxp][2]
  float* conn_sc;      // [N][19][maxp]
  int* overflow;       // [N] image i outgrew a list (its result is dropped here and recomputed with larger ones)
  double scale;
  int* out_cnt;        // [N]
  int* out_kp;         // [N][maxh][18][3]
  double* out_sc;      // [N][maxh]
  // <BIG> kernels only: lists in global memory, maps as a planar float32 copy
  const float* planar;          // [N][57][h][w]: PAF channels 0..37, heat-maps 38..56
  unsigned long long* g_cand;   // [N][18][cand_pitch], cand_pitch = maxp rounded up to a power of two (bitonic sort in place)
  int cand_pitch;
  unsigned long long* g_keys;   // [N][19][maxc rounded up to a power of two]; nullptr = limbs_kernel only counts (pair_true)
  unsigned* g_seen;             // [N][19][(maxp + 31) / 32]
  double* g_humans;             // [N][maxh][20]
  int keys_pitch;               // elements per (image, limb) of g_keys
};

// ATen's 4-tap accumulation (see bicubic_kernel): fma(v3,w3, fma(v2,w2, fma(v0,w0, v1*w1)))
__device__ __forceinline__ float op_chain(float v0, float v1, float v2, float v3, const float4 w) {
  float a = v1 * w.y;
  a = __builtin_fmaf(v0, w.x, a);
  a = __builtin_fmaf(v2, w.z, a);
  a = __builtin_fmaf(v3, w.w, a);
  return a;
}

__device__ __forceinline__ int op_clamp(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// value of the x8 map at (Y, X) from the low-resolution map `sm` (h x w, row-major, in LDS): horizontal chains on the four
// source rows, then the vertical chain -- bit for bit what bicubic_kernel writes at that pixel
__device__ __forceinline__ float op_up_at(const float* sm, int h, int w, int Y, int X, const float4* wph) {
  const int j = Y >> 3, py = Y & 7, q = X >> 3, px = X & 7;
  const int r0 = j - (py < 4 ? 2 : 1), c0 = q - (px < 4 ? 2 : 1);
  const float4 xw = wph[px], yw = wph[py];
  const int c[4] = {op_clamp(c0, w - 1), op_clamp(c0 + 1, w - 1), op_clamp(c0 + 2, w - 1), op_clamp(c0 + 3, w - 1)};
  float rows[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float* s = sm + op_clamp(r0 + r, h - 1) * w;
    rows[r] = op_chain(s[c[0]], s[c[1]], s[c[2]], s[c[3]], xw);
  }
  return op_chain(rows[0], rows[1], rows[2], rows[3], yw);
}

// One workgroup per (image, part).  A thread takes one source cell (j, q): from its 5 x 5 neighbourhood it builds the
// 10 x 10 patch of x8 values around the cell's 8 x 8 output block (the block plus the one-pixel ring the 4-neighbour test
// needs: the ring pixels belong to phases 7 / 0 of the adjacent cells, which read the same 5 x 5 values), tests the 64
// block pixels, and appends the peaks to an LDS list; the list is then sorted by pixel index = the reference's row-major
// `nonzero` order (wrapper.py:241-262).
#define PK_T 512
// planar float32 copy of the 57 network-resolution maps (the <BIG> kernels read it instead of staging maps in LDS)
__global__ __launch_bounds__(256) void op_planar_kernel(const op_maps m, int img_base, int N, float* out) {
  const size_t cells = (size_t)m.h * m.w, total = (size_t)N * 57 * cells;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cell = (int)(i % cells);
    const int c = (int)((i / cells) % 57), img = (int)(i / (cells * 57));
    const float* src = m.base + (size_t)(img_base + img) * m.img + m.off0 + (size_t)(cell / m.w) * m.row + (size_t)(cell % m.w) * m.pix;
    out[i] = ta_ld1(src, c < 38 ? m.paf_ch + c : m.hm_ch + (c - 38), m.fmt) * (m.unscale ? m.unscale[c < 38 ? m.paf_ch + c : m.hm_ch + (c - 38)] : 1.0f);
  }
}

template <bool BIG>
__global__ __launch_bounds__(PK_T) void peaks_kernel(const op_work w) {
  extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
  const int h = w.m.h, wd = w.m.w;
  const int part = blockIdx.x % 18, img = blockIdx.x / 18;
  const int tid = threadIdx.x;
  const int maxp = BIG ? w.maxp : OP_MAXP;
  const float* sm;
  float4* wph;
  unsigned long long* cand;
  __shared__ int s_count;
  if constexpr (BIG) {
    wph = (float4*)psm;
    sm = w.planar + ((size_t)img * 57 + 38 + part) * h * wd;
    cand = w.g_cand + ((size_t)img * 18 + part) * w.cand_pitch;
  } else {
    float* smw = (float*)psm;                                        // h * wd
    wph = (float4*)(psm + (((size_t)h * wd * 4 + 15) & ~(size_t)15));   // 8
    cand = (unsigned long long*)(wph + 8);                           // OP_MAXP
    const float* src = w.m.base + (size_t)(w.img_base + img) * w.m.img + w.m.off0;
    for (int i = tid; i < h * wd; i += PK_T) smw[i] = ta_ld1(src + (size_t)(i / wd) * w.m.row + (size_t)(i % wd) * w.m.pix, w.m.hm_ch + part, w.m.fmt) * (w.m.unscale ? w.m.unscale[w.m.hm_ch + part] : 1.0f);
    sm = smw;
  }
  if (tid < 8) wph[tid] = ((const float4*)w.wphase)[tid];
  if (tid == 0) s_count = 0;
  __syncthreads();
  const int H8 = w.H8, W8 = w.W8;
  for (int cell = tid; cell < h * wd; cell += PK_T) {
    const int j = cell / wd, q = cell - j * wd;
    float s[5][5];
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
      for (int k = 0; k < 5; ++k) s[r][k] = sm[op_clamp(j - 2 + r, h - 1) * wd + op_clamp(q - 2 + k, wd - 1)];
    // patch column xx (0..9) = output column 8 q - 1 + xx: phase (xx + 7) & 7, source columns q - 2 + o .. with o = xx >= 5
    float hz[5][10];
#pragma unroll
    for (int xx = 0; xx < 10; ++xx) {
      const float4 xw = wph[(xx + 7) & 7];
      const int o = xx >= 5 ? 1 : 0;
#pragma unroll
      for (int r = 0; r < 5; ++r) hz[r][xx] = op_chain(s[r][o], s[r][o + 1], s[r][o + 2], s[r][o + 3], xw);
    }
    float up[3][10];                                                // three consecutive patch rows
#pragma unroll
    for (int yy = 0; yy < 10; ++yy) {
      const float4 yw = wph[(yy + 7) & 7];
      const int o = yy >= 5 ? 1 : 0;
#pragma unroll
      for (int xx = 0; xx < 10; ++xx) up[yy % 3][xx] = op_chain(hz[o][xx], hz[o + 1][xx], hz[o + 2][xx], hz[o + 3][xx], yw);
      if (yy >= 2) {                                               // rows yy-2, yy-1, yy are there: test row yy-1
        const int Y = 8 * j + yy - 2;
        if (Y >= 1 && Y <= H8 - 2) {
          const float* a = up[(yy - 2) % 3];
          const float* b = up[(yy - 1) % 3];
          const float* c = up[yy % 3];
#pragma unroll
          for (int xx = 1; xx <= 8; ++xx) {
            const int X = 8 * q + xx - 1;
            const float v = b[xx];
            if (X >= 1 && X <= W8 - 2 && v >= a[xx] && v >= b[xx - 1] && v >= c[xx] && v >= b[xx + 1] && v >= 0.1f) {
              const int pos = atomicAdd(&s_count, 1);
              if (pos < maxp) cand[pos] = ((unsigned long long)(unsigned)(Y * W8 + X) << 32) | __float_as_uint(v);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  int C = s_count;
  if (tid == 0) w.peak_true[img * 18 + part] = C;
  if (C > maxp) {
    if (tid == 0) atomicExch(w.overflow + img, 1);
    C = maxp;
  }
  int P = 1;                                       // (the list's storage is a power of two >= maxp: cand_pitch)
  while (P < C) P <<= 1;
  for (int i = C + tid; i < P; i += PK_T) cand[i] = ~0ull;
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1)
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      for (int i = tid; i < P; i += PK_T) {
        const int ixj = i ^ jj;
        if (ixj > i) {
          const unsigned long long a = cand[i], b = cand[ixj];
          if ((a > b) == ((i & k) == 0)) {
            cand[i] = b;
            cand[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  int* yx = w.peak_yx + ((size_t)img * 18 + part) * maxp * 2;
  float* sc = w.peak_sc + ((size_t)img * 18 + part) * maxp;
  for (int i = tid; i < C; i += PK_T) {
    const unsigned long long key = cand[i];
    const int pix = (int)(unsigned)(key >> 32);
    yx[i * 2] = pix / W8;
    yx[i * 2 + 1] = pix - (pix / W8) * W8;
    sc[i] = __uint_as_float((unsigned)(key & 0xFFFFFFFFull));
  }
  if (tid == 0) w.peak_cnt[img * 18 + part] = C;
}

// ---- 3. limb scoring + greedy matching --------------------------------------------------------------
__device__ __forceinline__ int lin_trunc(float a, float b, float step, int i) {
  const float v = i < OP_NMID / 2 ? a + step * (float)i : b - step * (float)(OP_NMID - 1 - i);
  return (int)truncf(v);
}

// BIG: lists in global memory; w.g_keys == nullptr makes it a counting pass (pair_true only: sizes the keys of the real pass)
template <bool BIG>
__global__ __launch_bounds__(256) void limbs_kernel(const op_work w) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int limb = blockIdx.x % 19, img = blockIdx.x / 19;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int maxp = BIG ? w.maxp : OP_MAXP, maxc = BIG ? w.maxc : OP_MAXC;
  unsigned long long* keys;
  unsigned* seen;
  int* wave_tot;
  float4* wph;
  const float *smx, *smy;
  const int mh = w.m.h, mw = w.m.w;
  const int chx = c_map_idx[limb][0] - 19, chy = c_map_idx[limb][1] - 19;      // PAF channel pair of the limb
  if constexpr (BIG) {
    wave_tot = (int*)smem;                                              // 4 + 1 (+ 3 pad)
    wph = (float4*)(wave_tot + 8);
    keys = w.g_keys ? w.g_keys + ((size_t)img * 19 + limb) * w.keys_pitch : nullptr;
    seen = w.g_seen + ((size_t)img * 19 + limb) * ((maxp + 31) / 32);
    smx = w.planar + ((size_t)img * 57 + chx) * mh * mw;
    smy = w.planar + ((size_t)img * 57 + chy) * mh * mw;
  } else {
    keys = (unsigned long long*)smem;                                   // OP_MAXC
    seen = (unsigned*)(smem + (size_t)OP_MAXC * 8);                     // OP_MAXP bits
    wave_tot = (int*)(seen + OP_MAXP / 32);
    wph = (float4*)(wave_tot + 8);                                      // 8 phase weight sets
    float* sx = (float*)(wph + 8);                                      // the limb's two PAF channels at network resolution
    float* sy = sx + mh * mw;
    const float* srcm = w.m.base + (size_t)(w.img_base + img) * w.m.img + w.m.off0;
    for (int i = tid; i < mh * mw; i += 256) {
      const float* px = srcm + (size_t)(i / mw) * w.m.row + (size_t)(i % mw) * w.m.pix;
      sx[i] = ta_ld1(px, w.m.paf_ch + chx, w.m.fmt) * (w.m.unscale ? w.m.unscale[w.m.paf_ch + chx] : 1.0f);
      sy[i] = ta_ld1(px, w.m.paf_ch + chy, w.m.fmt) * (w.m.unscale ? w.m.unscale[w.m.paf_ch + chy] : 1.0f);
    }
    smx = sx;
    smy = sy;
  }
  int& s_base = wave_tot[4];
  const int ks = c_limbseq[limb][0] - 1, kd = c_limbseq[limb][1] - 1;
  const int ns = w.peak_cnt[img * 18 + ks], nd = w.peak_cnt[img * 18 + kd];
  int* conn_ij = w.conn_ij + ((size_t)img * 19 + limb) * maxp * 2;
  float* conn_sc = w.conn_sc + ((size_t)img * 19 + limb) * maxp;
  if (ns == 0 || nd == 0) {
    if (tid == 0) {
      w.conn_cnt[img * 19 + limb] = -1;     // "missing limb" (wrapper.py:293-296)
      w.pair_true[img * 19 + limb] = 0;
    }
    return;
  }
  const int* syx = w.peak_yx + ((size_t)img * 18 + ks) * maxp * 2;
  const int* dyx = w.peak_yx + ((size_t)img * 18 + kd) * maxp * 2;
  const float half_h = (float)(0.5 * (double)w.H8);
  if (tid < 8) wph[tid] = ((const float4*)w.wphase)[tid];
  if (tid == 0) s_base = 0;
  __syncthreads();
  const unsigned total = (unsigned)ns * (unsigned)nd;       // the launcher keeps ns, nd <= 65535
  for (unsigned t0 = 0; t0 < total; t0 += 256) {
    const unsigned t = t0 + tid;
    bool ok = false;
    float reg = 0.f;
    if (t < total) {
      const int i = (int)(t / (unsigned)nd), j = (int)(t - (unsigned)i * (unsigned)nd);
      const int sy = syx[i * 2], sx = syx[i * 2 + 1], ty = dyx[j * 2], tx = dyx[j * 2 + 1];
      const float dyf = (float)(ty - sy), dxf = (float)(tx - sx);
      // correctly rounded float sqrt (v_sqrt_f32 alone is 1 ulp): go through float64; the argument is an integer
      const float norm = (float)sqrt((double)(dyf * dyf + dxf * dxf));
      const float uy = __fdiv_rn(dyf, norm), ux = __fdiv_rn(dxf, norm);
      const float ay = (float)sy, by = (float)ty, ax = (float)sx, bx = (float)tx;
      const float stepy = __fdiv_rn(by - ay, (float)(OP_NMID - 1)), stepx = __fdiv_rn(bx - ax, (float)(OP_NMID - 1));
      float tot = 0.f;
      int cnt = 0;
#pragma unroll
      for (int k = 0; k < OP_NMID; ++k) {
        const int yy = lin_trunc(ay, by, stepy, k), xx = lin_trunc(ax, bx, stepx, k);
        const float mid = op_up_at(smx, w.m.h, w.m.w, yy, xx, wph) * ux + op_up_at(smy, w.m.h, w.m.w, yy, xx, wph) * uy;
        tot = k == 0 ? mid : tot + mid;
        cnt += mid > 0.05f ? 1 : 0;
      }
      const float pen = fminf(__fdiv_rn(half_h, norm) - 1.0f, 0.0f);
      reg = __fdiv_rn(tot, (float)OP_NMID) + pen;
      ok = (cnt > 8) && (reg > 0.0f);
    }
    const unsigned long long bal = __ballot(ok);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wv] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int k = 0; k < wv; ++k) off += wave_tot[k];
    if (ok && (!BIG || keys)) {
      const int pos = off + before;
      if (pos < maxc) keys[pos] = ((unsigned long long)(0xFFFFFFFFu - __float_as_uint(reg)) << 32) | (unsigned)t;
    }
    __syncthreads();
    if (tid == 0) s_base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
  }
  int C = s_base;
  if (tid == 0) w.pair_true[img * 19 + limb] = C;
  if (BIG && !keys) return;                         // counting pass
  if (C > maxc) {
    if (tid == 0) atomicExch(w.overflow + img, 1);
    C = maxc;
  }
  int P = 1;
  while (P < C) P <<= 1;
  for (int i = C + tid; i < P; i += 256) keys[i] = ~0ull;
  for (int i = tid; i < (maxp + 31) / 32; i += 256) seen[i] = 0;
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += 256) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], b = keys[ixj];
          if ((a > b) == ((i & k) == 0)) {
            keys[i] = b;
            keys[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  // greedy matching (wrapper.py:335-359): ONE `seen` set for source and destination indices; the
  // length check precedes the insertions.
  if (tid == 0) {
    const int lim = ns < nd ? ns : nd;
    int n = 0;
    for (int c = 0; c < C; ++c) {
      const unsigned long long key = keys[c];
      const unsigned t = (unsigned)(key & 0xFFFFFFFFull);
      const int i = (int)(t / (unsigned)nd), j = (int)(t - (unsigned)i * (unsigned)nd);
      if (((seen[i >> 5] >> (i & 31)) & 1u) || ((seen[j >> 5] >> (j & 31)) & 1u)) continue;
      conn_ij[n * 2] = i;
      conn_ij[n * 2 + 1] = j;
      conn_sc[n] = __uint_as_float(0xFFFFFFFFu - (unsigned)(key >> 32));
      ++n;
      if (n >= lim) break;
      seen[i >> 5] |= 1u << (i & 31);
      seen[j >> 5] |= 1u << (j & 31);
    }
    w.conn_cnt[img * 19 + limb] = n;
  }
}

// ---- 4. assembly -------------------------------------------------------------------------------------
// humans[h][0..17] = global peak id or -1; [18] = score sum; [19] = keypoint count (all float64, as the
// reference's numpy array).  One wavefront per image; lane-parallel search for matching humans.
template <bool BIG>
__global__ __launch_bounds__(64) void assemble_kernel(const op_work w) {
  __shared__ double humans_lds[BIG ? 1 : OP_MAXH][20];
  __shared__ int offs[19];
  const int img = blockIdx.x, lane = threadIdx.x;
  const int maxp = BIG ? w.maxp : OP_MAXP, maxh = BIG ? w.maxh : OP_MAXH;
  double (*humans)[20];
  if constexpr (BIG) humans = (double (*)[20])(w.g_humans + (size_t)img * maxh * 20);
  else humans = humans_lds;
  if (lane == 0) {
    int o = 0;
    for (int p = 0; p < 18; ++p) {
      offs[p] = o;
      o += w.peak_cnt[img * 18 + p];
    }
    offs[18] = o;
  }
  __syncthreads();
  int nh = 0;
  bool over = false;
  for (int limb = 0; limb < 19; ++limb) {
    const int nc = w.conn_cnt[img * 19 + limb];
    if (nc < 0) continue;
    const int ks = c_limbseq[limb][0] - 1, kd = c_limbseq[limb][1] - 1;
    const int* cij = w.conn_ij + ((size_t)img * 19 + limb) * maxp * 2;
    const float* csc = w.conn_sc + ((size_t)img * 19 + limb) * maxp;
    const float* psc_d = w.peak_sc + ((size_t)img * 18 + kd) * maxp;
    const float* psc_s = w.peak_sc + ((size_t)img * 18 + ks) * maxp;
    for (int c = 0; c < nc; ++c) {
      const int i = cij[c * 2], j = cij[c * 2 + 1];
      const double a = (double)(offs[ks] + i), b = (double)(offs[kd] + j);
      const double s = (double)csc[c];
      const double peak_b = (double)psc_d[j];
      // matched humans in ascending index order (up to two matter)
      int first = -1, second = -1, nmatch = 0;
      for (int h0 = 0; h0 < nh; h0 += 64) {
        const int h = h0 + lane;
        const bool mt = h < nh && (humans[h][ks] == a || humans[h][kd] == b);
        unsigned long long bal = __ballot(mt);
        while (bal) {
          const int l = __ffsll((long long)bal) - 1;
          bal &= bal - 1;
          if (nmatch == 0) first = h0 + l;
          else if (nmatch == 1) second = h0 + l;
          ++nmatch;
        }
      }
      if (nmatch == 1) {
        if (lane == 0 && humans[first][kd] != b) {
          humans[first][kd] = b;
          humans[first][19] += 1.0;
          humans[first][18] += peak_b + s;
        }
      } else if (nmatch == 2) {
        bool overlap = false;
        if (lane < 18) overlap = humans[first][lane] >= 0.0 && humans[second][lane] >= 0.0;
        const bool any = __ballot(overlap) != 0ull;
        if (!any) {
          if (lane < 18) humans[first][lane] += humans[second][lane] + 1.0;
          if (lane == 18) humans[first][18] = (humans[first][18] + humans[second][18]) + s;
          if (lane == 19) humans[first][19] += humans[second][19];
          __syncthreads();
          // np.delete(humans, second): shift the tail down by one row
          for (int h = second; h + 1 < nh; ++h) {
            double v = 0.0;
            if (lane < 20) v = humans[h + 1][lane];
            __syncthreads();
            if (lane < 20) humans[h][lane] = v;
            __syncthreads();
          }
          --nh;
        } else if (lane == 0) {
          humans[first][kd] = b;
          humans[first][19] += 1.0;
          humans[first][18] += peak_b + s;
        }
      } else if (nmatch == 0 && limb < 17) {
        if (nh < maxh) {
          if (lane < 18) humans[nh][lane] = -1.0;
          __syncthreads();
          if (lane == 0) {
            humans[nh][ks] = a;
            humans[nh][kd] = b;
            humans[nh][19] = 2.0;
            humans[nh][18] = ((0.0 + (double)psc_s[i]) + peak_b) + s;
          }
          ++nh;
        } else {
          over = true;
        }
      }
      __syncthreads();
    }
  }
  const bool truncated = w.overflow[img] != 0;      // peaks / limbs of this image were cut short by the kernels before
  if (over && lane == 0) atomicExch(w.overflow + img, 1);
  over = over || truncated;
  // filter + keypoints (wrapper.py:470-483, 37-90)
  int kept = 0;
  for (int h = 0; h < nh; ++h) {
    const double cnt = humans[h][19], tot = humans[h][18];
    if (cnt < 4.0 || tot / cnt < 0.4) continue;
    int* kp = w.out_kp + ((size_t)img * maxh + kept) * 54;
    if (lane < 18) {
      const double pid = humans[h][lane];
      int x = 0, y = 0, pr = 0;
      if ((int)pid != -1) {
        const int id = (int)pid;
        int part = 0;
        while (part < 17 && id >= offs[part + 1]) ++part;
        const int* yx = w.peak_yx + (((size_t)img * 18 + part) * maxp + (id - offs[part])) * 2;
        y = (int)((double)yx[0] / w.scale);
        x = (int)((double)yx[1] / w.scale);
        pr = 1;
      }
      kp[lane * 3] = x;
      kp[lane * 3 + 1] = y;
      kp[lane * 3 + 2] = pr;
    }
    if (lane == 0) w.out_sc[(size_t)img * maxh + kept] = tot / cnt;
    ++kept;
  }
  if (lane == 0) w.out_cnt[img] = over ? 0 : kept;   // a truncated image reports nothing (flagged in w.overflow)
}

__global__ __launch_bounds__(256) void op_gather_kernel(const op_work w, int* o_kp, double* o_sc) {
  const int img = blockIdx.x;
  int base = 0;
  for (int i = 0; i < img; ++i) base += w.out_cnt[i];
  const int K = w.out_cnt[img];
  for (int t = threadIdx.x; t < K * 54; t += blockDim.x) o_kp[(size_t)base * 54 + t] = w.out_kp[(size_t)img * w.maxh * 54 + t];
  for (int t = threadIdx.x; t < K; t += blockDim.x) o_sc[base + t] = w.out_sc[(size_t)img * w.maxh + t];
}

// ---- host side ---------------------------------------------------------------------------------------
static int upload_axis_tables(ta_ctx* ctx, int h, int w, char* dev, size_t* sizes) {
  std::vector<int> yi, xi;
  std::vector<float> yw, xw;
  cubic_axis(h, yi, yw);
  cubic_axis(w, xi, xw);
  void* pin = nullptr;
  const size_t by = yi.size() * 4, bx = xi.size() * 4;
  TA_TRY(ta_pinned(ctx, 2 * (by + bx), &pin));
  char* p = (char*)pin;
  memcpy(p, yi.data(), by);
  memcpy(p + by, yw.data(), by);
  memcpy(p + 2 * by, xi.data(), bx);
  memcpy(p + 2 * by + bx, xw.data(), bx);
  TA_HIP(ctx, hipMemcpyAsync(dev, pin, 2 * (by + bx), hipMemcpyHostToDevice, ctx->stream));
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  sizes[0] = by;
  sizes[1] = bx;
  return TA_OK;
}

static size_t rup256(size_t b) { return (b + 255) & ~(size_t)255; }

// The 8 phase weight sets of the x8 cubic (A = -0.75, align_corners = False): they depend on the output phase only.
static void phase_weights(float out[32]) {
  std::vector<int> idx;
  std::vector<float> wts;
  cubic_axis(2, idx, wts);                       // 16 outputs: entries 8..15 are one interior period
  memcpy(out, wts.data() + 32, 32 * sizeof(float));
}


// x8 upsample alone, into host memory (debug tap ta_bicubic_x8)
static int op_upsample_dev(ta_ctx* ctx, const op_maps& m, int N, float* up_out_host) {
  const int H8 = m.h * 8, W8 = m.w * 8;
  const size_t up_bytes = rup256((size_t)N * 57 * H8 * W8 * 4);
  const size_t tab_bytes = rup256((size_t)(H8 + W8) * 32);
  char* scr = nullptr;
  TA_TRY(ta_scratch(ctx, up_bytes + tab_bytes, (void**)&scr));
  size_t ts[2];
  TA_TRY(upload_axis_tables(ctx, m.h, m.w, scr + up_bytes, ts));
  const float* ywt = (const float*)(scr + up_bytes + ts[0]);
  const float* xwt = (const float*)(scr + up_bytes + 2 * ts[0] + ts[1]);
  float* up = (float*)scr;
  if (N > 65535 || m.h > 65535) return ta_fail(ctx, TA_E_INVALID, "openpose: batch too large for one upsample launch");
  const int strips = (m.w + BC_TW - 1) / BC_TW;
  const int lw = (m.w < BC_TW ? m.w : BC_TW) + 4;
  const size_t lds = (size_t)5 * 57 * (lw | 1) * sizeof(float);
  TA_SET_LDS_ATTR(ctx, bicubic_kernel, (size_t)5 * 57 * ((BC_TW + 4) | 1) * sizeof(float));
  // the x weights depend only on the phase x % 8: entries 8..15 of the table are an interior period
  hipLaunchKernelGGL(bicubic_kernel, dim3(m.h, strips, N), dim3(256), lds, ctx->stream, m, N, up, ywt, xwt + (m.w >= 2 ? 32 : 0));
  TA_HIP(ctx, hipGetLastError());
  TA_HIP(ctx, hipMemcpyAsync(up_out_host, up, (size_t)N * 57 * H8 * W8 * 4, hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return TA_OK;
}

// ---- work areas -------------------------------------------------------------------------------------
static int pow2_at_least(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Lays the lists of `N` images out in one device block (base == nullptr: size only).  big: the <BIG> kernels' extra
// lists (candidates / seen bitmaps / humans in global memory) and the planar map copy; keys come separately (their size
// is known only after the counting pass of a re-run).
static size_t layout_work(op_work& w, char* base, int N, bool big, size_t cells) {
  size_t off = 0;
  auto carve = [&](size_t b) {
    char* p = base ? base + off : nullptr;
    off += rup256(b);
    return p;
  };
  const size_t P = (size_t)w.maxp, H = (size_t)w.maxh;
  w.peak_cnt = (int*)carve((size_t)N * 18 * 4);
  w.peak_true = (int*)carve((size_t)N * 18 * 4);
  w.peak_yx = (int*)carve((size_t)N * 18 * P * 8);
  w.peak_sc = (float*)carve((size_t)N * 18 * P * 4);
  w.conn_cnt = (int*)carve((size_t)N * 19 * 4);
  w.pair_true = (int*)carve((size_t)N * 19 * 4);
  w.conn_ij = (int*)carve((size_t)N * 19 * P * 8);
  w.conn_sc = (float*)carve((size_t)N * 19 * P * 4);
  w.overflow = (int*)carve((size_t)N * 4);
  w.out_cnt = (int*)carve((size_t)N * 4);
  w.out_kp = (int*)carve((size_t)N * H * 54 * 4);
  w.out_sc = (double*)carve((size_t)N * H * 8);
  if (big) {
    w.cand_pitch = pow2_at_least(w.maxp);
    w.planar = (const float*)carve((size_t)N * 57 * cells * 4);
    w.g_cand = (unsigned long long*)carve((size_t)N * 18 * w.cand_pitch * 8);
    w.g_seen = (unsigned*)carve((size_t)N * 19 * ((P + 31) / 32) * 4);
    w.g_humans = (double*)carve((size_t)N * H * 20 * 8);
  }
  return off;
}

static int launch_planar(ta_ctx* ctx, const op_work& w, int N) {
  const size_t total = (size_t)N * 57 * w.m.h * w.m.w;
  size_t g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  ta_prof_scope scope(ctx, 3, (double)total * 8);
  hipLaunchKernelGGL(op_planar_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, w.m, w.img_base, N, (float*)w.planar);
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

template <bool BIG>
static int launch_peaks(ta_ctx* ctx, const op_work& w, int N, size_t lds) {
  ta_prof_scope scope(ctx, 3, (double)N * 18 * w.m.h * w.m.w * 4);     // algorithmic bytes: the 18 part maps, read once at network resolution
  if (!BIG) TA_SET_LDS_ATTR(ctx, peaks_kernel<BIG>, 150 * 1024);
  hipLaunchKernelGGL(peaks_kernel<BIG>, dim3(N * 18), dim3(PK_T), lds, ctx->stream, w);
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

template <bool BIG>
static int launch_limbs(ta_ctx* ctx, const op_work& w, int N, size_t lds) {
  ta_prof_scope scope(ctx, 3, (double)N * 38 * w.m.h * w.m.w * 4);
  if (!BIG) TA_SET_LDS_ATTR(ctx, limbs_kernel<BIG>, 150 * 1024);
  hipLaunchKernelGGL(limbs_kernel<BIG>, dim3(N * 19), dim3(256), lds, ctx->stream, w);
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

template <bool BIG>
static int launch_assemble(ta_ctx* ctx, const op_work& w, int N) {
  ta_prof_scope scope(ctx, 3, 0.0);
  hipLaunchKernelGGL(assemble_kernel<BIG>, dim3(N), dim3(64), 0, ctx->stream, w);
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

void ta_pose_free_big(ta_ctx* ctx) {
  for (auto& o : ctx->pose_dbg.over) (void)hipFree(o.mem);
  ctx->pose_dbg.over.clear();
}

// One image that outgrew the fast path's lists (or the <BIG> pass's default ones): the same kernels with every list in
// global memory, sized from what the passes report -- peaks from the first pass's true counts, candidate pairs from a
// counting pass of limbs_kernel, humans from the connections kept.  Nothing is capped (wrapper.py:235-262,335-366)
// beyond 65535 peaks per part (pair indices are 32 bits).
static int op_rerun_image(ta_ctx* ctx, const op_work& first, int img, const int* peak_true18, std::vector<int32_t>& kp,
                          std::vector<double>& sc, long long* peaks, long long* conns) {
  op_work w = first;
  w.N = 1;
  w.img_base = first.img_base + img;
  int maxp = 1;
  for (int p = 0; p < 18; ++p) maxp = peak_true18[p] > maxp ? peak_true18[p] : maxp;
  if (maxp > 65535) return ta_fail(ctx, TA_E_OVERFLOW, "openpose: image %d has %d peaks of one part (at most 65535 are supported)", img, maxp);
  w.maxp = maxp;
  w.maxc = 1;
  w.maxh = 1;                                      // out_kp / out_sc / humans of the layout below are placeholders (re-pointed later)
  const size_t cells = (size_t)w.m.h * w.m.w;
  const size_t bytes = layout_work(w, nullptr, 1, true, cells);
  char* mem = nullptr;
  TA_HIP(ctx, hipMalloc((void**)&mem, bytes));
  ctx->pose_dbg.over.push_back({img, maxp, mem, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr});
  layout_work(w, mem, 1, true, cells);
  TA_HIP(ctx, hipMemsetAsync(w.overflow, 0, 4, ctx->stream));
  TA_TRY(launch_planar(ctx, w, 1));
  TA_TRY(launch_peaks<true>(ctx, w, 1, 256));
  w.g_keys = nullptr;                              // counting pass
  TA_TRY(launch_limbs<true>(ctx, w, 1, 256));
  int pairs[19];
  TA_HIP(ctx, hipMemcpyAsync(pairs, w.pair_true, sizeof(pairs), hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int maxc = 1;
  for (int l = 0; l < 19; ++l) maxc = pairs[l] > maxc ? pairs[l] : maxc;
  w.maxc = maxc;
  w.keys_pitch = pow2_at_least(maxc);
  char* mem2 = nullptr;
  TA_HIP(ctx, hipMalloc((void**)&mem2, (size_t)19 * w.keys_pitch * 8));
  struct free_t {
    char*& p;
    ~free_t() { if (p) (void)hipFree(p); }
  } g2{mem2};
  w.g_keys = (unsigned long long*)mem2;
  TA_TRY(launch_limbs<true>(ctx, w, 1, 256));
  int cc[19];
  TA_HIP(ctx, hipMemcpyAsync(cc, w.conn_cnt, sizeof(cc), hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  long long nconn = 0;
  for (int l = 0; l < 19; ++l) nconn += cc[l] > 0 ? cc[l] : 0;
  w.maxh = (int)(nconn + 1);                       // every human is born from a connection
  const size_t hb = rup256((size_t)w.maxh * 20 * 8), kb = rup256((size_t)w.maxh * 54 * 4), sb = rup256((size_t)w.maxh * 8);
  char* mem3 = nullptr;
  TA_HIP(ctx, hipMalloc((void**)&mem3, hb + kb + sb));
  free_t g3{mem3};
  w.g_humans = (double*)mem3;
  w.out_kp = (int*)(mem3 + hb);
  w.out_sc = (double*)(mem3 + hb + kb);
  TA_TRY(launch_assemble<true>(ctx, w, 1));
  int head[2] = {0, 0};
  TA_HIP(ctx, hipMemcpyAsync(&head[0], w.out_cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipMemcpyAsync(&head[1], w.overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (head[1]) return ta_fail(ctx, TA_E_OVERFLOW, "openpose: internal error: image %d outgrew lists sized from its own counts", img);
  kp.resize((size_t)head[0] * 54);
  sc.resize((size_t)head[0]);
  if (head[0] > 0) {
    TA_HIP(ctx, hipMemcpyAsync(kp.data(), w.out_kp, kp.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    TA_HIP(ctx, hipMemcpyAsync(sc.data(), w.out_sc, sc.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  auto& o = ctx->pose_dbg.over.back();             // debug taps of this image live in `mem` until the next grouping on the context
  o.peak_cnt = w.peak_cnt;
  o.peak_yx = w.peak_yx;
  o.peak_sc = w.peak_sc;
  o.conn_cnt = w.conn_cnt;
  o.conn_ij = w.conn_ij;
  o.conn_sc = w.conn_sc;
  for (int p = 0; p < 18; ++p) *peaks += peak_true18[p];
  *conns += nconn;
  return TA_OK;
}

// network-resolution maps -> packed results on the host
static int op_postprocess_dev(ta_ctx* ctx, const op_maps& m, int N, double scale, int capacity, int32_t* counts,
                              int32_t* keypoints, double* scores, int32_t* required) {
  const int H8 = m.h * 8, W8 = m.w * 8;
  ta_pose_free_big(ctx);                                  // re-run areas of the previous grouping (kept for its debug taps)
  if (!ctx->pose_wphase) {                                // constants: uploaded once per context, no per-call copy / sync
    float wp[32];
    phase_weights(wp);
    TA_HIP(ctx, hipMalloc((void**)&ctx->pose_wphase, sizeof(wp)));
    TA_HIP(ctx, hipMemcpy(ctx->pose_wphase, wp, sizeof(wp), hipMemcpyHostToDevice));
  }
  const size_t cells = (size_t)m.h * m.w, map_bytes = cells * 4;
  const size_t lds_peaks = ((map_bytes + 15) & ~(size_t)15) + 128 + (size_t)OP_MAXP * 8;
  const size_t lds_limbs = (size_t)OP_MAXC * 8 + OP_MAXP / 8 + 32 + 128 + 2 * map_bytes;
  // maps beyond the LDS staging (~90 x 160 cells): the whole batch takes the <BIG> kernels (planar float32 copy of the
  // maps, lists in global memory), at the same list sizes; images that outgrow those are re-run one by one either way
  const bool big = lds_peaks > 150 * 1024 || lds_limbs > 150 * 1024;
  op_work w;
  memset(&w, 0, sizeof(w));
  w.m = m;
  w.wphase = ctx->pose_wphase;
  w.N = N;
  w.H8 = H8;
  w.W8 = W8;
  w.scale = scale;
  w.maxp = OP_MAXP;
  w.maxc = OP_MAXC;
  w.maxh = OP_MAXH;
  w.keys_pitch = OP_MAXC;
  const size_t cap = capacity > 0 ? (size_t)capacity : 0;
  const size_t work_bytes = layout_work(w, nullptr, N, big, cells);
  const size_t keys_bytes = big ? rup256((size_t)N * 19 * OP_MAXC * 8) : 0;
  const size_t o_gkp = work_bytes + keys_bytes, o_gsc = o_gkp + rup256(cap * 54 * 4);
  char* scr = nullptr;
  TA_TRY(ta_scratch(ctx, o_gsc + rup256(cap * 8), (void**)&scr));
  layout_work(w, scr, N, big, cells);
  if (big) w.g_keys = (unsigned long long*)(scr + work_bytes);
  TA_HIP(ctx, hipMemsetAsync(w.overflow, 0, (size_t)N * 4, ctx->stream));
  ctx->pose_dbg.n = N;
  ctx->pose_dbg.maxp = OP_MAXP;
  ctx->pose_dbg.peak_cnt = w.peak_cnt;
  ctx->pose_dbg.peak_yx = w.peak_yx;
  ctx->pose_dbg.peak_sc = w.peak_sc;
  ctx->pose_dbg.conn_cnt = w.conn_cnt;
  ctx->pose_dbg.conn_ij = w.conn_ij;
  ctx->pose_dbg.conn_sc = w.conn_sc;
  if (big) {
    TA_TRY(launch_planar(ctx, w, N));
    TA_TRY(launch_peaks<true>(ctx, w, N, 256));
    TA_TRY(launch_limbs<true>(ctx, w, N, 256));
    TA_TRY(launch_assemble<true>(ctx, w, N));
  } else {
    TA_TRY(launch_peaks<false>(ctx, w, N, lds_peaks));
    TA_TRY(launch_limbs<false>(ctx, w, N, lds_limbs));
    TA_TRY(launch_assemble<false>(ctx, w, N));
  }
  std::vector<int> ovf(N), ptrue((size_t)N * 18);
  std::vector<int> stat((size_t)N * 37);           // peaks per (image, part) and connections per (image, limb): statistics only
  TA_HIP(ctx, hipMemcpyAsync(counts, w.out_cnt, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipMemcpyAsync(ovf.data(), w.overflow, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipMemcpyAsync(ptrue.data(), w.peak_true, (size_t)N * 18 * 4, hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipMemcpyAsync(stat.data(), w.peak_cnt, (size_t)N * 18 * 4, hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipMemcpyAsync(stat.data() + (size_t)N * 18, w.conn_cnt, (size_t)N * 19 * 4, hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  long long n_peaks = 0, n_conns = 0;
  for (int i = 0; i < N; ++i) {
    if (ovf[i]) continue;                                   // counted by its re-run
    for (int p = 0; p < 18; ++p) n_peaks += stat[(size_t)i * 18 + p];
    for (int l = 0; l < 19; ++l) n_conns += stat[(size_t)N * 18 + (size_t)i * 19 + l] > 0 ? stat[(size_t)N * 18 + (size_t)i * 19 + l] : 0;   // -1 = limb missing
  }
  // images that outgrew a list: recomputed alone, lists sized from their own counts (their out_cnt above is 0)
  std::vector<int> over_img;
  std::vector<std::vector<int32_t>> over_kp;
  std::vector<std::vector<double>> over_sc;
  for (int i = 0; i < N; ++i) {
    if (!ovf[i]) continue;
    over_img.push_back(i);
    over_kp.emplace_back();
    over_sc.emplace_back();
    TA_TRY(op_rerun_image(ctx, w, i, &ptrue[(size_t)i * 18], over_kp.back(), over_sc.back(), &n_peaks, &n_conns));
    counts[i] = (int32_t)over_sc.back().size();
  }
  ctx->pose_peaks = n_peaks;
  ctx->pose_connections = n_conns;
  long long total = 0, fast_total = 0;
  for (int i = 0; i < N; ++i) {
    total += counts[i];
    if (!ovf[i]) fast_total += counts[i];
  }
  if (required) *required = (int32_t)total;
  if (total > capacity) return ta_fail(ctx, TA_E_CAPACITY, "openpose: %lld humans, capacity %d", total, capacity);
  if (total == 0) return TA_OK;
  if (over_img.empty()) {
    hipLaunchKernelGGL(op_gather_kernel, dim3(N), dim3(256), 0, ctx->stream, w, (int*)(scr + o_gkp), (double*)(scr + o_gsc));
    TA_HIP(ctx, hipGetLastError());
    TA_HIP(ctx, hipMemcpyAsync(keypoints, scr + o_gkp, (size_t)total * 54 * 4, hipMemcpyDeviceToHost, ctx->stream));
    TA_HIP(ctx, hipMemcpyAsync(scores, scr + o_gsc, (size_t)total * 8, hipMemcpyDeviceToHost, ctx->stream));
    TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TA_OK;
  }
  // splice: the fast path's rows (packed in image order, re-run images contribute none there) and the re-run images' rows
  std::vector<int32_t> fkp((size_t)fast_total * 54);
  std::vector<double> fsc((size_t)fast_total);
  if (fast_total > 0) {
    hipLaunchKernelGGL(op_gather_kernel, dim3(N), dim3(256), 0, ctx->stream, w, (int*)(scr + o_gkp), (double*)(scr + o_gsc));
    TA_HIP(ctx, hipGetLastError());
    TA_HIP(ctx, hipMemcpyAsync(fkp.data(), scr + o_gkp, fkp.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    TA_HIP(ctx, hipMemcpyAsync(fsc.data(), scr + o_gsc, fsc.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  size_t o = 0, fo = 0, k = 0;
  for (int i = 0; i < N; ++i) {
    const size_t c = (size_t)counts[i];
    if (ovf[i]) {
      if (c) {
        memcpy(keypoints + o * 54, over_kp[k].data(), c * 54 * 4);
        memcpy(scores + o, over_sc[k].data(), c * 8);
      }
      ++k;
    } else if (c) {
      memcpy(keypoints + o * 54, fkp.data() + fo * 54, c * 54 * 4);
      memcpy(scores + o, fsc.data() + fo, c * 8);
      fo += c;
    }
    o += c;
  }
  return TA_OK;
}

// host NCHW maps -> device NHWC(c channels) tensor
static int upload_maps(ta_ctx* ctx, const float* const* srcs, const int* chs, int nsrc, int n, int h, int w, float** dev,
                       int* ctot) {
  int c = 0;
  for (int k = 0; k < nsrc; ++k) c += chs[k];
  std::vector<float> host((size_t)n * h * w * c);
  int co = 0;
  for (int k = 0; k < nsrc; ++k) {
    for (int i = 0; i < n; ++i)
      for (int ch = 0; ch < chs[k]; ++ch)
        for (int y = 0; y < h; ++y)
          for (int x = 0; x < w; ++x)
            host[(((size_t)i * h + y) * w + x) * c + co + ch] = srcs[k][(((size_t)i * chs[k] + ch) * h + y) * w + x];
    co += chs[k];
  }
  TA_HIP(ctx, hipMalloc((void**)dev, host.size() * sizeof(float)));
  hipError_t e = hipMemcpy(*dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(*dev);
    return ta_fail(ctx, TA_E_DEVICE, "upload failed: %s", hipGetErrorString(e));
  }
  *ctot = c;
  return TA_OK;
}

extern "C" {

int ta_openpose_run(ta_model* m, const ta_frames* frames, double scale, int capacity, int32_t* counts,
                    int32_t* keypoints, double* scores, int32_t* required) {
  ta_enter(m ? m->ctx : nullptr);
  if (!m || !frames || !counts) return TA_E_INVALID;
  ta_ctx* ctx = m->ctx;
  if (m->kind != TA_MODEL_OPENPOSE) return ta_fail(ctx, TA_E_INVALID, "openpose_run: wrong model kind");
  if (capacity > 0 && (!keypoints || !scores)) return ta_fail(ctx, TA_E_INVALID, "openpose_run: null outputs");
  if (!(scale > 0.0)) return ta_fail(ctx, TA_E_INVALID, "openpose_run: scale must be positive");
  if (required) *required = 0;
  if (frames->n == 0) return TA_OK;
  TA_TRY(ta_model_forward_frames(m, frames));
  const ta_tensor& X = m->tensors[m->hdr.outputs[0]];
  op_maps mp;
  mp.base = X.dev;
  mp.img = (int)((size_t)X.hp() * X.wp() * X.c);
  mp.row = X.wp() * X.c;
  mp.pix = X.c;
  mp.off0 = (int)X.off(0, 0, 0);
  mp.paf_ch = 128;
  mp.hm_ch = 168;
  mp.fmt = X.fmt;
  mp.h = X.h;
  mp.w = X.w;
  mp.unscale = X.unscale_dev;
  TA_TRY(ta_range_enqueue(ctx));                 // behind the network, ahead of the grouping's syncs
  const int rc = op_postprocess_dev(ctx, mp, frames->n, scale, capacity, counts, keypoints, scores, required);
  return ta_range_finish(ctx, rc);               // f16x3: TA_E_RANGE when an activation left the half-float range
}

int ta_openpose_group(ta_ctx* ctx, const float* pafs, const float* heatmaps, int n, int h, int w, double scale,
                      int capacity, int32_t* counts, int32_t* keypoints, double* scores, int32_t* required) {
  ta_enter(ctx);
  if (!ctx || n < 0 || h <= 0 || w <= 0 || !counts) return TA_E_INVALID;
  if (required) *required = 0;
  if (n == 0) return TA_OK;
  if (!pafs || !heatmaps) return ta_fail(ctx, TA_E_INVALID, "openpose_group: null maps");
  if (!(scale > 0.0)) return ta_fail(ctx, TA_E_INVALID, "openpose_group: scale must be positive");
  const float* srcs[2] = {pafs, heatmaps};
  const int chs[2] = {38, 19};
  float* dev = nullptr;
  int c = 0;
  TA_TRY(upload_maps(ctx, srcs, chs, 2, n, h, w, &dev, &c));
  op_maps mp;
  mp.base = dev;
  mp.img = h * w * c;
  mp.row = w * c;
  mp.pix = c;
  mp.off0 = 0;
  mp.paf_ch = 0;
  mp.hm_ch = 38;
  mp.fmt = TA_FMT_F32;
  mp.h = h;
  mp.w = w;
  mp.unscale = nullptr;
  const int rc = op_postprocess_dev(ctx, mp, n, scale, capacity, counts, keypoints, scores, required);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(dev);
  return rc;
}

int ta_openpose_last_stats(const ta_ctx* ctx, int64_t* peaks, int64_t* connections) {
  if (!ctx) return TA_E_INVALID;
  if (peaks) *peaks = ctx->pose_peaks;
  if (connections) *connections = ctx->pose_connections;
  return TA_OK;
}

int ta_openpose_debug_read(ta_ctx* ctx, int n, int cap_peaks, int32_t* peak_counts, int32_t* peaks_yx, float* peak_scores,
                           int cap_conn, int32_t* conn_counts, int32_t* conn_ij, float* conn_scores) {
  ta_enter(ctx);
  if (!ctx || n <= 0 || cap_peaks < 0 || cap_conn < 0) return TA_E_INVALID;
  const auto& d = ctx->pose_dbg;
  if (d.n != n || !d.peak_cnt) return ta_fail(ctx, TA_E_INVALID, "openpose_debug_read: the last grouping on this context had %d images, not %d", d.n, n);
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int img = 0; img < n; ++img) {
    // an image that was re-run with larger lists reports from its own work area (local image 0 there)
    const int *pcnt = d.peak_cnt + (size_t)img * 18, *pyx = d.peak_yx + (size_t)img * 18 * d.maxp * 2;
    const int *ccnt = d.conn_cnt + (size_t)img * 19, *cij = d.conn_ij + (size_t)img * 19 * d.maxp * 2;
    const float *psc = d.peak_sc + (size_t)img * 18 * d.maxp, *csc = d.conn_sc + (size_t)img * 19 * d.maxp;
    int P = d.maxp;
    for (const auto& o : d.over)
      if (o.img == img && o.peak_cnt) {
        pcnt = o.peak_cnt, pyx = o.peak_yx, psc = o.peak_sc, ccnt = o.conn_cnt, cij = o.conn_ij, csc = o.conn_sc;
        P = o.maxp;
      }
    int pc[18], cc[19];
    TA_HIP(ctx, hipMemcpy(pc, pcnt, sizeof(pc), hipMemcpyDeviceToHost));
    TA_HIP(ctx, hipMemcpy(cc, ccnt, sizeof(cc), hipMemcpyDeviceToHost));
    if (peak_counts) memcpy(peak_counts + (size_t)img * 18, pc, sizeof(pc));
    if (conn_counts) memcpy(conn_counts + (size_t)img * 19, cc, sizeof(cc));
    for (int p = 0; p < 18; ++p) {
      const int k = pc[p] < cap_peaks ? pc[p] : cap_peaks;
      if (k <= 0) continue;
      const size_t dst = (size_t)img * 18 + p;
      if (peaks_yx) TA_HIP(ctx, hipMemcpy(peaks_yx + dst * cap_peaks * 2, pyx + (size_t)p * P * 2, (size_t)k * 8, hipMemcpyDeviceToHost));
      if (peak_scores) TA_HIP(ctx, hipMemcpy(peak_scores + dst * cap_peaks, psc + (size_t)p * P, (size_t)k * 4, hipMemcpyDeviceToHost));
    }
    for (int l = 0; l < 19; ++l) {
      const int k = cc[l] < cap_conn ? cc[l] : cap_conn;
      if (k <= 0) continue;
      const size_t dst = (size_t)img * 19 + l;
      if (conn_ij) TA_HIP(ctx, hipMemcpy(conn_ij + dst * cap_conn * 2, cij + (size_t)l * P * 2, (size_t)k * 8, hipMemcpyDeviceToHost));
      if (conn_scores) TA_HIP(ctx, hipMemcpy(conn_scores + dst * cap_conn, csc + (size_t)l * P, (size_t)k * 4, hipMemcpyDeviceToHost));
    }
  }
  return TA_OK;
}

int ta_bicubic_x8(ta_ctx* ctx, const float* maps, int n, int c, int h, int w, float* out) {
  ta_enter(ctx);
  if (!ctx || n < 0 || c <= 0 || h <= 0 || w <= 0) return TA_E_INVALID;
  if (n == 0) return TA_OK;
  if (!maps || !out) return ta_fail(ctx, TA_E_INVALID, "bicubic: null pointer");
  // run the 57-channel kernel over channel groups: pad channels to 57 by replicating channel 0
  for (int c0 = 0; c0 < c; c0 += 57) {
    const int cc = c - c0 < 57 ? c - c0 : 57;
    std::vector<float> a((size_t)n * 38 * h * w), b((size_t)n * 19 * h * w);
    for (int i = 0; i < n; ++i)
      for (int ch = 0; ch < 57; ++ch) {
        const int srcc = c0 + (ch < cc ? ch : 0);
        const float* s = maps + ((size_t)i * c + srcc) * h * w;
        float* d = ch < 38 ? &a[((size_t)i * 38 + ch) * h * w] : &b[((size_t)i * 19 + (ch - 38)) * h * w];
        memcpy(d, s, (size_t)h * w * 4);
      }
    const float* srcs[2] = {a.data(), b.data()};
    const int chs[2] = {38, 19};
    float* dev = nullptr;
    int ct = 0;
    TA_TRY(upload_maps(ctx, srcs, chs, 2, n, h, w, &dev, &ct));
    op_maps mp;
    mp.base = dev;
    mp.img = h * w * ct;
    mp.row = w * ct;
    mp.pix = ct;
    mp.off0 = 0;
    mp.paf_ch = 0;
    mp.hm_ch = 38;
    mp.fmt = TA_FMT_F32;
    mp.h = h;
    mp.w = w;
    mp.unscale = nullptr;
    std::vector<float> up((size_t)n * 57 * 64 * h * w);
    const int rc = op_upsample_dev(ctx, mp, n, up.data());
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(dev);
    if (rc != TA_OK) return rc;
    for (int i = 0; i < n; ++i)
      for (int ch = 0; ch < cc; ++ch)
        memcpy(out + ((size_t)i * c + c0 + ch) * 64 * h * w, &up[((size_t)i * 57 + ch) * 64 * h * w], (size_t)64 * h * w * 4);
  }
  return TA_OK;
}

}  // extern "C"

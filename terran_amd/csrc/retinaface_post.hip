// RetinaFace post-processing on the device (retinaface/wrapper.py:153-236):
// fg-probability, threshold (>=), ordered compaction, sort by descending score (ties: ascending
// anchor index), anchor + box/landmark decode (retinaface/anchors.py:7-51, wrapper.py:25-89) and
// greedy NMS (torchvision.ops.nms semantics: IoU = inter/(a+b-inter), suppress iff IoU > thr).
//
// One workgroup (1024 threads = 16 waves) per image does the whole selection: wave-ballot
// prefix sums for the ordered compaction, a bitonic sort of 64-bit (score, index) keys staged in
// LDS, and the sequential-greedy NMS with the suppression bitmap in LDS.  All float math is
// single-rounded (this file is compiled with -ffp-contract=off) so decisions are bit-exact with
// oracle/retinaface_post.py when fed the same numbers.
#include <string.h>
#include <vector>

#include "act_format.h"
#include "ta_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define RF_THREADS 1024
#define RF_LDS_KEYS 8192
#define RF_FAST_C 1024                 // candidates up to which the NMS runs on a suppression matrix in LDS (see step 4)
#define RF_FAST_LDS (24 * 1024 + (RF_FAST_C / 64) * (RF_FAST_C / 64 + 1) / 2 * 64 * 8)   // keys 8 KB | boxes 16 KB | matrix 68 KB
#define RF_LDS_MAX (159 * 1024)     // dynamic LDS the kernel may be launched with (the attribute and the limit check agree)

struct rf_level {
  const float* head;   // NHWC, 32 channels: cls[0:4] | bbox[4:12] | landmark[12:32]
  int img, row, pix, off0;
  int fh, fw, stride;
  int t0;              // first anchor index of this level in the concatenated order
  float ref[2][4];     // anchor reference boxes
};

struct rf_params {
  rf_level lv[3];
  int T;               // anchors per image
  int cls_is_prob;     // heads hold softmax probabilities (reference layout) instead of logits
  float score_thr, nms_thr;
  float margin;        // fg - bg below this: the score is clearly under score_thr (-inf: no shortcut)
  unsigned long long* keys;   // [N][P_max] global sort buffer (used when candidates > RF_LDS_KEYS)
  int p_max;
  float* boxes;        // [N][T][4]  sorted candidates, decoded
  float* lmks;         // [N][T][10]
  float* scores;       // [N][T]
  int* keep;           // [N][T] sorted positions kept by NMS
  int* counts;         // [N] kept
  int* ncand;          // [N] candidates
  long long* dbg;
  int fast_max;        // candidates up to which the NMS takes the suppression-matrix path (RF_FAST_C; tools: TA_RF_FAST_MAX)
};

__device__ __forceinline__ const float* rf_cell(const rf_level& L, int img, int t, int& a) {
  const int cell = t >> 1;
  a = t & 1;
  const int y = cell / L.fw, x = cell - y * L.fw;
  return L.head + (size_t)img * L.img + (size_t)y * L.row + (size_t)x * L.pix + L.off0;
}

__device__ __forceinline__ float rf_score(const rf_params& p, int img, int t) {
  const int l = t >= p.lv[2].t0 ? 2 : (t >= p.lv[1].t0 ? 1 : 0);
  int a;
  const float* c = rf_cell(p.lv[l], img, t - p.lv[l].t0, a);
  if (p.cls_is_prob) return c[2 + a];
  const float bg = c[a], fg = c[2 + a];
  const float m = fmaxf(bg, fg);
  const float eb = expf(bg - m), ef = expf(fg - m);
  return ef / (eb + ef);
}

// 64-bit value of lane `l` (wave-uniform l).  The builtin returns a SIGNED int: the halves go through unsigned first
__device__ __forceinline__ unsigned long long rf_readlane64(unsigned long long v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

// first word of row i of the packed suppression matrix: rows of block b = i / 64 hold W - b words
__device__ __forceinline__ int rf_row_off(int i, int W) {
  const int b = i >> 6;
  return 64 * (b * W - b * (b - 1) / 2) + (i - 64 * b) * (W - b);
}

__global__ __launch_bounds__(RF_THREADS) void rf_select_kernel(const rf_params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* lkeys = (unsigned long long*)smem;                    // RF_LDS_KEYS
  unsigned* flags = (unsigned*)(smem + (size_t)RF_LDS_KEYS * 8);            // ceil(T/32) words
  int* wave_tot = (int*)(flags + ((p.T + 31) / 32 + 3) / 4 * 4);            // 16 ints (all LDS is dynamic: G17)
  int& s_base = wave_tot[16];

  const int img = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned long long* gkeys = p.keys + (size_t)img * p.p_max;

  if (p.dbg && tid == 0) p.dbg[img * 8 + 0] = wall_clock64();
  // ---- 1. threshold + ordered compaction into gkeys -----------------------------------------
  // thread = a contiguous run of anchors: count the passing ones, one exclusive prefix sum over the workgroup (shuffles
  // inside a wave, the 16 wave totals through LDS), then the run is scored again and its keys written in place -- the order
  // is the anchor order, as with a ballot per 1024 anchors, for two barriers instead of four per 1024 anchors.
  // thread = a contiguous run of CELLS (2 anchors each: one 16-byte load brings bg0 bg1 fg0 fg1), four cells at a time so
  // that their loads are in flight together.  (level, y, x) of the first cell once, then increments -- no division per
  // anchor.  An anchor whose logit margin fg - bg lies clearly below logit(threshold) (p.margin, 1e-2 of slack: the float
  // softmax is good to ~1e-6) fails the exact test too and skips the two exps and the division.
  const int ncell = p.T >> 1;
  const int perc = (ncell + RF_THREADS - 1) / RF_THREADS;
  const int c0 = tid * perc, c1 = (c0 + perc < ncell) ? c0 + perc : ncell;
  const int per = 2 * perc, a0 = 2 * c0, a1 = 2 * c1;
  int cnt = 0;
  unsigned long long bits = 0;                       // which anchors of the run passed (runs of up to 64; longer: rescored)
  if (c0 < c1) {
    int l = a0 >= p.lv[2].t0 ? 2 : (a0 >= p.lv[1].t0 ? 1 : 0);
    const int cell = (a0 - p.lv[l].t0) >> 1;
    int y = cell / p.lv[l].fw, x = cell - y * p.lv[l].fw;
    const float* c = p.lv[l].head + (size_t)img * p.lv[l].img + (size_t)y * p.lv[l].row + (size_t)x * p.lv[l].pix + p.lv[l].off0;
    for (int q = c0; q < c1; q += 4) {
      f32x4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = *(const f32x4*)c;                     // past the run's end: the last cell again (ignored below)
        if (q + k + 1 < c1) {
          c += p.lv[l].pix;
          if (++x == p.lv[l].fw) {
            x = 0;
            ++y;
            c = p.lv[l].head + (size_t)img * p.lv[l].img + (size_t)y * p.lv[l].row + p.lv[l].off0;
          }
          if (l < 2 && 2 * (q + k + 1) == p.lv[l + 1].t0) {
            ++l;
            x = y = 0;
            c = p.lv[l].head + (size_t)img * p.lv[l].img + p.lv[l].off0;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          if (q + k >= c1) continue;
          bool pass;
          if (p.cls_is_prob) {
            pass = v[k][2 + a] >= p.score_thr;
          } else {
            const float bg = v[k][a], fg = v[k][2 + a];
            pass = false;
            if (!(fg - bg < p.margin)) {             // also taken by NaNs: the exact test decides
              const float m = fmaxf(bg, fg);
              const float eb = expf(bg - m), ef = expf(fg - m);
              pass = ef / (eb + ef) >= p.score_thr;
            }
          }
          if (pass) {
            ++cnt;
            const int o = 2 * (q + k - c0) + a;
            if (o < 64) bits |= 1ull << o;
          }
        }
    }
  }
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  int off = incl - cnt;
  for (int w = 0; w < wv; ++w) off += wave_tot[w];
  if (tid == RF_THREADS - 1) s_base = off + cnt;
  if (per <= 64) {
    while (bits) {
      const int t = a0 + __builtin_ctzll(bits);
      bits &= bits - 1;
      // ascending key order == descending score, then ascending anchor index
      const unsigned hi = 0xFFFFFFFFu - __float_as_uint(rf_score(p, img, t));
      gkeys[off++] = ((unsigned long long)hi << 32) | (unsigned)t;
    }
  } else {
    for (int t = a0; t < a1; ++t) {
      const float sc = rf_score(p, img, t);
      if (sc >= p.score_thr) gkeys[off++] = ((unsigned long long)(0xFFFFFFFFu - __float_as_uint(sc)) << 32) | (unsigned)t;
    }
  }
  __syncthreads();
  const int C = s_base;
  if (tid == 0) p.ncand[img] = C;
  if (C == 0) {
    if (tid == 0) p.counts[img] = 0;
    return;
  }

  if (p.dbg && tid == 0) p.dbg[img * 8 + 1] = wall_clock64();
  // ---- 2. bitonic sort of the keys (LDS when they fit) ---------------------------------------
  int P = 1;
  while (P < C) P <<= 1;
  unsigned long long* keys = (P <= RF_LDS_KEYS) ? lkeys : gkeys;
  for (int i = tid; i < P; i += RF_THREADS) {
    const unsigned long long k = i < C ? gkeys[i] : ~0ull;
    if (keys != gkeys || i >= C) keys[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += RF_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], b = keys[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            keys[i] = b;
            keys[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }

  if (p.dbg && tid == 0) p.dbg[img * 8 + 2] = wall_clock64();
  // ---- 3. decode the sorted candidates ---------------------------------------------------------
  // C <= RF_FAST_C (the keys then occupy the first 8 KB of the key region): the decoded boxes also stay in LDS at +8 KB and
  // the suppression matrix of step 4 goes to +24 KB -- row i holds words i / 64 .. W - 1 only (j > i), rows packed
  const bool fast = C <= p.fast_max && keys == lkeys;
  f32x4* sbox = (f32x4*)(smem + 8 * 1024);
  unsigned long long* smask = (unsigned long long*)(smem + 24 * 1024);
  float* boxes = p.boxes + (size_t)img * p.T * 4;
  float* lmks = p.lmks + (size_t)img * p.T * 10;
  float* scores = p.scores + (size_t)img * p.T;
  for (int i = tid; i < C; i += RF_THREADS) {
    const unsigned long long key = keys[i];
    const int t = (int)(unsigned)(key & 0xFFFFFFFFull);
    scores[i] = __uint_as_float(0xFFFFFFFFu - (unsigned)(key >> 32));
    const int l = t >= p.lv[2].t0 ? 2 : (t >= p.lv[1].t0 ? 1 : 0);
    const rf_level& L = p.lv[l];
    int a;
    const int tl = t - L.t0;
    const float* c = rf_cell(L, img, tl, a);
    const int cell = tl >> 1;
    const int y = cell / L.fw, x = cell - y * L.fw;
    const float sx = (float)x * (float)L.stride, sy = (float)y * (float)L.stride;
    const float ax1 = L.ref[a][0] + sx, ay1 = L.ref[a][1] + sy, ax2 = L.ref[a][2] + sx, ay2 = L.ref[a][3] + sy;
    const float w = ax2 - ax1 + 1.0f, h = ay2 - ay1 + 1.0f;
    const float cx = ax1 + 0.5f * (w - 1.0f), cy = ay1 + 0.5f * (h - 1.0f);
    const float* bd = c + 4 + a * 4;
    const float pcx = bd[0] * w + cx, pcy = bd[1] * h + cy;
    const float pw = expf(bd[2]) * w, ph = expf(bd[3]) * h;
    const f32x4 bx = {pcx - 0.5f * (pw - 1.0f), pcy - 0.5f * (ph - 1.0f), pcx + 0.5f * (pw - 1.0f), pcy + 0.5f * (ph - 1.0f)};
    *(f32x4*)(boxes + i * 4) = bx;
    if (fast) sbox[i] = bx;
    const float* ld = c + 12 + a * 10;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      lmks[i * 10 + 2 * k] = ld[2 * k] * w + cx;
      lmks[i * 10 + 2 * k + 1] = ld[2 * k + 1] * h + cy;
    }
  }
  for (int i = tid; i < (C + 31) / 32; i += RF_THREADS) flags[i] = 0;
  __syncthreads();

  if (p.dbg && tid == 0) p.dbg[img * 8 + 3] = wall_clock64();
  // ---- 4. greedy NMS ------------------------------------------------------------------------
  int* keep = p.keep + (size_t)img * p.T;
  if (fast) {
    // The greedy loop below costs a barrier (and, from global memory, a box) per kept detection: ~0.7 us each, 265 us per
    // image at 1080p (360 candidates) -- a quarter of the detector.  Up to RF_FAST_C candidates the same decisions come
    // from a suppression MATRIX built by all 16 waves at once (wave = (i, 64 candidates j): one IoU per lane, the ballot is
    // the row's word) and one wave's walk over it: lane w keeps word w of the removed set; candidate i is kept iff its
    // bit is clear, and then ORs row i in.  Same IoU expression, same `>`: the kept set equals the loop's bit for bit.
    const int W = (C + 63) >> 6;
    for (int i = wv; i < C; i += RF_THREADS / 64) {          // wave = row i, lane = candidate j of word w
      const f32x4 a = sbox[i];
      const float area = (a[2] - a[0]) * (a[3] - a[1]);
      const int ro = rf_row_off(i, W) - (i >> 6);
      for (int w = i >> 6; w < W; ++w) {
        const int j = w * 64 + lane;
        bool sup = false;
        if (j > i && j < C) {
          const f32x4 b = sbox[j];
          const float xx1 = fmaxf(a[0], b[0]), yy1 = fmaxf(a[1], b[1]);
          const float xx2 = fminf(a[2], b[2]), yy2 = fminf(a[3], b[3]);
          const float iw = fmaxf(0.0f, xx2 - xx1), ih = fmaxf(0.0f, yy2 - yy1);
          const float inter = iw * ih;
          const float ovr = inter / (area + (b[2] - b[0]) * (b[3] - b[1]) - inter);
          sup = ovr > p.nms_thr;
        }
        const unsigned long long m = __ballot(sup);
        if (lane == 0) smask[ro + w] = m;
      }
    }
    __syncthreads();
    if (p.dbg && tid == 0) p.dbg[img * 8 + 4] = wall_clock64();
    if (wv == 0) {
      // one wave walks the matrix a block of 64 candidates at a time.  Lane l holds word l of the removed set.  Inside
      // block w only the block's own column of the rows matters: lane i holds that word of row 64 w + i, the 64 decisions
      // run on scalars (readlane), and the kept rows' later words are then ORed in with one wave-wide reduction per word.
      unsigned long long removed = 0;
      int kept = 0;
      for (int w = 0; w < W; ++w) {
        const int i0 = w * 64, n = (C - i0 < 64) ? C - i0 : 64;
        const int row = i0 + lane;
        const int ro = (lane < n) ? rf_row_off(row, W) : 0;   // row's word w is its first
        const unsigned long long own = (lane < n) ? smask[ro] : 0ull;
        unsigned long long cur = rf_readlane64(removed, w);
        unsigned long long keptbits = 0;
        for (int i = 0; i < n; ++i) {
          if ((cur >> i) & 1ull) continue;
          keptbits |= 1ull << i;
          cur |= rf_readlane64(own, i);
        }
        const bool mine = (keptbits >> lane) & 1ull;
        if (mine) keep[kept + __popcll(keptbits & ((1ull << lane) - 1ull))] = row;
        kept += __popcll(keptbits);
        for (int w2 = w + 1; w2 < W; ++w2) {                  // OR of the kept rows' word w2 over the wave -> lane w2
          unsigned long long v = mine ? smask[ro + w2 - w] : 0ull;
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) v |= __shfl_xor(v, d, 64);
          if (lane == w2) removed |= v;
        }
      }
      if (lane == 0) p.counts[img] = kept;
      if (p.dbg && lane == 0) { p.dbg[img * 8 + 5] = wall_clock64(); p.dbg[img * 8 + 6] = C; p.dbg[img * 8 + 7] = kept; }
    }
    return;
  }
  int kept = 0;
  for (int i = 0; i < C; ++i) {
    if ((flags[i >> 5] >> (i & 31)) & 1u) continue;   // uniform: stable since the last barrier
    if (tid == 0) keep[kept] = i;
    ++kept;
    const float x1 = boxes[i * 4], y1 = boxes[i * 4 + 1], x2 = boxes[i * 4 + 2], y2 = boxes[i * 4 + 3];
    const float area = (x2 - x1) * (y2 - y1);
    for (int j = i + 1 + tid; j < C; j += RF_THREADS) {
      if ((flags[j >> 5] >> (j & 31)) & 1u) continue;
      const float bx1 = boxes[j * 4], by1 = boxes[j * 4 + 1], bx2 = boxes[j * 4 + 2], by2 = boxes[j * 4 + 3];
      const float xx1 = fmaxf(x1, bx1), yy1 = fmaxf(y1, by1);
      const float xx2 = fminf(x2, bx2), yy2 = fminf(y2, by2);
      const float iw = fmaxf(0.0f, xx2 - xx1), ih = fmaxf(0.0f, yy2 - yy1);
      const float inter = iw * ih;
      const float ovr = inter / (area + (bx2 - bx1) * (by2 - by1) - inter);
      if (ovr > p.nms_thr) atomicOr(&flags[j >> 5], 1u << (j & 31));
    }
    __syncthreads();
  }
  if (tid == 0) p.counts[img] = kept;
  if (p.dbg && tid == 0) { p.dbg[img * 8 + 4] = p.dbg[img * 8 + 3]; p.dbg[img * 8 + 5] = wall_clock64(); p.dbg[img * 8 + 6] = C; p.dbg[img * 8 + 7] = kept; }
}

// gather kept detections of all images into packed output arrays (image order preserved)
__global__ __launch_bounds__(256) void rf_gather_kernel(const rf_params p, int N, float* o_boxes, float* o_lmks,
                                                         float* o_scores) {
  const int img = blockIdx.x;
  int base = 0;
  for (int i = 0; i < img; ++i) base += p.counts[i];
  const int K = p.counts[img];
  const int* keep = p.keep + (size_t)img * p.T;
  const float* boxes = p.boxes + (size_t)img * p.T * 4;
  const float* lmks = p.lmks + (size_t)img * p.T * 10;
  const float* scores = p.scores + (size_t)img * p.T;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const int i = keep[k];
    const int o = base + k;
#pragma unroll
    for (int e = 0; e < 4; ++e) o_boxes[o * 4 + e] = boxes[i * 4 + e];
#pragma unroll
    for (int e = 0; e < 10; ++e) o_lmks[o * 10 + e] = lmks[i * 10 + e];
    o_scores[o] = scores[i];
  }
}

static void anchor_refs(int stride, float ref[2][4]) {
  // base 16x16 box, ratio 1, scales per stride (retinaface/wrapper.py:101-117, anchors.py:54-134)
  const int scales[3][2] = {{32, 16}, {8, 4}, {2, 1}};
  const int li = stride == 32 ? 0 : (stride == 16 ? 1 : 2);
  const float ctr = 0.5f * (16 - 1);
  for (int a = 0; a < 2; ++a) {
    const float half = 0.5f * (16.0f * scales[li][a] - 1.0f);
    ref[a][0] = ctr - half;
    ref[a][1] = ctr - half;
    ref[a][2] = ctr + half;
    ref[a][3] = ctr + half;
  }
}

// Shared tail: heads are three NHWC/32ch device tensors.
static int rf_postprocess_dev(ta_ctx* ctx, const ta_tensor heads[3], int N, int H, int W, int cls_is_prob,
                              float score_thr, float nms_thr, int capacity, int32_t* counts, float* boxes,
                              float* landmarks, float* scores, int32_t* required) {
  if (N <= 0) {
    if (required) *required = 0;
    return TA_OK;
  }
  rf_params p;
  memset(&p, 0, sizeof(p));
  const int strides[3] = {32, 16, 8};
  int T = 0;
  for (int l = 0; l < 3; ++l) {
    const ta_tensor& t = heads[l];
    const int fh = (H + strides[l] - 1) / strides[l], fw = (W + strides[l] - 1) / strides[l];
    if (t.h != fh || t.w != fw || t.c != 32) return ta_fail(ctx, TA_E_INVALID, "retinaface: head %d has shape %dx%dx%d, expected %dx%dx32", l, t.h, t.w, t.c, fh, fw);
    rf_level& L = p.lv[l];
    L.head = t.dev;
    L.img = (int)((size_t)t.hp() * t.wp() * t.c);
    L.row = t.wp() * t.c;
    L.pix = t.c;
    L.off0 = (int)t.off(0, 0, 0);
    L.fh = fh;
    L.fw = fw;
    L.stride = strides[l];
    L.t0 = T;
    anchor_refs(strides[l], L.ref);
    T += fh * fw * 2;
  }
  p.T = T;
  p.cls_is_prob = cls_is_prob;
  p.score_thr = score_thr;
  p.margin = -INFINITY;
  if (score_thr > 1e-4f && score_thr < 1.0f - 1e-4f) {
    const double lg = log((double)score_thr / (1.0 - (double)score_thr));
    p.margin = (float)(lg - 0.01 - 0.001 * fabs(lg));
  }
  p.nms_thr = nms_thr;
  int pmax = 1;
  while (pmax < T) pmax <<= 1;
  p.p_max = pmax;
  // scratch layout
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t o_keys = carve((size_t)N * pmax * 8), o_boxes = carve((size_t)N * T * 16), o_lmks = carve((size_t)N * T * 40);
  const size_t o_scores = carve((size_t)N * T * 4), o_keep = carve((size_t)N * T * 4), o_counts = carve((size_t)N * 4);
  const size_t o_ncand = carve((size_t)N * 4);
  const size_t cap = capacity > 0 ? (size_t)capacity : 0;
  const size_t o_ob = carve(cap * 16), o_ol = carve(cap * 40), o_os = carve(cap * 4);
  char* scr = nullptr;
  TA_TRY(ta_scratch(ctx, off, (void**)&scr));
  p.keys = (unsigned long long*)(scr + o_keys);
  p.boxes = (float*)(scr + o_boxes);
  p.lmks = (float*)(scr + o_lmks);
  p.scores = (float*)(scr + o_scores);
  p.keep = (int*)(scr + o_keep);
  p.counts = (int*)(scr + o_counts);
  p.ncand = (int*)(scr + o_ncand);
  size_t lds = (size_t)RF_LDS_KEYS * 8 + (size_t)(((T + 31) / 32 + 3) / 4 * 4) * 4 + 32 * 4;
  if (lds < RF_FAST_LDS) lds = RF_FAST_LDS;              // the suppression-matrix path reuses the key / bitmap space and then some
  if (lds > RF_LDS_MAX) return ta_fail(ctx, TA_E_OVERFLOW, "retinaface: %d anchors per image exceed the NMS bitmap limit", T);
  static long long* dbg_dev = nullptr;
  if (getenv("TA_RF_DEBUG") && !dbg_dev) (void)hipMalloc((void**)&dbg_dev, 8 * 8 * 4096);
  p.dbg = dbg_dev;
  static const int fast_max = getenv("TA_RF_FAST_MAX") ? atoi(getenv("TA_RF_FAST_MAX")) : RF_FAST_C;
  p.fast_max = fast_max < RF_FAST_C ? fast_max : RF_FAST_C;
  {
    ta_prof_scope scope(ctx, 3, (double)N * T * 32 * 4);
    TA_SET_LDS_ATTR(ctx, rf_select_kernel, RF_LDS_MAX);
    hipLaunchKernelGGL(rf_select_kernel, dim3(N), dim3(RF_THREADS), lds, ctx->stream, p);
    TA_HIP(ctx, hipGetLastError());
  }
  TA_HIP(ctx, hipMemcpyAsync(counts, p.counts, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (dbg_dev) {
    std::vector<long long> h(8 * N);
    (void)hipMemcpy(h.data(), dbg_dev, 8 * 8 * N, hipMemcpyDeviceToHost);
    for (int i = 0; i < N; ++i) fprintf(stderr, "rf img %d C %lld kept %lld: compact %.1f sort %.1f decode %.1f matrix %.1f scan %.1f us\n", i, h[8*i+6], h[8*i+7], (h[8*i+1]-h[8*i])/100.0, (h[8*i+2]-h[8*i+1])/100.0, (h[8*i+3]-h[8*i+2])/100.0, (h[8*i+4]-h[8*i+3])/100.0, (h[8*i+5]-h[8*i+4])/100.0);
  }
  long long total = 0;
  for (int i = 0; i < N; ++i) total += counts[i];
  if (required) *required = (int32_t)total;
  if (total > capacity) return ta_fail(ctx, TA_E_CAPACITY, "retinaface: %lld detections, capacity %d", total, capacity);
  if (total == 0) return TA_OK;
  hipLaunchKernelGGL(rf_gather_kernel, dim3(N), dim3(256), 0, ctx->stream, p, N, (float*)(scr + o_ob), (float*)(scr + o_ol),
                     (float*)(scr + o_os));
  TA_HIP(ctx, hipGetLastError());
  TA_HIP(ctx, hipMemcpyAsync(boxes, scr + o_ob, (size_t)total * 16, hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipMemcpyAsync(landmarks, scr + o_ol, (size_t)total * 40, hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipMemcpyAsync(scores, scr + o_os, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return TA_OK;
}

extern "C" {

int ta_retinaface_run(ta_model* m, const ta_frames* frames, float score_thr, float nms_thr, int capacity,
                      int32_t* counts, float* boxes, float* landmarks, float* scores, int32_t* required) {
  ta_enter(m ? m->ctx : nullptr);
  if (!m || !frames || !counts) return TA_E_INVALID;
  ta_ctx* ctx = m->ctx;
  if (m->kind != TA_MODEL_RETINAFACE) return ta_fail(ctx, TA_E_INVALID, "retinaface_run: wrong model kind");
  if (capacity > 0 && (!boxes || !landmarks || !scores)) return ta_fail(ctx, TA_E_INVALID, "retinaface_run: null outputs");
  if (frames->n == 0) {
    if (required) *required = 0;
    return TA_OK;
  }
  TA_TRY(ta_model_forward_frames(m, frames));
  TA_TRY(ta_range_enqueue(ctx));                 // behind the network, ahead of the post-processing's syncs
  ta_tensor heads[3];
  for (int l = 0; l < 3; ++l) {
    heads[l] = m->tensors[m->hdr.outputs[l]];
    if (heads[l].unscale_dev || heads[l].fmt != TA_FMT_F32) return ta_fail(ctx, TA_E_INVALID, "retinaface_run: the head tensors must be plain float32");
  }
  const int rc = rf_postprocess_dev(ctx, heads, frames->n, frames->h, frames->w, 0, score_thr, nms_thr, capacity, counts, boxes,
                                    landmarks, scores, required);
  return ta_range_finish(ctx, rc);               // f16x3 layers: TA_E_RANGE when an activation left the half-float range
}

int ta_retinaface_postprocess(ta_ctx* ctx, const float* const heads[9], int n, int h, int w, float score_thr,
                              float nms_thr, int capacity, int32_t* counts, float* boxes, float* landmarks,
                              float* scores, int32_t* required) {
  ta_enter(ctx);
  if (!ctx || !heads || !counts || n < 0) return TA_E_INVALID;
  if (n == 0) {
    if (required) *required = 0;
    return TA_OK;
  }
  // repack the reference layout (NCHW: cls_prob 4ch, bbox 8ch, landmark 20ch per stride) to NHWC/32
  const int strides[3] = {32, 16, 8};
  ta_tensor t[3];
  size_t total = 0;
  for (int l = 0; l < 3; ++l) {
    t[l].n = n;
    t[l].h = (h + strides[l] - 1) / strides[l];
    t[l].w = (w + strides[l] - 1) / strides[l];
    t[l].c = 32;
    t[l].halo = 0;
    total += t[l].elems();
  }
  std::vector<float> host(total);
  size_t base = 0;
  std::vector<size_t> bases;
  for (int l = 0; l < 3; ++l) {
    bases.push_back(base);
    const int fh = t[l].h, fw = t[l].w;
    const int chs[3] = {4, 8, 20}, offs[3] = {0, 4, 12};
    for (int k = 0; k < 3; ++k) {
      const float* src = heads[3 * l + k];
      if (!src) return ta_fail(ctx, TA_E_INVALID, "retinaface_postprocess: null head %d", 3 * l + k);
      for (int i = 0; i < n; ++i)
        for (int c = 0; c < chs[k]; ++c)
          for (int y = 0; y < fh; ++y)
            for (int x = 0; x < fw; ++x)
              host[base + (((size_t)i * fh + y) * fw + x) * 32 + offs[k] + c] = src[(((size_t)i * chs[k] + c) * fh + y) * fw + x];
    }
    base += t[l].elems();
  }
  float* dev = nullptr;
  TA_HIP(ctx, hipMalloc((void**)&dev, total * sizeof(float)));
  hipError_t e = hipMemcpy(dev, host.data(), total * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(dev);
    return ta_fail(ctx, TA_E_DEVICE, "upload failed: %s", hipGetErrorString(e));
  }
  for (int l = 0; l < 3; ++l) t[l].dev = dev + bases[l];
  const int rc = rf_postprocess_dev(ctx, t, n, h, w, 1, score_thr, nms_thr, capacity, counts, boxes, landmarks, scores, required);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(dev);
  return rc;
}

}  // extern "C"

// Implicit-GEMM convolution for gfx950 (CDNA4), float32 in / float32 accumulate on the
// exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 157 TF peak).
//
// One kernel serves every dense conv of the three networks (RetinaFace 1x1/3x3,
// ArcFace 3x3 s1/s2 + 1x1 s2 shortcut + FC, OpenPose 3x3/7x7/1x1):
//
//   D[cout][pixel] = sum_k  W[cout][k] * X[k][pixel],     k = (ky, kx, cin)
//
// * Activations are NHWC float32 with a physical zero halo, so a filter tap is a constant
//   byte offset from a pixel's base address: no bounds checks in the K loop, and padding
//   costs nothing (reference convs are all "same"/zero padded, e.g. openpose/model.py:6-24).
// * K is cut in slabs of 32 floats = 8 chunks of 16 B; `ktab[slab*8+chunk]` is the byte offset
//   (tap + channel) of that chunk from the pixel base.  One table covers 1x1, 3x3, 7x7, strided,
//   channel-sliced and tiny-Cin (several taps per slab) convs alike.
// * Both operands are DMA'd straight into LDS with global_load_lds (16 B per lane, no VGPR
//   round trip), double buffered: slab s+1 streams in under the 64-cycle MFMAs of slab s.
//   The LDS image of a tile row is 128 B; the chunk a lane fetches is XOR-swizzled with
//   (row>>1)&7 on the SOURCE address (the DMA destination is lane-linear), which makes the
//   ds_read_b128 fragment reads bank-conflict free.
// * MFMA rows are output channels, columns are pixels: each lane ends up with 4 consecutive
//   channels of one pixel per accumulator quad -> 16-byte NHWC stores, fused bias / ReLU /
//   PReLU / residual (optionally nearest-x2 upsampled) / second affine output.
#include "ta_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32(const ta_conv_launch p) {
  constexpr int BN = WAVES_M * WM_TILES * 32;   // output channels per workgroup
  constexpr int BM = WAVES_N * WN_TILES * 32;   // pixels per workgroup
  constexpr int QA = BN / 32;                   // A-tile DMA instructions per wave per slab
  constexpr int QB = BM / 32;                   // B-tile DMA instructions per wave per slab
  constexpr int STAGE = (BN + BM) * 32;         // floats per pipeline stage
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");

  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;

  // XCD-aware tile order: all cout-tiles of one pixel-tile run back to back on one XCD
  // (block b lands on XCD b%8), so the activation tile is fetched into one L2 only.
  const int n_ct = p.coutp / BN;
  const int bid = blockIdx.x;
  const int grp = bid >> 3, xcd = bid & 7;
  const int ct = grp % n_ct;
  const int pt = (grp / n_ct) * 8 + xcd;
  const int n_pt = (p.M + BM - 1) / BM;
  if (pt >= n_pt) return;
  const int ct0 = ct * BN;
  const int pt0 = pt * BM;

  // ---- per-lane DMA source bases -------------------------------------------------------
  // DMA instruction t (0..(BN+BM)/8) copies 8 tile rows x 128 B; lane -> (row = t*8 + lane/8,
  // physical chunk = lane%8).  Instruction t = q*4 + wave.
  const int pchunk = lane & 7;
  const int lchunk = pchunk ^ ((4 * (wave & 1) + (lane >> 4)) & 7);   // logical chunk fetched
  const int HoWo = p.Ho * p.Wo;

  const char* a_src[QA];
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    const int row = (q * 4 + wave) * 8 + (lane >> 3);
    a_src[q] = (const char*)(p.w + ((size_t)(ct0 + row)) * 32 + lchunk * 4);
  }
  const char* b_src[QB];
#pragma unroll
  for (int q = 0; q < QB; ++q) {
    const int row = (q * 4 + wave) * 8 + (lane >> 3);
    int pix = pt0 + row;
    if (pix >= p.M) pix = 0;                       // clamp: value unused (store is masked)
    const int img = pix / HoWo;
    const int rem = pix - img * HoWo;
    const int y = rem / p.Wo;
    const int x = rem - y * p.Wo;
    const size_t off = (size_t)img * p.in_img + (size_t)(y * p.stride) * p.in_row +
                       (size_t)(x * p.stride) * p.in_pix + p.in_off0;
    b_src[q] = (const char*)(p.in + off);
  }
  const size_t a_slab_bytes = (size_t)p.coutp * 128;

  auto issue = [&](int s, int stage, int koff) {
    float* base = lds + stage * STAGE;
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int t = q * 4 + wave;
      __builtin_amdgcn_global_load_lds(GLB_PTR(a_src[q] + (size_t)s * a_slab_bytes),
                                       LDS_PTR(base + t * 256), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int t = q * 4 + wave;
      __builtin_amdgcn_global_load_lds(GLB_PTR(b_src[q] + koff), LDS_PTR(base + BN * 32 + t * 256),
                                       16, 0, 0);
    }
  };

  f32x16 acc[WM_TILES][WN_TILES];
#pragma unroll
  for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
    for (int b = 0; b < WN_TILES; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment read addresses: row = tile row (lane&31), logical chunk = (lane>>5)*4 + g,
  // physical chunk = logical ^ ((row>>1)&7); tile bases are multiples of 32 rows.
  const int frow = lane & 31;
  const int fsw = (frow >> 1) & 7;
  const int fcb = (lane >> 5) * 4;
  const int a_row0 = wm * WM_TILES * 32 + frow;
  const int b_row0 = BN + wn * WN_TILES * 32 + frow;

  const int S = p.n_slabs;
  int kt = p.ktab[lchunk];
  issue(0, 0, kt);
  int kt_next = (S > 1) ? p.ktab[8 + lchunk] : 0;

  for (int s = 0; s < S; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // slab s landed for every wave; everyone is done reading the other stage
    if (s + 1 < S) {
      issue(s + 1, (s + 1) & 1, kt_next);
      if (s + 2 < S) kt_next = p.ktab[(s + 2) * 8 + lchunk];
    }
    const float* st = lds + (s & 1) * STAGE;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int pc = ((fcb + g) ^ fsw) * 4;
      f32x4 av[WM_TILES], bv[WN_TILES];
#pragma unroll
      for (int a = 0; a < WM_TILES; ++a) av[a] = *(const f32x4*)(st + (a_row0 + a * 32) * 32 + pc);
#pragma unroll
      for (int b = 0; b < WN_TILES; ++b) bv[b] = *(const f32x4*)(st + (b_row0 + b * 32) * 32 + pc);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
          for (int b = 0; b < WN_TILES; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][e], bv[b][e], acc[a][b], 0, 0, 0);
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------
  // acc[a][b][r]: pixel = tile col (lane&31); cout = 8*(r>>2) + 4*(lane>>5) + (r&3) within the tile.
  // Loads are grouped ahead of the math and only the stores are predicated, so the compiler can
  // keep them all in flight instead of waiting per access.
  const int co_base = ct0 + wm * WM_TILES * 32 + 4 * (lane >> 5);
  f32x4 bias[WM_TILES][4];
#pragma unroll
  for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) bias[a][j] = *(const f32x4*)(p.bias + co_base + a * 32 + 8 * j);   // padded to coutp
  f32x4 slope[WM_TILES][4];
  if (p.act == TA_ACT_PRELU) {
#pragma unroll
    for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) slope[a][j] = *(const f32x4*)(p.prelu + co_base + a * 32 + 8 * j);
  }
  const int co_max = p.cout - 4;
#pragma unroll
  for (int b = 0; b < WN_TILES; ++b) {
    const int pix_raw = pt0 + wn * WN_TILES * 32 + b * 32 + (lane & 31);
    const bool pix_ok = pix_raw < p.M;
    const int pix = pix_ok ? pix_raw : 0;
    const int img = pix / HoWo;
    const int rem = pix - img * HoWo;
    const int y = rem / p.Wo;
    const int x = rem - y * p.Wo;
    f32x4 v[WM_TILES][4];
#pragma unroll
    for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[a][j][e] = acc[a][b][4 * j + e] + bias[a][j][e];
    if (p.act == TA_ACT_RELU) {
#pragma unroll
      for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[a][j][e] = v[a][j][e] > 0.f ? v[a][j][e] : 0.f;
    } else if (p.act == TA_ACT_PRELU) {
#pragma unroll
      for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[a][j][e] = v[a][j][e] > 0.f ? v[a][j][e] : v[a][j][e] * slope[a][j][e];
    }
    if (p.res) {
      const int ry = p.res_up2 ? (y >> 1) : y, rx = p.res_up2 ? (x >> 1) : x;
      const float* rs = p.res + (size_t)img * p.res_img + (size_t)ry * p.res_row + (size_t)rx * p.res_pix + p.res_off0;
      f32x4 r4[WM_TILES][4];
#pragma unroll
      for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = co_base + a * 32 + 8 * j;
          r4[a][j] = *(const f32x4*)(rs + (co < co_max ? co : co_max));   // clamped: masked at the store
        }
#pragma unroll
      for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[a][j][e] += r4[a][j][e];
    }
    float* o = p.out + (size_t)img * p.out_img + (size_t)y * p.out_row + (size_t)x * p.out_pix + p.out_off0;
    if (pix_ok) {
#pragma unroll
      for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = co_base + a * 32 + 8 * j;
          if (co < p.cout) *(f32x4*)(o + co) = v[a][j];
        }
    }
    if (p.out2) {
      float* o2 = p.out2 + (size_t)img * p.o2_img + (size_t)y * p.o2_row + (size_t)x * p.o2_pix + p.o2_off0;
      f32x4 sc[WM_TILES][4], sh[WM_TILES][4];
#pragma unroll
      for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sc[a][j] = *(const f32x4*)(p.scale2 + co_base + a * 32 + 8 * j);   // padded to coutp
          sh[a][j] = *(const f32x4*)(p.shift2 + co_base + a * 32 + 8 * j);
        }
      if (pix_ok) {
#pragma unroll
        for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int co = co_base + a * 32 + 8 * j;
            f32x4 z;
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = v[a][j][e] * sc[a][j][e] + sh[a][j][e];
            if (co < p.cout) *(f32x4*)(o2 + co) = z;
          }
      }
    }
  }
}

template <int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES>
static int launch_cfg(ta_ctx* ctx, const ta_conv_launch& p) {
  constexpr int BN = WAVES_M * WM_TILES * 32;
  constexpr int BM = WAVES_N * WN_TILES * 32;
  const int n_ct = p.coutp / BN;
  const int n_pt = (p.M + BM - 1) / BM;
  const int groups = ((n_pt + 7) / 8) * n_ct;
  const size_t lds_bytes = 2 * (size_t)(BN + BM) * 32 * sizeof(float);
  auto kern = conv_igemm_f32<WAVES_M, WAVES_N, WM_TILES, WN_TILES>;
  static bool attr_set = false;
  if (!attr_set) {
    TA_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_bytes));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(groups * 8), dim3(256), lds_bytes, ctx->stream, p);
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

int ta_launch_conv(ta_ctx* ctx, const ta_conv_launch& p, double flops) {
  if (p.M <= 0) return TA_OK;
  if (p.coutp % 32 != 0 || p.cout % 4 != 0) return ta_fail(ctx, TA_E_INVALID, "conv: bad cout padding");
  ta_prof_scope scope(ctx, 0, flops);
  if (p.coutp % 128 == 0) return launch_cfg<2, 2, 2, 2>(ctx, p);
  if (p.coutp % 64 == 0) return launch_cfg<1, 4, 2, 1>(ctx, p);
  return launch_cfg<1, 4, 1, 1>(ctx, p);
}

// Implicit-GEMM convolution for gfx950 (CDNA4): float32 accumulate on the exact-f32 matrix pipe
// (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 157 TF peak) or on the bf16 pipe with split operands
// (x = hi + lo, three v_mfma_f32_32x32x16_bf16 per product term: float32-class accuracy; modes below).
//
// One kernel serves every dense conv of the three networks (RetinaFace 1x1/3x3,
// ArcFace 3x3 s1/s2 + 1x1 s2 shortcut + FC, OpenPose 3x3/7x7/1x1):
//
//   D[cout][pixel] = sum_k  W[cout][k] * X[k][pixel],     k = (ky, kx, cin)
//
// * Activations are NHWC, 4 bytes per element, with a physical zero halo, so a filter tap is a constant
//   byte offset from a pixel's base address: no bounds checks in the K loop, and padding
//   costs nothing (reference convs are all "same"/zero padded, e.g. openpose/model.py:6-24).
// * K is cut in slabs of 32 floats = 8 chunks of 16 B; `ktab[slab*8+chunk]` is the byte offset
//   (tap + channel) of that chunk from the pixel base.  One table covers 1x1, 3x3, 7x7, strided,
//   channel-sliced and tiny-Cin (several taps per slab) convs alike.
// * Both operands are DMA'd straight into LDS with global_load_lds (16 B per lane, no VGPR
//   round trip), 2 or 3 LDS stages: later slabs stream in under the MFMAs of slab s.
//   The LDS image of a tile row is 128 B; the chunk a lane fetches is XOR-swizzled with
//   (row>>1)&7 on the SOURCE address (the DMA destination is lane-linear), which makes the
//   ds_read_b128 fragment reads bank-conflict free.
// * MFMA rows are output channels, columns are pixels: each lane ends up with 4 consecutive
//   channels of one pixel per accumulator quad -> 16-byte NHWC stores, fused bias / ReLU /
//   PReLU / residual (optionally nearest-x2 upsampled) / second affine output.
#include <stdlib.h>
#include <string.h>

#include "act_format.h"
#include "ta_internal.h"

#ifdef TA_CONV_TRACE
// Debug build only (TA_EXTRA_FLAGS=-DTA_CONV_TRACE): cycle stamps of workgroup 0 of the split kernel.
__device__ long long ta_trace_buf[64];
// which workgroup is stamped: TA_CONV_PROBE >> 8 (0 = the first one, which starts on an idle chip; a middle block sees the loaded one)
#define TA_STAMP(i)                                                              \
  do {                                                                           \
    if (blockIdx.x == (unsigned)(p.probe >> 8) && (threadIdx.x & 63) == 0) ta_trace_buf[(i)] = __builtin_readcyclecounter(); \
  } while (0)
extern "C" int ta_debug_trace_read(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ta_trace_buf), sizeof(long long) * n);
}
#else
#define TA_STAMP(i) do { } while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Arithmetic modes of the MFMA inner loop (activation tensors are float32, or pre-split bf16 hi|lo words in the
// bf16 modes: act_format.h):
//   PREC_F32    : v_mfma_f32_32x32x2_f32, exact f32 products.                          157 TF peak
//   PREC_BF16X3 : x = hi + lo (two bf16), products hi*hi + hi*lo + lo*hi on
//                 v_mfma_f32_32x32x16_bf16, f32 accumulate: ~1e-5 relative per product,
//                 i.e. float32-class accuracy at 3/16 of the f32 MFMA cost.              833 TF-equivalent peak
//   PREC_BF16   : hi*hi only (throughput mode, NOT within the 1e-3 parity bar).          2.5 PF peak
//   PREC_F16    : hi*hi only on IEEE half words (11 bits per operand, 2^-11 per product): for networks that take no discrete
//                 decision and whose outputs have a tolerance -- ArcFace: 3e-4 on unit-norm embedding components against a
//                 1e-3 bar (tests/probe_embed_precision.py); same weight scaling and range flag as PREC_F16X3.
//   PREC_F16X3  : the same three-product scheme with IEEE half words on v_mfma_f32_32x32x16_f16: x = hi + lo carries
//                 22 significant bits, every product hi*hi / hi*lo / lo*hi is exact in the float32 accumulator and the
//                 dropped lo*lo term is <= 2^-22 of the product -- below the rounding noise of a float32 dot product.
//                 Same MFMA count and rate as PREC_BF16X3.  Half floats end at 65504: weights are packed times a power of
//                 two per layer (their lo halves stay normal numbers; the epilogue multiplies the sums back, exactly) and
//                 an epilogue that would store |x| > 65504 raises the context's range flag (TA_E_RANGE) instead.
// Weights are split at pack time ([hi x32 | lo x32] 16-bit words per 128-byte row).  Activations either arrive in the
// same image (TA_FMT_SPLIT / TA_FMT_SPLIT16, written by the producer's epilogue) or are float32 and split in registers
// right after the ds_read.
//   PREC_F16X2  : TWO of the three products on the same operands as PREC_F16X3: (w_hi + w_lo) * x_hi -- the weights keep their 22 bits,
//                 every activation enters the contraction rounded to its hi half (11 bits; the `lo` words of the pre-split tensors
//                 are simply not read, so the shortcut trunk of a residual network still carries 22 bits from unit to unit).  A
//                 tolerance mode for networks that take no discrete decision (the embedder: tests/probe_embed_2mfma.py), 2/3 of the
//                 MFMAs of PREC_F16X3; tensors, weight image, scales and range flag are PREC_F16X3's.
enum { PREC_F32 = 0, PREC_BF16X3 = 1, PREC_BF16 = 2, PREC_F16X3 = 3, PREC_F16 = 4, PREC_F16X2 = 5 };
__host__ __device__ constexpr bool prec_x3(int prec) { return prec == PREC_BF16X3 || prec == PREC_F16X3; }
__host__ __device__ constexpr bool prec_x2(int prec) { return prec == PREC_F16X2; }                        // w_lo * x_hi + w_hi * x_hi
__host__ __device__ constexpr bool prec_half(int prec) { return prec == PREC_F16X3 || prec == PREC_F16 || prec == PREC_F16X2; }   // IEEE half words (else bf16)
__host__ __device__ constexpr int prec_nmma(int prec) { return prec_x3(prec) ? 3 : (prec_x2(prec) ? 2 : 1); }   // MFMAs per product term (16-bit modes)

// one 32x32x16 MFMA on 16-bit operand fragments held as raw bits (bf16x8 is the container type for both formats)
template <int PREC>
__device__ __forceinline__ f32x16 ta_mfma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (prec_half(PREC))
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// float32 -> the mode's 16-bit word (round to nearest even) and back, for operands split in registers
template <int PREC>
__device__ __forceinline__ __bf16 ta_to16(float x) {
  if constexpr (prec_half(PREC)) return __builtin_bit_cast(__bf16, (_Float16)x);
  else return (__bf16)x;
}
template <int PREC>
__device__ __forceinline__ float ta_from16(__bf16 h) {
  if constexpr (prec_half(PREC)) return (float)__builtin_bit_cast(_Float16, h);
  else return (float)h;
}

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 16 B per lane global -> LDS, address = uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset (ONE VGPR),
// LDS destination (wave-uniform) through M0.  The two-VGPR address form the builtin emits costs the DMA stream a
// quarter of its rate when MFMAs run on the same SIMD (VGPR read-port contention; tools/probe/dma_mfma_probe.hip:
// 4.8 vs 6.0 B/clk per issuing wave), the saddr form none.
__device__ __forceinline__ void ta_dma16(const char* ubase, unsigned lane_off, const float* lds_dst) {
  const unsigned ldsa = (unsigned)(size_t)LDS_PTR(lds_dst);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(ubase), "s"(ldsa)
               : "memory", "m0");
}

// Block b runs on XCD b % 8 (each XCD has its own L2).  XCD x owns a CONTIGUOUS run of pixel tiles (balanced split
// of n_pt over the 8 XCDs), walked cout-tile fastest: the workgroups of one XCD that are in flight together work on
// neighbouring image rows, so the 3x3 / 7x7 halo rows and the activation tile shared by all cout tiles are fetched
// into ONE L2 once instead of into up to 8 of them.
__device__ __forceinline__ int ta_xcd_tile(int n_pt, int xcd, int local) {
  const int base = n_pt >> 3, rem = n_pt & 7;
  if (local >= base + (xcd < rem ? 1 : 0)) return -1;
  return xcd * base + (xcd < rem ? xcd : rem) + local;
}

// A conv that FOLDS a per-channel affine of its INPUT (ArcFace's BatchNorm in front of a zero-padded 3x3 conv,
// arcface/model.py:12-14) into its weights needs a bias that depends on which filter taps fall into the padding: the
// shift reaches the sum only through in-bounds taps.  Per axis a pixel is first / middle / last -- or the only one (maps of
// one row or column: both outer taps are padding): 4 x 4 classes, bias9[class][coutp] (nine of them occur on maps of two or
// more rows and columns), TA_INTERIOR (middle, middle) == the ordinary bias.  3x3, stride 1, pad 1 only.
#define TA_INTERIOR 5
__device__ __forceinline__ int ta_border_class(int y, int x, int Ho, int Wo) {
  const int cy = Ho == 1 ? 3 : (y == 0 ? 0 : (y == Ho - 1 ? 2 : 1));
  const int cx = Wo == 1 ? 3 : (x == 0 ? 0 : (x == Wo - 1 ? 2 : 1));
  return cy * 4 + cx;
}

// ---- range guard of the half-float programs -----------------------------------------------------------------------------------
// Every epilogue of a program with half-float convs (ta_conv_launch::range_check) tracks the largest |x| it STORES -- whatever
// the op's own arithmetic mode and the tensor's format: a float32 tensor written by an exact-f32 op may be split into half
// floats in registers by its consumer.  The maximum is taken on BIT PATTERNS (sign cleared): for non-negative floats integer
// order is float order, and inf / NaN sort above every finite value -- fmaxf would drop a NaN and let it through.
#define TA_F16_MAX_BITS 0x477FE000u               /* 65504.0f */
__device__ __forceinline__ unsigned ta_absbits(float x) { return __float_as_uint(x) & 0x7FFFFFFFu; }
__device__ __forceinline__ unsigned ta_amax4(unsigned m, const f32x4& v) {
  return max(max(m, max(ta_absbits(v[0]), ta_absbits(v[1]))), max(ta_absbits(v[2]), ta_absbits(v[3])));
}
// end of an epilogue: raise the flag; tools (ta_model_debug_amax) also collect the maximum itself per op
__device__ __forceinline__ void ta_range_report(const ta_conv_launch& p, unsigned amax) {
  if (amax > TA_F16_MAX_BITS) *p.range_flag = 1;
  // tools: the slots live behind the flag word (ta_ctx::range_flag, TA_AMAX_SLOT0).  Every lane reports its own maximum: callers
  // reach this point with part of the wave already returned, so a cross-lane reduction here would read exited lanes
  if (p.amax_index >= 0 && amax) atomicMax((unsigned*)p.range_flag + TA_AMAX_SLOT0 + 2 * p.amax_index, amax);
}
// ReLU that keeps a NaN a NaN (`v > 0 ? v : 0` turns it into 0 and hides it from the range guard)
__device__ __forceinline__ float ta_relu(float v) { return v < 0.f ? 0.f : v; }

// Fused epilogue shared by both kernels.  acc[a][b][r]: pixel = tile col (lane&31);
// cout = 8*(r>>2) + 4*(lane>>5) + (r&3) within the 32x32 tile.
template <int WM_TILES, int WN_TILES>
__device__ __forceinline__ void conv_epilogue(const ta_conv_launch& p, f32x16 (&acc)[WM_TILES][WN_TILES], int co_tile0,
                                              int pix_tile0, int lane, int HoWo) {
  // ---- epilogue ------------------------------------------------------------------------------
  // acc[a][b][r]: pixel = tile col (lane&31); cout = 8*(r>>2) + 4*(lane>>5) + (r&3) within the tile.
  // Loads are grouped ahead of the math and only the stores are predicated, so the compiler can
  // keep them all in flight instead of waiting per access.
  const int co_base = co_tile0 + 4 * (lane >> 5);
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
  // per-channel power of two (weight-row exponent and activation scale of the channel written, ta_op_desc.wus_off): the
  // accumulators are scaled in place, one short-lived vector at a time -- exact, so acc * us + bias rounds once like the fma
  // would, and no [WM_TILES][4] vector array stays live next to the bias through the epilogue (register pressure of the kernel)
#pragma unroll
  for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 u = *(const f32x4*)((p.bias + p.coutp) + co_base + a * 32 + 8 * j);
#pragma unroll
      for (int b = 0; b < WN_TILES; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[a][b][4 * j + e] *= u[e];
    }
  unsigned amax = 0;                              // largest |x| stored, as a bit pattern (ta_range_report)
#pragma unroll
  for (int b = 0; b < WN_TILES; ++b) {
    const int pix_raw = pix_tile0 + b * 32 + (lane & 31);
    const int pixc = pix_raw < p.M ? pix_raw : 0;
    const int img = pixc / HoWo;
    const int rem = pixc - img * HoWo;
    const int y = rem / p.Wo;
    const int x = rem - y * p.Wo;
    const bool pix_ok = pix_raw < p.M;
    f32x4 v[WM_TILES][4];
#pragma unroll
    for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[a][j][e] = acc[a][b][4 * j + e] + bias[a][j][e];
    if (p.bias9) {                                      // border pixels: the class's bias instead (see ta_border_class)
      const int cls = ta_border_class(y, x, p.Ho, p.Wo);
      if (cls != TA_INTERIOR) {
#pragma unroll
        for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 b9 = *(const f32x4*)(p.bias9 + (size_t)cls * p.coutp + co_base + a * 32 + 8 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[a][j][e] = acc[a][b][4 * j + e] + b9[e];
          }
      }
    }
    if (p.act == TA_ACT_RELU) {
#pragma unroll
      for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[a][j][e] = ta_relu(v[a][j][e]);
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
          r4[a][j] = ta_ld4(rs, p.res_ch + (co < co_max ? co : co_max), p.res_fmt);   // clamped: masked at the store
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
          if (co < p.cout) {
            ta_st4(o, p.out_ch + co, p.out_fmt, v[a][j]);
            if (p.range_check) amax = ta_amax4(amax, v[a][j]);
          }
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
            if (co < p.cout) {
              ta_st4(o2, p.o2_ch + co, p.o2_fmt, z);
              if (p.range_check) amax = ta_amax4(amax, z);
            }
          }
      }
    }
  }
  ta_range_report(p, amax);
}

// ---- LDS-staged epilogue of the symmetric-wave kernels (conv_igemm, conv_igemm_pipe, conv_dwpw) ---------------------
// The direct epilogue above stores straight from the accumulators: a store instruction scatters 16 B to 32 different
// pixels (32 cache lines per instruction).  Like the split-role kernel, these kernels now park the raw tile in the LDS ring
// they are done with -- [pixel][cout], 16-byte chunks XOR-swizzled with the pixel row -- and drain it with one lane per
// (pixel, 8 consecutive channels), whole 128-byte lines per instruction, through the same conv_epilogue_drain.  Same
// arithmetic in the same order (fma(acc, unscale, bias), activation, shortcut, second output): same bits.
template <int BN, int BM, int NT>
__device__ __forceinline__ void conv_epilogue_drain(const ta_conv_launch& p, const float* lds, int ct0, int pt0, int tid,
                                                    int HoWo, int ks);

// DIRECT_OK: the kernel also carries the direct epilogue (stores straight from the accumulators) for channel slices that are
// not on 8-channel boundaries.  Only the generic kernel does: the fallback sets the register budget of whatever kernel it is
// compiled into (its bias / slope / value arrays are live next to all accumulators), and no layer of the three networks takes it.
template <int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES, bool DIRECT_OK = false>
__device__ __forceinline__ void conv_finish_sym(const ta_conv_launch& p, f32x16 (&acc)[WM_TILES][WN_TILES], float* lds, int ct0,
                                                int pt0, int wm, int wn, int tid, int lane, int HoWo) {
  constexpr int BN = WAVES_M * WM_TILES * 32, BM = WAVES_N * WN_TILES * 32, NCH = BN / 4;
  if constexpr (DIRECT_OK) {
    const bool staged = ((p.out_ch | p.res_ch | p.o2_ch | p.direct_epilogue) & 7) == 0 && (p.cout & 3) == 0;
    if (!staged) {
      conv_epilogue<WM_TILES, WN_TILES>(p, acc, ct0 + wm * WM_TILES * 32, pt0 + wn * WN_TILES * 32, lane, HoWo);
      return;
    }
  }
  __syncthreads();                                  // every wave is done reading operand fragments: the ring is free
  if (tid == 0) TA_STAMP(21);
#pragma unroll
  for (int b = 0; b < WN_TILES; ++b) {
    const int row = (wn * WN_TILES + b) * 32 + (lane & 31);
#pragma unroll
    for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = ((wm * WM_TILES + a) * 32 + 8 * j + 4 * (lane >> 5)) >> 2;
        *(f32x4*)(lds + (row * NCH + (c ^ (row & (NCH - 1)))) * 4) =
            f32x4{acc[a][b][4 * j], acc[a][b][4 * j + 1], acc[a][b][4 * j + 2], acc[a][b][4 * j + 3]};
      }
  }
  __syncthreads();
  if (tid == 0) TA_STAMP(22);                       // tile parked
  conv_epilogue_drain<BN, BM, 256>(p, lds, ct0, pt0, tid, HoWo, 0);
}

// One K slab (32) of a symmetric-wave tile: A fragments from the packed weight rows, B fragments from float32 pixel rows
// (split into 16-bit hi / lo words in registers in the split modes).  Shared by conv_igemm and conv_dwpw.
template <int WM_TILES, int WN_TILES, int PREC>
__device__ __forceinline__ void conv_slab_mma(const float* st, f32x16 (&acc)[WM_TILES][WN_TILES], int a_row0, int b_row0, int fsw,
                                              int fcb, int lane) {
  if constexpr (PREC == PREC_F32) {
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
  } else {
    // K-step t covers k = 16*kgrp + 8t + (0..7): weight chunk 2*kgrp+t (hi) / 4+2*kgrp+t (lo),
    // activation float chunks kgrp*4 + 2t and kgrp*4 + 2t + 1.
    const int kg = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16x8 ah[WM_TILES], al[WM_TILES], bh[WN_TILES], bl[WN_TILES];
#pragma unroll
      for (int a = 0; a < WM_TILES; ++a) {
        ah[a] = *(const bf16x8*)(st + (a_row0 + a * 32) * 32 + ((2 * kg + t) ^ fsw) * 4);
        if constexpr (prec_x3(PREC) || prec_x2(PREC)) al[a] = *(const bf16x8*)(st + (a_row0 + a * 32) * 32 + ((4 + 2 * kg + t) ^ fsw) * 4);
      }
#pragma unroll
      for (int b = 0; b < WN_TILES; ++b) {
        const f32x4 x0 = *(const f32x4*)(st + (b_row0 + b * 32) * 32 + ((fcb + 2 * t) ^ fsw) * 4);
        const f32x4 x1 = *(const f32x4*)(st + (b_row0 + b * 32) * 32 + ((fcb + 2 * t + 1) ^ fsw) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const __bf16 h0 = ta_to16<PREC>(x0[e]), h1 = ta_to16<PREC>(x1[e]);
          bh[b][e] = h0;
          bh[b][4 + e] = h1;
          if constexpr (prec_x3(PREC)) {
            bl[b][e] = ta_to16<PREC>(x0[e] - ta_from16<PREC>(h0));
            bl[b][4 + e] = ta_to16<PREC>(x1[e] - ta_from16<PREC>(h1));
          }
        }
      }
      if constexpr (prec_x3(PREC) || prec_x2(PREC)) {
#pragma unroll
        for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
          for (int b = 0; b < WN_TILES; ++b)
            acc[a][b] = ta_mfma16<PREC>(al[a], bh[b], acc[a][b]);
      }
      if constexpr (prec_x3(PREC)) {
#pragma unroll
        for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
          for (int b = 0; b < WN_TILES; ++b)
            acc[a][b] = ta_mfma16<PREC>(ah[a], bl[b], acc[a][b]);
      }
#pragma unroll
      for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
        for (int b = 0; b < WN_TILES; ++b)
          acc[a][b] = ta_mfma16<PREC>(ah[a], bh[b], acc[a][b]);
    }
  }
}

template <int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES, int PREC, bool DIRECT = false>
__global__ __launch_bounds__(256, 2) void conv_igemm(const ta_conv_launch p) {
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
  const int n_pt = (p.M + BM - 1) / BM;
  const int pt = ta_xcd_tile(n_pt, xcd, grp / n_ct);
  if (pt < 0) return;
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
    conv_slab_mma<WM_TILES, WN_TILES, PREC>(st, acc, a_row0, b_row0, fsw, fcb, lane);
  }

  conv_finish_sym<WAVES_M, WAVES_N, WM_TILES, WN_TILES, DIRECT>(p, acc, lds, ct0, pt0, wm, wn, tid, lane, HoWo);
}

// K-slab order of the uniform-K kernels (conv_igemm_pipe, conv_igemm_split): channel block OUTERMOST, then ky, then kx.
// A workgroup re-reads its input patch once per filter tap; with the channel block innermost (the packed weight order)
// every slab touches another 128-byte block of every patch pixel, so the bytes a 128 x 256-pixel tile keeps coming back to
// are the whole patch (150 KB at 23 x 40 x 128 ch; 32 co-resident tiles per XCD = 4.8 MB against a 4 MiB L2: measured
// 2.8-4x the fabric reads of the 128 x 128 tiling, tools/fetch_probe.sh).  Walking all kh x kw taps of ONE channel block
// before moving on shrinks that to 1 / cblocks of it.  Every uniform-K kernel uses this one order, so a layer's float
// summation order -- and with it every output bit -- does not depend on which of them the launcher picks for a batch size.
// Slab s' of the walk is packed weight slab (tap * cblocks + cb).  All scalar (wave-uniform) arithmetic.
struct ta_k_walk {
  int cb, kx, ky, b_off, a_slab;
  int cblocks, kw, kh, pix_bytes, row_bytes;
  __device__ __forceinline__ ta_k_walk(const ta_conv_launch& p, int s0) {
    cblocks = p.k_cblocks;
    kw = p.k_w;
    kh = p.k_h;
    pix_bytes = p.in_pix * 4;
    row_bytes = p.in_row * 4;
    if (s0 == 0) {                               // every launch but the K-split ones starts at slab 0: no division
      cb = kx = ky = b_off = a_slab = 0;
      return;
    }
    const int taps = kw * kh;
    cb = s0 / taps;
    const int tap = s0 - cb * taps;
    ky = tap / kw;
    kx = tap - ky * kw;
    b_off = cb * 128 + kx * pix_bytes + ky * row_bytes;
    a_slab = tap * cblocks + cb;
  }
  __device__ __forceinline__ void advance() {
    ++kx;
    b_off += pix_bytes;
    a_slab += cblocks;
    if (kx == kw) {
      kx = 0;
      b_off += row_bytes - kw * pix_bytes;
      if (++ky == kh) {
        ky = 0;
        ++cb;
        b_off += 128 - kh * row_bytes;
        a_slab = cb;
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------
// Deep-pipelined variant for convs whose K slabs never straddle a filter tap (cin % 32 == 0: every heavy
// layer).  Differences from conv_igemm above:
//  * the per-slab source offset is walked with scalar counters (channel block -> kx -> ky), so the K loop
//    contains no ordinary global load at all (hipcc would otherwise drain the DMA queue with vmcnt(0) at
//    the load's first use);
//  * STAGES LDS buffers, raw s_barrier and a COUNTED s_waitcnt vmcnt(N): STAGES-1 slabs stay in flight
//    across the barrier, which is what hides the L2/HBM latency once the MFMA work per slab shrinks
//    (bf16x3 / bf16: 768 / 256 MFMA cycles per slab instead of 4096).
template <int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES, int PREC, int STAGES, bool BSPLIT>
__global__ __launch_bounds__(256, (STAGES * (WAVES_M * WM_TILES + WAVES_N * WN_TILES) * 32 * 128 <= 80 * 1024) ? 2 : 1) void conv_igemm_pipe(const ta_conv_launch p) {
  constexpr int BN = WAVES_M * WM_TILES * 32;
  constexpr int BM = WAVES_N * WN_TILES * 32;
  constexpr int QA = BN / 32;
  constexpr int QB = BM / 32;
  constexpr int NI = QA + QB;                    // DMA instructions per wave per slab
  constexpr int STAGE = (BN + BM) * 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  static_assert(STAGES >= 2 && STAGES <= 4, "2..4 stages");

  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;

  const int n_ct = p.coutp / BN;
  const int bid = blockIdx.x;
  const int grp = bid >> 3, xcd = bid & 7;
  const int ct = grp % n_ct;
  const int n_pt = (p.M + BM - 1) / BM;
  const int pt = ta_xcd_tile(n_pt, xcd, grp / n_ct);
  if (pt < 0) return;
  const int ct0 = ct * BN;
  const int pt0 = pt * BM;

  const int pchunk = lane & 7;
  const int lchunk = pchunk ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
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
    if (pix >= p.M) pix = 0;
    const int img = pix / HoWo;
    const int rem = pix - img * HoWo;
    const int y = rem / p.Wo;
    const int x = rem - y * p.Wo;
    const size_t off = (size_t)img * p.in_img + (size_t)(y * p.stride) * p.in_row +
                       (size_t)(x * p.stride) * p.in_pix + p.in_off0 + p.in_ch_off;
    b_src[q] = (const char*)(p.in + off) + lchunk * 16;
  }
  const size_t a_slab_bytes = (size_t)p.coutp * 128;

  ta_k_walk kw_(p, 0);                            // the next slab to issue (ta_k_walk: channel block outermost)
  auto issue = [&](int, int stage) {
    float* base = lds + stage * STAGE;
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int t = q * 4 + wave;
      __builtin_amdgcn_global_load_lds(GLB_PTR(a_src[q] + (size_t)kw_.a_slab * a_slab_bytes), LDS_PTR(base + t * 256), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int t = q * 4 + wave;
      __builtin_amdgcn_global_load_lds(GLB_PTR(b_src[q] + kw_.b_off), LDS_PTR(base + BN * 32 + t * 256), 16, 0, 0);
    }
    kw_.advance();
  };

  f32x16 acc[WM_TILES][WN_TILES];
#pragma unroll
  for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
    for (int b = 0; b < WN_TILES; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int frow = lane & 31;
  const int fsw = (frow >> 1) & 7;
  const int fcb = (lane >> 5) * 4;
  const int a_row0 = wm * WM_TILES * 32 + frow;
  const int b_row0 = BN + wn * WN_TILES * 32 + frow;
  const int kg = lane >> 5;

  // Register-level software pipeline on top of the LDS ring: while the MFMAs of slab s run from one
  // fragment set, the other set is filled from LDS (ds_read_b128) and split into bf16 hi/lo for slab s+1,
  // so matrix pipe, LDS and VALU of ONE wave overlap instead of serialising (measured additive before:
  // MFMA 37 % + conversion 25 % + DMA 23 % + reads/barrier 36 % of a 7x7 layer).
  struct Frag {
    f32x4 a32[WM_TILES][4], b32[WN_TILES][4];                      // raw 16-byte chunks (f32 mode uses them directly)
    bf16x8 ah[WM_TILES][2], al[WM_TILES][2], bh[WN_TILES][2], bl[WN_TILES][2];
  };
  auto load_raw = [&](Frag& f, const float* st) {
    if constexpr (PREC == PREC_F32) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int pc = ((fcb + g) ^ fsw) * 4;
#pragma unroll
        for (int a = 0; a < WM_TILES; ++a) f.a32[a][g] = *(const f32x4*)(st + (a_row0 + a * 32) * 32 + pc);
#pragma unroll
        for (int b = 0; b < WN_TILES; ++b) f.b32[b][g] = *(const f32x4*)(st + (b_row0 + b * 32) * 32 + pc);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int a = 0; a < WM_TILES; ++a) {
          f.ah[a][t] = *(const bf16x8*)(st + (a_row0 + a * 32) * 32 + ((2 * kg + t) ^ fsw) * 4);
          if constexpr (prec_x3(PREC) || prec_x2(PREC)) f.al[a][t] = *(const bf16x8*)(st + (a_row0 + a * 32) * 32 + ((4 + 2 * kg + t) ^ fsw) * 4);
        }
#pragma unroll
        for (int b = 0; b < WN_TILES; ++b) {
          if constexpr (BSPLIT) {      // pre-split activations: same [hi | lo] row image as the weights
            f.bh[b][t] = *(const bf16x8*)(st + (b_row0 + b * 32) * 32 + ((2 * kg + t) ^ fsw) * 4);
            if constexpr (prec_x3(PREC)) f.bl[b][t] = *(const bf16x8*)(st + (b_row0 + b * 32) * 32 + ((4 + 2 * kg + t) ^ fsw) * 4);
          } else {
            f.b32[b][2 * t] = *(const f32x4*)(st + (b_row0 + b * 32) * 32 + ((fcb + 2 * t) ^ fsw) * 4);
            f.b32[b][2 * t + 1] = *(const f32x4*)(st + (b_row0 + b * 32) * 32 + ((fcb + 2 * t + 1) ^ fsw) * 4);
          }
        }
      }
    }
  };
  auto convert = [&](Frag& f) {
    if constexpr (PREC != PREC_F32 && !BSPLIT) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int b = 0; b < WN_TILES; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x0 = f.b32[b][2 * t][e], x1 = f.b32[b][2 * t + 1][e];
            const __bf16 h0 = ta_to16<PREC>(x0), h1 = ta_to16<PREC>(x1);
            f.bh[b][t][e] = h0;
            f.bh[b][t][4 + e] = h1;
            if constexpr (prec_x3(PREC)) {
              f.bl[b][t][e] = ta_to16<PREC>(x0 - ta_from16<PREC>(h0));
              f.bl[b][t][4 + e] = ta_to16<PREC>(x1 - ta_from16<PREC>(h1));
            }
          }
    }
  };
  auto mma = [&](const Frag& f) {
    if constexpr (PREC == PREC_F32) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
            for (int b = 0; b < WN_TILES; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a32[a][g][e], f.b32[b][g][e], acc[a][b], 0, 0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if constexpr (prec_x3(PREC) || prec_x2(PREC)) {
#pragma unroll
          for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
            for (int b = 0; b < WN_TILES; ++b)
              acc[a][b] = ta_mfma16<PREC>(f.al[a][t], f.bh[b][t], acc[a][b]);
        }
        if constexpr (prec_x3(PREC)) {
#pragma unroll
          for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
            for (int b = 0; b < WN_TILES; ++b)
              acc[a][b] = ta_mfma16<PREC>(f.ah[a][t], f.bl[b][t], acc[a][b]);
        }
#pragma unroll
        for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
          for (int b = 0; b < WN_TILES; ++b)
            acc[a][b] = ta_mfma16<PREC>(f.ah[a][t], f.bh[b][t], acc[a][b]);
      }
    }
  };

  const int S = p.n_slabs;
#pragma unroll
  for (int i = 0; i < STAGES - 1; ++i)
    if (i < S) issue(i, i);
  // slab 0 -> fragment set X
  {
    const int ahead = (S - 1) < (STAGES - 2) ? (S - 1) : (STAGES - 2);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (STAGES - 1 < S) issue(STAGES - 1, STAGES - 1);
  }
  Frag X, Y;
  load_raw(X, lds);
  convert(X);
  int nxt_stage = 1;              // LDS stage of slab s+1
  int free_stage = 0;             // stage of slab s: free once every wave has loaded its fragments
  // one step (s + 1 < S): fragments of slab s are in `cur`; bring slab s+1 into `nxt` under the MFMAs of slab s.
  // Everything after the issue is one straight-line block so the scheduler can interleave it.
  auto step = [&](Frag& cur, Frag& nxt, int s) {
    const int rem = S - 2 - s;                         // slabs younger than s+1 that exist
    const int ahead = rem < (STAGES - 2) ? rem : (STAGES - 2);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the fragment reads of slab s (issued in the previous step) must have RETURNED before the barrier hands its stage to the
    // next DMA: with pre-split or float32 operands nothing consumes them before the barrier (convert() is empty), so the
    // compiler's own wait sits in front of their first MFMA -- behind the barrier.  (A read that lost the race returned the
    // next slab's bytes: one run in a few hundred, found when the kernel's register allocation changed.)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // slab s+1 visible; all waves are done reading slab s from LDS
    asm volatile("" ::: "memory");
    if (s + STAGES < S) issue(s + STAGES, free_stage);
    __builtin_amdgcn_sched_barrier(0);
    load_raw(nxt, lds + nxt_stage * STAGE);
    mma(cur);
    convert(nxt);
    if constexpr (PREC != PREC_F32 && !BSPLIT) {
      // hipcc otherwise emits the MFMAs back to back and the hi/lo split after them: pin an interleave
      // (all fragment reads first, then 1 MFMA : VPM VALU) so the split runs in the MFMA shadows.
      constexpr int NREAD = 2 * (WM_TILES * ((prec_x3(PREC) || prec_x2(PREC)) ? 2 : 1) + 2 * WN_TILES);
      constexpr int NMFMA = 2 * WM_TILES * WN_TILES * prec_nmma(PREC);
      constexpr int VPM = (prec_x3(PREC) ? 58 : 30) * WN_TILES / NMFMA + 1;
      __builtin_amdgcn_sched_group_barrier(0x100, NREAD, 0);
#pragma unroll
      for (int i = 0; i < NMFMA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
      }
    }
    free_stage = nxt_stage;
    nxt_stage = nxt_stage + 1 == STAGES ? 0 : nxt_stage + 1;
  };
  int s = 0;
  for (; s + 2 < S; s += 2) {
    step(X, Y, s);
    step(Y, X, s + 1);
  }
  if (s + 1 < S) {          // S - s == 2
    step(X, Y, s);
    mma(Y);
  } else {                  // S - s == 1
    mma(X);
  }

  conv_finish_sym<WAVES_M, WAVES_N, WM_TILES, WN_TILES>(p, acc, lds, ct0, pt0, wm, wn, tid, lane, HoWo);
}


// ---------------------------------------------------------------------------------------------------
// Depthwise 3x3 (+BN+ReLU) fused into the following 1x1 conv (+BN+ReLU): RetinaFace's ConvSepBlock chain
// (retinaface/model.py:26-39, 60-99) regrouped as [dw_k -> pw_{k+1}].  The graph is HBM-bound (35 FLOP/B): the
// depthwise output never leaves the CU.  Same tile machinery as conv_igemm above (weights DMA'd per K slab, fragments,
// epilogue), but the pixel rows of a slab are COMPUTED into LDS -- 9 taps x 16-byte loads per (pixel, 4 channels), fmaf
// chain in (ky, kx) order exactly like dwconv3x3_kernel -- instead of DMA'd.  The 1x1 runs on the exact-f32 MFMA, or
// (PREC_F16X3: pack.dwpw(precision='f16x3')) on the split-half MFMA with the float32 depthwise rows split in registers.
template <int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES, int PREC = PREC_F32>
__global__ __launch_bounds__(256, 2) void conv_dwpw(const ta_conv_launch p) {
  // Measured alternatives (32 x 416 x 739 frames, 12 blocks): this symmetric 4-wave kernel, two workgroups per CU,
  // 541 us; 8 waves with the tap loads of slab s+1 issued ahead of the MFMAs of slab s (208 VGPRs, one workgroup per CU,
  // 306 tiles -> two rounds) 631 us.
  constexpr int BN = WAVES_M * WM_TILES * 32;   // output channels per workgroup
  constexpr int BM = WAVES_N * WN_TILES * 32;   // pixels per workgroup
  constexpr int QA = BN / 32;
  constexpr int RP = BM / 32;                   // pixel rows computed per thread per slab (thread = (row % 32, chunk))
  constexpr int STAGE = (BN + BM) * 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int n_ct = p.coutp / BN;
  const int bid = blockIdx.x;
  const int grp = bid >> 3, xcd = bid & 7;
  const int ct = grp % n_ct;
  const int n_pt = (p.M + BM - 1) / BM;
  const int pt = ta_xcd_tile(n_pt, xcd, grp / n_ct);
  if (pt < 0) return;
  const int ct0 = ct * BN;
  const int pt0 = pt * BM;
  const int HoWo = p.Ho * p.Wo;
  if (wave == 0) TA_STAMP(16);                      // (debug build) entry

  // weight rows: DMA, lane -> (row = t*8 + lane/8, physical chunk lane%8), logical chunk = pchunk ^ ((row>>1)&7)
  const int pchunk = lane & 7;
  const int lchunk = pchunk ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  const char* a_src[QA];
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    const int row = (q * 4 + wave) * 8 + (lane >> 3);
    a_src[q] = (const char*)(p.w + ((size_t)(ct0 + row)) * 32 + lchunk * 4);
  }
  const size_t a_slab_bytes = (size_t)p.coutp * 128;

  // pixel rows: thread -> chunk c4 = tid & 7 (4 channels of the slab), rows (tid >> 3) + 32 j
  const int c4 = tid & 7;
  const float* src[RP];
#pragma unroll
  for (int j = 0; j < RP; ++j) {
    const int row = (tid >> 3) + 32 * j;
    int pix = pt0 + row;
    if (pix >= p.M) pix = 0;                      // clamp: the store is masked
    const int img = pix / HoWo;
    const int rem = pix - img * HoWo;
    const int y = rem / p.Wo;
    const int x = rem - y * p.Wo;
    src[j] = p.in + (size_t)img * p.in_img + (size_t)(y * p.dw_stride) * p.in_row + (size_t)(x * p.dw_stride) * p.in_pix + p.in_off0;
  }

  unsigned dw_amax = 0;                            // largest depthwise value (bit pattern), split-half variant only
  auto produce = [&](int s, int stage) {
    float* base = lds + stage * STAGE;
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int t = q * 4 + wave;
      __builtin_amdgcn_global_load_lds(GLB_PTR(a_src[q] + (size_t)s * a_slab_bytes), LDS_PTR(base + t * 256), 16, 0, 0);
    }
    const int ch = s * 32 + c4 * 4;
    f32x4 w9[9], bias;
    const bool real = ch < p.dw_c;
    if (real) {
#pragma unroll
      for (int t = 0; t < 9; ++t) w9[t] = *(const f32x4*)(p.dw_w + t * p.dw_c + ch);
      bias = *(const f32x4*)(p.dw_bias + ch);
    }
#pragma unroll
    for (int j = 0; j < RP; ++j) {
      const int row = (tid >> 3) + 32 * j;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (real) {
        acc = bias;
        f32x4 v[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
#ifdef TA_CONV_TRACE
            // debug build only (TA_DWPW_PROBE, timing ablations, wrong results): 1 = every tap reads the window's first pixel
            // (one line per pixel instead of nine), 2 = no tap loads at all -- what the block costs beyond its input traffic
            const size_t toff = (p.probe & 3) == 1 ? 0 : (size_t)ky * p.in_row + (size_t)kx * p.in_pix;
            v[ky * 3 + kx] = (p.probe & 3) == 2 ? bias : *(const f32x4*)(src[j] + toff + ch);
#else
            v[ky * 3 + kx] = *(const f32x4*)(src[j] + (size_t)ky * p.in_row + (size_t)kx * p.in_pix + ch);
#endif
          }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(v[t][e], w9[t][e], acc[e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = ta_relu(acc[e]);
        // split-half 1x1: these rows are split into half floats in registers -- they are range-checked like a stored tensor
        if constexpr (PREC != PREC_F32) dw_amax = ta_amax4(dw_amax, acc);
      }
      *(f32x4*)(base + (BN + row) * 32 + ((c4 ^ ((row >> 1) & 7)) * 4)) = acc;
    }
  };

  f32x16 acc[WM_TILES][WN_TILES];
#pragma unroll
  for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
    for (int b = 0; b < WN_TILES; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int frow = lane & 31;
  const int fsw = (frow >> 1) & 7;
  const int fcb = (lane >> 5) * 4;
  const int a_row0 = wm * WM_TILES * 32 + frow;
  const int b_row0 = BN + wn * WN_TILES * 32 + frow;

  const int S = p.n_slabs;
  if (wave == 0) TA_STAMP(17);                      // pixel addresses ready
  produce(0, 0);
  if (wave == 0) TA_STAMP(18);                      // slab 0: taps loaded, depthwise rows written (weight DMA in flight)
  for (int s = 0; s < S; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // slab s (weights landed, pixel rows written); everyone is done reading the other stage
    if (s == 0 && wave == 0) TA_STAMP(19);          // past the first barrier
    if (s + 1 < S) produce(s + 1, (s + 1) & 1);
    const float* st = lds + (s & 1) * STAGE;
    conv_slab_mma<WM_TILES, WN_TILES, PREC>(st, acc, a_row0, b_row0, fsw, fcb, lane);
  }
  if (wave == 0) TA_STAMP(20);                      // MFMAs issued
  if constexpr (PREC != PREC_F32) {
    if (dw_amax > TA_F16_MAX_BITS) *p.range_flag = 1;
    if (p.amax_index >= 0 && dw_amax) atomicMax((unsigned*)p.range_flag + TA_AMAX_SLOT0 + 2 * p.amax_index + 1, dw_amax);
  }
  conv_finish_sym<WAVES_M, WAVES_N, WM_TILES, WN_TILES>(p, acc, lds, ct0, pt0, wm, wn, tid, lane, HoWo);
  if (wave == 0) TA_STAMP(23);                      // drained: stores issued
}

// t / d for a launch-uniform divisor whose float32 reciprocal the launcher supplied: one multiply and a +-1 fix-up
// (exact for 0 <= t < 2^24, which the launcher checks: fast_div)
__device__ __forceinline__ int ta_div_r(int t, int d, float rd, int fast) {
  if (!fast) return t / d;
  int q = (int)((float)t * rd);
  const int r = t - q * d;
  if (r < 0) --q;
  else if (r >= d) ++q;
  return q;
}

// (img, y, x) of the pixels pt0 + d of a tile, without a full integer division per lane: the tile's first pixel is
// decomposed once (wave-uniform), every other pixel is d < 65536 further in raster order, so its carries are small
// quotients that an f32 multiply by the reciprocal gets right to +-1 (fixed up exactly).
struct ta_pixel_walk {
  int img0, y0, x0, Wo, Ho;
  float rWo, rHo;
  __device__ __forceinline__ ta_pixel_walk(const ta_conv_launch& p, int pt0, int HoWo) {
    Wo = p.Wo;
    Ho = p.Ho;
    if (p.fast_div) {                            // split-role launches: reciprocals from the launcher
      img0 = ta_div_r(pt0, HoWo, p.r_HoWo, 1);
      const int rem = pt0 - img0 * HoWo;
      y0 = ta_div_r(rem, Wo, p.r_Wo, 1);
      x0 = rem - y0 * Wo;
      rWo = p.r_Wo;
      rHo = p.r_Ho;
      return;
    }
    img0 = pt0 / HoWo;
    const int rem = pt0 - img0 * HoWo;
    y0 = rem / p.Wo;
    x0 = rem - y0 * p.Wo;
    rWo = 1.0f / (float)p.Wo;
    rHo = 1.0f / (float)p.Ho;
  }
  static __device__ __forceinline__ void divmod(int t, int d, float rd, int& q, int& r) {
    q = (int)((float)t * rd);
    r = t - q * d;
    if (r < 0) {
      --q;
      r += d;
    } else if (r >= d) {
      ++q;
      r -= d;
    }
  }
  __device__ __forceinline__ void at(int d, int& img, int& y, int& x) const {
    int qy, qi;
    divmod(x0 + d, Wo, rWo, qy, x);
    divmod(y0 + qy, Ho, rHo, qi, y);
    img = img0 + qi;
  }
};

// Epilogue of the split-role kernel, staged through LDS.  Straight from the accumulators a store instruction
// scatters 8-16 B to 32 different pixels (32 cache lines per instruction, 4-8x write amplification: measured 6.2 us
// per tile, most of a launch's fixed cost).  Here the consumers first park the raw 128 x 128 (64 x 256) tile in LDS
// as [pixel][cout] (16-byte chunks XOR-swizzled with the pixel row so the column-wise writes are conflict-free),
// then every lane takes 8 consecutive channels of one pixel -- a wave instruction covers whole 128-byte lines --
// and applies bias / ReLU / PReLU / residual / second affine output on the way out.
struct ta_f32x8 {
  f32x4 a, b;             // channels ch..ch+3, ch+4..ch+7
};
__device__ __forceinline__ ta_f32x8 ta_ld8(const float* pix, int ch, int fmt) {      // ch % 8 == 0
  ta_f32x8 r;
  if (fmt == TA_FMT_F32) {
    r.a = *(const f32x4*)(pix + ch);
    r.b = *(const f32x4*)(pix + ch + 4);
    return r;
  }
  if (fmt == TA_FMT_F16) {
    const uint4 w = *(const uint4*)((const char*)pix + 2 * ch);
    float v[8];
    ta_unpack2<true>(w.x, v[0], v[1]);
    ta_unpack2<true>(w.y, v[2], v[3]);
    ta_unpack2<true>(w.z, v[4], v[5]);
    ta_unpack2<true>(w.w, v[6], v[7]);
    r.a = f32x4{v[0], v[1], v[2], v[3]};
    r.b = f32x4{v[4], v[5], v[6], v[7]};
    return r;
  }
  const char* q = (const char*)pix + ta_split_chan(ch);
  const uint4 h = *(const uint4*)q, l = *(const uint4*)(q + 64);
  const unsigned hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
  float v[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float h0, h1, l0, l1;
    if (fmt == TA_FMT_SPLIT16) {
      ta_unpack2<true>(hw[i], h0, h1);
      ta_unpack2<true>(lw[i], l0, l1);
    } else {
      ta_unpack2<false>(hw[i], h0, h1);
      ta_unpack2<false>(lw[i], l0, l1);
    }
    v[2 * i] = h0 + l0;
    v[2 * i + 1] = h1 + l1;
  }
  r.a = f32x4{v[0], v[1], v[2], v[3]};
  r.b = f32x4{v[4], v[5], v[6], v[7]};
  return r;
}
__device__ __forceinline__ void ta_st8(float* pix, int ch, int fmt, const ta_f32x8& v) {   // ch % 8 == 0
  if (fmt == TA_FMT_F32) {
    *(f32x4*)(pix + ch) = v.a;
    *(f32x4*)(pix + ch + 4) = v.b;
    return;
  }
  if (fmt == TA_FMT_F16) {
    *(uint4*)((char*)pix + 2 * ch) = make_uint4(ta_pack_half2(v.a[0], v.a[1]), ta_pack_half2(v.a[2], v.a[3]),
                                                 ta_pack_half2(v.b[0], v.b[1]), ta_pack_half2(v.b[2], v.b[3]));
    return;
  }
  const float x[8] = {v.a[0], v.a[1], v.a[2], v.a[3], v.b[0], v.b[1], v.b[2], v.b[3]};
  unsigned hw[4], lw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (fmt == TA_FMT_SPLIT16) ta_pack2<true>(x[2 * i], x[2 * i + 1], hw[i], lw[i]);
    else ta_pack2<false>(x[2 * i], x[2 * i + 1], hw[i], lw[i]);
  }
  char* q = (char*)pix + ta_split_chan(ch);
  *(uint4*)q = make_uint4(hw[0], hw[1], hw[2], hw[3]);
  *(uint4*)(q + 64) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
}

// largest |x| (bit pattern) among the n4 (1 or 2) stored 4-channel halves of v
__device__ __forceinline__ unsigned ta_absmax8(unsigned m, const ta_f32x8& v, int n4) {
  m = ta_amax4(m, v.a);
  if (n4 == 2) m = ta_amax4(m, v.b);
  return m;
}

template <int BN>
__device__ __forceinline__ void conv_epilogue_park(f32x16 (&acc)[2][2], float* lds, int cm, int cn, int lane) {
  constexpr int NCH = BN / 4;                      // 16-byte chunks per pixel row of the staged tile
  // ---- phase 1: accumulators -> LDS [pixel][cout]; acc[a][b][r]: pixel = lane & 31, cout = 8 (r >> 2) + 4 (lane >> 5) + (r & 3)
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int row = cn * 64 + b * 32 + (lane & 31);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = (cm * 64 + a * 32 + 8 * j + 4 * (lane >> 5)) >> 2;
        *(f32x4*)(lds + (row * NCH + (c ^ (row & (NCH - 1)))) * 4) =
            f32x4{acc[a][b][4 * j], acc[a][b][4 * j + 1], acc[a][b][4 * j + 2], acc[a][b][4 * j + 3]};
      }
  }
}

// ---- phase 2 (all NT threads of the workgroup, producers included): lane = (pixel row, 8 consecutive channels)
template <int BN, int BM, int NT>
__device__ __forceinline__ void conv_epilogue_drain(const ta_conv_launch& p, const float* lds, int ct0, int pt0, int tid,
                                                    int HoWo, int ks) {
  constexpr int NCH = BN / 4;
  constexpr int G = BN / 8;                        // 8-channel groups per pixel
  constexpr int RPI = NT / G;                      // pixel rows per pass of the workgroup
  const int k8 = tid % G, r0 = tid / G;
  const int co = ct0 + 8 * k8;
  unsigned amax = 0;                               // largest |x| stored, as a bit pattern (ta_range_report)
  const bool chk = p.range_check, chk2 = chk && p.out2;
  if (p.k_split > 1) {                             // K-split: raw sums of this K range -> partial[ks][pixel][coutp]
    float* dst = p.partial + (size_t)ks * p.M * p.coutp + co;
    for (int row = r0; row < BM && pt0 + row < p.M; row += RPI) {
      const int sw = row & (NCH - 1);
      float* o = dst + (size_t)(pt0 + row) * p.coutp;
      *(f32x4*)o = *(const f32x4*)(lds + (row * NCH + ((2 * k8) ^ sw)) * 4);
      *(f32x4*)(o + 4) = *(const f32x4*)(lds + (row * NCH + ((2 * k8 + 1) ^ sw)) * 4);
    }
    return;
  }
  const int n4 = p.cout - co >= 8 ? 2 : (p.cout - co >= 4 ? 1 : 0);    // valid 4-channel halves (cout % 4 == 0)
  if (n4 == 0) return;
  const f32x4 bias0 = *(const f32x4*)(p.bias + co), bias1 = *(const f32x4*)(p.bias + co + 4);   // padded to coutp
  const f32x4 us0 = *(const f32x4*)(p.bias + p.coutp + co), us1 = *(const f32x4*)(p.bias + p.coutp + co + 4);   // per-channel power of two (ta_op_desc.wus_off)
  f32x4 sl0 = {0, 0, 0, 0}, sl1 = {0, 0, 0, 0}, sc0 = sl0, sc1 = sl0, sh0 = sl0, sh1 = sl0;
  if (p.act == TA_ACT_PRELU) {
    sl0 = *(const f32x4*)(p.prelu + co);
    sl1 = *(const f32x4*)(p.prelu + co + 4);
  }
  if (p.out2) {
    sc0 = *(const f32x4*)(p.scale2 + co);
    sc1 = *(const f32x4*)(p.scale2 + co + 4);
    sh0 = *(const f32x4*)(p.shift2 + co);
    sh1 = *(const f32x4*)(p.shift2 + co + 4);
  }
  if (p.pool) {
    // 2x2 max-pool in the epilogue: rows 4 w .. 4 w + 3 of the staged tile are the four pixels of window w and sit in lanes
    // G and 2 G apart (same 8 channels): two shuffle-max steps, then the window's first lane stores the pooled pixel.
    // (bias and ReLU are applied first -- the values the separate pool kernel would have read.)
    static_assert(4 * G <= 64 && (RPI & 3) == 0, "a window's four rows live in one wave");
    const ta_pixel_walk walk(p, pt0 >> 2, HoWo);
    for (int row = r0; row < BM; row += RPI) {
      if (pt0 + row >= p.M) break;                   // M is a multiple of 4: whole windows drop out together
      const int sw = row & (NCH - 1);
      ta_f32x8 v;
      v.a = *(const f32x4*)(lds + (row * NCH + ((2 * k8) ^ sw)) * 4);
      v.b = *(const f32x4*)(lds + (row * NCH + ((2 * k8 + 1) ^ sw)) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v.a[e] = __builtin_fmaf(v.a[e], us0[e], bias0[e]);     // us == 1 outside the half-float programs: v + bias
        v.b[e] = __builtin_fmaf(v.b[e], us1[e], bias1[e]);
        if (p.act == TA_ACT_RELU) {
          v.a[e] = ta_relu(v.a[e]);
          v.b[e] = ta_relu(v.b[e]);
        }
      }
#pragma unroll
      for (int m = G; m <= 2 * G; m <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v.a[e] = fmaxf(v.a[e], __shfl_xor(v.a[e], m));
          v.b[e] = fmaxf(v.b[e], __shfl_xor(v.b[e], m));
        }
      if ((row & 3) == 0) {
        int img, qy, qx;
        walk.at(row >> 2, img, qy, qx);
        float* o = p.out + (size_t)img * p.out_img + (size_t)qy * p.out_row + (size_t)qx * p.out_pix + p.out_off0;
        if (n4 == 2) ta_st8(o, p.out_ch + co, p.out_fmt, v);
        else ta_st4(o, p.out_ch + co, p.out_fmt, v.a);
        if (chk) amax = ta_absmax8(amax, v, n4);
      }
    }
    ta_range_report(p, amax);
    return;
  }
  int pix = pt0 + r0;
  int img, y, x;
  ta_pixel_walk(p, pt0, HoWo).at(r0, img, y, x);
#pragma unroll 2
  for (int row = r0; row < BM; row += RPI, pix += RPI) {
    if (pix >= p.M) break;
    const int sw = row & (NCH - 1);
    ta_f32x8 v;
    v.a = *(const f32x4*)(lds + (row * NCH + ((2 * k8) ^ sw)) * 4);
    v.b = *(const f32x4*)(lds + (row * NCH + ((2 * k8 + 1) ^ sw)) * 4);
    f32x4 bb0 = bias0, bb1 = bias1;
    if (p.bias9) {
      const int cls = ta_border_class(y, x, p.Ho, p.Wo);
      if (cls != TA_INTERIOR) {
        bb0 = *(const f32x4*)(p.bias9 + (size_t)cls * p.coutp + co);
        bb1 = *(const f32x4*)(p.bias9 + (size_t)cls * p.coutp + co + 4);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v.a[e] = __builtin_fmaf(v.a[e], us0[e], bb0[e]);
      v.b[e] = __builtin_fmaf(v.b[e], us1[e], bb1[e]);
    }
    if (p.act == TA_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v.a[e] = ta_relu(v.a[e]);
        v.b[e] = ta_relu(v.b[e]);
      }
    } else if (p.act == TA_ACT_PRELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v.a[e] = v.a[e] > 0.f ? v.a[e] : v.a[e] * sl0[e];
        v.b[e] = v.b[e] > 0.f ? v.b[e] : v.b[e] * sl1[e];
      }
    }
    if (p.res) {
      const int ry = p.res_up2 ? (y >> 1) : y, rx = p.res_up2 ? (x >> 1) : x;
      const float* rs = p.res + (size_t)img * p.res_img + (size_t)ry * p.res_row + (size_t)rx * p.res_pix + p.res_off0;
      if (n4 == 2) {
        const ta_f32x8 r = ta_ld8(rs, p.res_ch + co, p.res_fmt);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v.a[e] += r.a[e];          // (a shortcut carries the exponents of the sum it joins: pack.Program.tensor_scales)
          v.b[e] += r.b[e];
        }
      } else {
        const f32x4 r = ta_ld4(rs, p.res_ch + co, p.res_fmt);
#pragma unroll
        for (int e = 0; e < 4; ++e) v.a[e] += r[e];
      }
    }
    float* o = p.out + (size_t)img * p.out_img + (size_t)y * p.out_row + (size_t)x * p.out_pix + p.out_off0;
    if (n4 == 2) ta_st8(o, p.out_ch + co, p.out_fmt, v);
    else ta_st4(o, p.out_ch + co, p.out_fmt, v.a);
    if (chk) amax = ta_absmax8(amax, v, n4);
    if (p.out2) {
      float* o2 = p.out2 + (size_t)img * p.o2_img + (size_t)y * p.o2_row + (size_t)x * p.o2_pix + p.o2_off0;
      ta_f32x8 z;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        z.a[e] = v.a[e] * sc0[e] + sh0[e];
        z.b[e] = v.b[e] * sc1[e] + sh1[e];
      }
      if (n4 == 2) ta_st8(o2, p.o2_ch + co, p.o2_fmt, z);
      else ta_st4(o2, p.o2_ch + co, p.o2_fmt, z.a);
      if (chk2) amax = ta_absmax8(amax, z, n4);
    }
    x += RPI;                                        // next pass: RPI pixels further in raster order
    while (x >= p.Wo) {
      x -= p.Wo;
      if (++y == p.Ho) {
        y = 0;
        ++img;
      }
    }
  }
  ta_range_report(p, amax);
}

// ---- the same phase 2, specialised at compile time for the three epilogues that carry the bf16 workloads (launcher
// flag fast_drain: split-format tensors addressed with 32-bit byte offsets, every lane's 8 channels inside cout, no
// pool, no K-split).  The generic drain above spends ~19 lane-instructions per output element on run-time flags and
// 64-bit addressing and is VALU-issue-bound (tools/conv_trace.py); this one is ~2x leaner.  Same arithmetic, same
// order, same bits.
// F16 (the format kind): 0 = TA_FMT_SPLIT (bf16 words), 1 = TA_FMT_SPLIT16 (half words), 2 = TA_FMT_F16 (plain half floats, one
// 16-byte chunk per 8 channels); for 1 and 2 `amax` collects the largest |x| stored (range flag)
template <int F16>
__device__ __forceinline__ void ta_split_store8(char* q, const float (&x)[8], unsigned& amax) {
  if constexpr (F16 == 2) {
    *(uint4*)q = make_uint4(ta_pack_half2(x[0], x[1]), ta_pack_half2(x[2], x[3]), ta_pack_half2(x[4], x[5]), ta_pack_half2(x[6], x[7]));
  } else {
    unsigned hw[4], lw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ta_pack2<F16 == 1>(x[2 * i], x[2 * i + 1], hw[i], lw[i]);
    *(uint4*)q = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    *(uint4*)(q + 64) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
  }
  if constexpr (F16 != 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) amax = max(amax, max(ta_absbits(x[2 * i]), ta_absbits(x[2 * i + 1])));   // bit patterns: a NaN cannot hide
  }
}
template <int BN, int BM, int NT, int ACT, bool RES, int F16, bool POOL = false, bool OUT2 = RES, bool B9 = false>
// RES: + shortcut; OUT2: the second (affine) output; POOL: fused 2x2 max-pool; B9: border-class bias (ta_border_class)
__device__ __forceinline__ void conv_drain_fast(const ta_conv_launch& p, const float* lds, int ct0, int pt0, int tid, int HoWo) {
  constexpr int NCH = BN / 4;
  constexpr int G = BN / 8;
  constexpr int RPI = NT / G;
  const int k8 = tid % G, r0 = tid / G;
  const int co = ct0 + 8 * k8;
  if (co >= p.cout) return;                        // cout % 8 == 0: a lane is inside or outside with all 8 channels
  float bias[8], sl[8], sc[8], sh[8], us[8];
  *(f32x4*)bias = *(const f32x4*)(p.bias + co);
  *(f32x4*)(bias + 4) = *(const f32x4*)(p.bias + co + 4);
  *(f32x4*)us = *(const f32x4*)(p.bias + p.coutp + co);             // per-channel power of two, stored behind the bias (ta_op_desc.wus_off)
  *(f32x4*)(us + 4) = *(const f32x4*)(p.bias + p.coutp + co + 4);
  if (ACT == TA_ACT_PRELU) {
    *(f32x4*)sl = *(const f32x4*)(p.prelu + co);
    *(f32x4*)(sl + 4) = *(const f32x4*)(p.prelu + co + 4);
  }
  if (OUT2) {
    *(f32x4*)sc = *(const f32x4*)(p.scale2 + co);
    *(f32x4*)(sc + 4) = *(const f32x4*)(p.scale2 + co + 4);
    *(f32x4*)sh = *(const f32x4*)(p.shift2 + co);
    *(f32x4*)(sh + 4) = *(const f32x4*)(p.shift2 + co + 4);
  }
  auto chan = [](int ch) { return F16 == 2 ? (unsigned)(2 * ch) : ta_split_chan(ch); };
  unsigned amax = 0;
  char* const ob = (char*)p.out + chan(p.out_ch + co);
  const char* const rb = RES ? (const char*)p.res + chan(p.res_ch + co) : nullptr;
  char* const o2b = OUT2 ? (char*)p.out2 + chan(p.o2_ch + co) : nullptr;
  // POOL: rows 4 w .. 4 w + 3 of the staged tile are the pixels of window w (lanes G and 2 G apart hold the same 8
  // channels of a window's other rows); coordinates below are those of the POOLED map and a pass advances STEP of its pixels
  static_assert(!POOL || (4 * G <= 64 && (RPI & 3) == 0), "a window's four rows live in one wave");
  constexpr int STEP = POOL ? RPI / 4 : RPI;
  int img, y, x;
  ta_pixel_walk(p, POOL ? pt0 >> 2 : pt0, HoWo).at(POOL ? r0 >> 2 : r0, img, y, x);
  int pix = pt0 + r0;
  // a pass is STEP pixels further in raster order: (dy rows, dx columns) with at most one carry each when dy < Ho
  const int dy = ta_div_r(STEP, p.Wo, p.r_Wo, 1), dx = STEP - dy * p.Wo;
  const bool one_carry = dy < p.Ho;
#pragma unroll 2
  for (int row = r0; row < BM; row += RPI, pix += RPI) {
    if (pix >= p.M) break;
    const int sw = row & (NCH - 1);
    float v[8];
    *(f32x4*)v = *(const f32x4*)(lds + (row * NCH + ((2 * k8) ^ sw)) * 4);
    *(f32x4*)(v + 4) = *(const f32x4*)(lds + (row * NCH + ((2 * k8 + 1) ^ sw)) * 4);
    unsigned rh[4], rl[4];
    if (RES) {
      const int ry = p.res_up2 ? (y >> 1) : y, rx = p.res_up2 ? (x >> 1) : x;
      const char* rs = rb + 4u * (unsigned)(img * p.res_img + ry * p.res_row + rx * p.res_pix + p.res_off0);
      *(uint4*)rh = *(const uint4*)rs;
      if (F16 != 2) *(uint4*)rl = *(const uint4*)(rs + 64);
    }
    float bb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bb[e] = bias[e];
    if (B9) {
      const int cls = ta_border_class(y, x, p.Ho, p.Wo);
      if (cls != TA_INTERIOR) {
        *(f32x4*)bb = *(const f32x4*)(p.bias9 + cls * p.coutp + co);
        *(f32x4*)(bb + 4) = *(const f32x4*)(p.bias9 + cls * p.coutp + co + 4);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = __builtin_fmaf(v[e], us[e], bb[e]);       // us = 2^(a_out[co] - s[co]); all ones in the bf16 programs: v + bb
      if (ACT == TA_ACT_RELU) v[e] = ta_relu(v[e]);
      if (ACT == TA_ACT_PRELU) v[e] = v[e] > 0.f ? v[e] : v[e] * sl[e];
    }
    if (RES) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float h0, h1, l0 = 0.f, l1 = 0.f;
        ta_unpack2<F16 != 0>(rh[i], h0, h1);
        if (F16 != 2) ta_unpack2<F16 != 0>(rl[i], l0, l1);
        v[2 * i] += h0 + l0;
        v[2 * i + 1] += h1 + l1;
      }
    }
    if (POOL) {
#pragma unroll
      for (int m = G; m <= 2 * G; m <<= 1)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], __shfl_xor(v[e], m));
    }
    if (!POOL || (row & 3) == 0)
      ta_split_store8<F16>(ob + 4u * (unsigned)(img * p.out_img + y * p.out_row + x * p.out_pix + p.out_off0), v, amax);
    if (OUT2) {
      float z[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = v[e] * sc[e] + sh[e];
      ta_split_store8<F16>(o2b + 4u * (unsigned)(img * p.o2_img + y * p.o2_row + x * p.o2_pix + p.o2_off0), z, amax);
    }
    if (one_carry) {                                 // branch-free
      x += dx;
      const int cx = x >= p.Wo ? 1 : 0;
      x -= cx ? p.Wo : 0;
      y += dy + cx;
      const int cy = y >= p.Ho ? 1 : 0;
      y -= cy ? p.Ho : 0;
      img += cy;
    } else {
      x += STEP;
      while (x >= p.Wo) {
        x -= p.Wo;
        if (++y == p.Ho) {
          y = 0;
          ++img;
        }
      }
    }
  }
  if constexpr (F16 != 0) ta_range_report(p, amax);
}
// picks the lean drain when the launch qualifies; false = run the generic one
template <int BN, int BM, int NT, int F16>
__device__ __forceinline__ bool conv_drain_dispatch(const ta_conv_launch& p, const float* lds, int ct0, int pt0, int tid, int HoWo) {
  if (!p.fast_drain) return false;
  if (p.pool) {
    if constexpr (4 * (BN / 8) <= 64 && ((NT / (BN / 8)) & 3) == 0) {
      if (p.act == TA_ACT_RELU && !p.res && !p.out2) {
        conv_drain_fast<BN, BM, NT, TA_ACT_RELU, false, F16, true>(p, lds, ct0, pt0, tid, HoWo);
        return true;
      }
    }
    return false;
  }
  if (p.bias9) {                                      // ArcFace unit-opening convs: folded input BatchNorm, PReLU
    if (p.act == TA_ACT_PRELU && !p.res && !p.out2) {
      conv_drain_fast<BN, BM, NT, TA_ACT_PRELU, false, F16, false, false, true>(p, lds, ct0, pt0, tid, HoWo);
      return true;
    }
    return false;
  }
  if (!p.res && !p.out2) {
    if (p.act == TA_ACT_RELU) conv_drain_fast<BN, BM, NT, TA_ACT_RELU, false, F16>(p, lds, ct0, pt0, tid, HoWo);
    else if (p.act == TA_ACT_PRELU) conv_drain_fast<BN, BM, NT, TA_ACT_PRELU, false, F16>(p, lds, ct0, pt0, tid, HoWo);
    else conv_drain_fast<BN, BM, NT, TA_ACT_NONE, false, F16>(p, lds, ct0, pt0, tid, HoWo);
    return true;
  }
  if (p.res && p.act == TA_ACT_NONE) {                // unit-closing convs: + shortcut, with or without the second output
    if (p.out2) conv_drain_fast<BN, BM, NT, TA_ACT_NONE, true, F16, false, true>(p, lds, ct0, pt0, tid, HoWo);
    else conv_drain_fast<BN, BM, NT, TA_ACT_NONE, true, F16, false, false>(p, lds, ct0, pt0, tid, HoWo);
    return true;
  }
  return false;
}

// ---- split-role kernel (f32 mode, or bf16 modes on pre-split activations; 128 x 128 or 64 x 256 tiles) --
// Measured with tools/probe/*: the global -> LDS DMA path sustains at most ~34 B/clk/CU however many slabs are in
// flight (24 with only 4 issuing waves), and MFMA issue is NOT slowed by DMA waves on the same SIMD -- but a wave
// that has to issue its own DMA stalls in front of the saturated texture addresser with its MFMAs queued behind.
// So the roles are split: waves 0..3 (one per SIMD) are CONSUMERS, each owning a 64 x 64 register tile (4 MFMA
// tiles, 24 MFMAs per slab in bf16x3) and doing nothing but ds_read + MFMA; waves 4..4+NP-1 are PRODUCERS that walk
// K and issue the LDS-DMA for the whole 128 x 128 workgroup tile (32 KiB per slab -> 1.5x the FLOPs per DMA byte of
// the 64 x 128 kernel above).  One s_barrier per slab hands a landed slab to the consumers and a drained stage back
// to the producers.  The consumer loop is software-pipelined at k-step (16) granularity with the barrier in the
// middle, so both fragment reads of a slab hide under 12 MFMAs each and only two 8-fragment sets are live.
// STAGES == 2 (4 consumer waves only): the "two tiles per CU" variant for SHORT-K layers.  A 2-stage ring of a 128 x 128 or 64 x 256 tile
// is <= 80 KiB, so two workgroups share a CU: one's fixed cost (kernel entry, address set-up, first DMA latency, park, drain: 10 - 12 k
// cycles against the 14 - 28 k of an 18 / 36-slab loop) runs under the other's K loop.  Four waves per SIMD leave 128 VGPRs per lane: the
// consumer keeps ONE fragment set (reads of a k-step, then its MFMAs) -- the LDS latency that the three-stage kernel hides inside a wave
// is hidden by the other workgroup's consumer on the same SIMD.  Same tiles, K order and MFMA order: same bits.
template <int CM, int CN, int NP, int PREC, int STAGES>
__global__ __launch_bounds__(64 * (CM * CN + NP), STAGES == 2 ? 4 : (CM * CN + NP) / 4) void conv_igemm_split(const ta_conv_launch p) {
  static_assert(NP == 4 || NP == 8, "4 or 8 producer waves");
  static_assert(STAGES == 3 || (STAGES == 2 && CM * CN == 4 && NP == 4), "3-stage ring, or the 2-stage two-workgroups-per-CU variant of the 8-wave tiles");
  static_assert(CM * CN == 4 || CM * CN == 8, "consumer grid: 1x4 (64 cout x 256 px), 2x2 (128 x 128) or 2x4 (128 x 256)");
  constexpr int NC = CM * CN;                    // consumer waves (the first NC waves of the workgroup)
  constexpr int BN = CM * 64, BM = CN * 64;
  constexpr int NI = (BN + BM) / 8 / NP;         // DMA instructions per producer wave per slab
  constexpr int QA = BN / 8 / NP;                // ... of which weight rows
  constexpr int STAGE = (BN + BM) * 32;          // floats per stage
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave == 0) TA_STAMP(0);                       // kernel entry (consumer 0)
  if (wave == NC) TA_STAMP(8);                      // kernel entry (producer 0)

  // set-up without integer divisions (launch-uniform divisors come with their reciprocals; K ranges only when K-split)
  const int n_ct = p.coutp / BN;                                 // BN, BM: powers of two
  const int n_pt = (p.M + BM - 1) / BM;
  const int tile_blocks = (((n_pt + 7) >> 3) * n_ct) << 3;      // blocks per K range
  const int ks = p.k_split > 1 ? ta_div_r(blockIdx.x, tile_blocks, p.r_tile_blocks, p.fast_div) : 0;   // K range of this workgroup
  const int bid = blockIdx.x - ks * tile_blocks;
  const int grp = bid >> 3, xcd = bid & 7;
  const int gq = ta_div_r(grp, n_ct, p.r_nct, p.fast_div);
  const int pt = ta_xcd_tile(n_pt, xcd, gq);
  if (pt < 0) return;
  const int ct0 = (grp - gq * n_ct) * BN;
  const int pt0 = pt * BM;
  const int HoWo = p.Ho * p.Wo;
  int s_begin = 0, S = p.n_slabs;
  if (p.k_split > 1) {
    s_begin = (int)(((long long)ks * p.n_slabs) / p.k_split);
    S = (int)(((long long)(ks + 1) * p.n_slabs) / p.k_split) - s_begin;
  }
  // The epilogue is ALWAYS the LDS-staged, line-coalesced one: the launcher gives this kernel only layers whose channel
  // slices are 8-aligned (variant_eligible).  The direct epilogue (stores straight from the accumulators) used to be
  // compiled in as a fallback no layer of the three networks ever took -- and set the register budget of the whole kernel:
  // 168 VGPRs + 121 spilled in the 12-wave variants, 241 in the 8-wave ones, against 143 and no spill without it.

  if (wave >= NC) {
    // ================= producer =================
    const int pw = wave - NC;
    const int pchunk = lane & 7;
    const int lchunk = pchunk ^ ((4 * (pw & 1) + (lane >> 4)) & 7);
    // uniform 64-bit base (SGPRs) + per-lane 32-bit byte offset (one VGPR): the saddr form of global_load_lds
    const char* a_base = (const char*)p.w;
    const size_t a_slab_bytes = (size_t)p.coutp * 128;
    unsigned a_off[QA];
    unsigned b_off[NI - QA];
#pragma unroll
    for (int q = 0; q < QA; ++q) a_off[q] = (unsigned)(((ct0 + (q * NP + pw) * 8 + (lane >> 3)) * 32 + lchunk * 4) * 4);
    ta_k_walk wa(p, s_begin);                       // weight rows and pixel rows are issued in the same slab order, the
    auto issue_a = [&](int, int stage) {            // weight rows two slabs ahead at the start: two walkers
#pragma unroll
      for (int q = 0; q < QA; ++q) ta_dma16(a_base + (size_t)wa.a_slab * a_slab_bytes, a_off[q], lds + stage * STAGE + (q * NP + pw) * 256);
      wa.advance();
    };
    // the weight rows of the first two slabs need no pixel arithmetic: get them moving first
    issue_a(0, 0);
    if (S > 1) issue_a(1, 1);
    // pixel rows: offsets relative to the tile's first pixel (pixels of a tile ascend in raster order)
    // fused max-pool: the walk runs over 2x2 windows (pooled map), pixel d of the tile is position (d >> 1 & 1, d & 1) of window d >> 2
    const int pl = p.pool;
    const ta_pixel_walk walk(p, pl ? pt0 >> 2 : pt0, HoWo);
    const int in_ch = p.in_ch_off + (p.group_cout ? (ct0 / p.group_cout) * p.group_cin : 0);   // grouped conv: this tile's group
    const size_t off0 = (size_t)walk.img0 * p.in_img + (size_t)((walk.y0 << pl) * p.stride) * p.in_row +
                        (size_t)((walk.x0 << pl) * p.stride) * p.in_pix + p.in_off0 + in_ch;
    const char* b_base = (const char*)(p.in + off0);
    ta_k_walk wb(p, s_begin);
    // every pixel row's DMA of slab 0 goes out as soon as its address is known: the index arithmetic of the later rows
    // (two reciprocal divisions each) then runs under the flight time of the earlier ones instead of in front of them all
#pragma unroll
    for (int q = QA; q < NI; ++q) {
      const int d = (q * NP + pw) * 8 + (lane >> 3) - BN;   // pixel row of the stage image
      int img, y, x;
      const int dd = pt0 + d < p.M ? d : 0;
      walk.at(pl ? dd >> 2 : dd, img, y, x);
      if (pl) {
        y = 2 * y + ((dd >> 1) & 1);
        x = 2 * x + (dd & 1);
      }
      const size_t off = (size_t)img * p.in_img + (size_t)(y * p.stride) * p.in_row + (size_t)(x * p.stride) * p.in_pix +
                         p.in_off0 + in_ch;
      b_off[q - QA] = (unsigned)((off - off0) * 4 + lchunk * 16);
      if (!(p.probe & 4)) ta_dma16(b_base + wb.b_off, b_off[q - QA], lds + (q * NP + pw) * 256);      // slab 0 -> stage 0
    }
    if (!(p.probe & 4)) wb.advance();
    auto issue_b = [&](int stage) {                 // pixel rows of the next slab in K order
#pragma unroll
      for (int q = QA; q < NI; ++q) ta_dma16(b_base + wb.b_off, b_off[q - QA], lds + stage * STAGE + (q * NP + pw) * 256);
      wb.advance();
    };
    if (wave == NC) TA_STAMP(9);                    // producer: addresses ready, slab 0 issued
    if ((p.probe & 4)) issue_b(0);                        // tools (TA_CONV_LATE_B): the round-2 order, all addresses first
    if (S > 1) issue_b(1);
    if (wave == NC) TA_STAMP(10);                    // producer: first slabs issued
    int stage = STAGES == 2 ? 0 : 2;                // stage the next issued slab goes to
    if constexpr (STAGES == 2) {
      // two stages: slabs 0 and 1 are in flight; slab s + 1 (s >= 1) goes out once B_s has handed back the stage of slab s - 1
      for (int s = 0; s < S; ++s) {
        if (s == 0 && S > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI - QA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // B_s
        asm volatile("" ::: "memory");
        if (s >= 1 && s + 1 < S) {
          issue_a(s + 1, stage);
          issue_b(stage);
          stage ^= 1;
        }
      }
    } else
    for (int s = 0; s < S; ++s) {
      // slab s must have landed; issue order was [A0 A1 B0 B1] then [A B] per slab, and vmcnt counts in issue order
      if ((p.probe & 3) && s > 0) {                        // tools only (timing ablation): fewer DMAs in flight
        if ((p.probe & 3) == 1 && s + 1 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (s == 0 && S > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI - QA) : "memory");
      else if (s + 1 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                 // B_s: slab s landed (all producers); consumers have drained slab s-1
      asm volatile("" ::: "memory");
      if (s + 2 < S) {
        if ((p.probe & 3) != 2) issue_a(s + 2, stage);
        if (!(p.probe & 3)) issue_b(stage);
        stage = stage == 2 ? 0 : stage + 1;
      }
    }
    {                                               // help drain the parked tile: twice the lanes for the epilogue math
      __builtin_amdgcn_s_barrier();                 // E0
      __builtin_amdgcn_s_barrier();                 // E1
      asm volatile("" ::: "memory");
      bool done = false;
      if constexpr (PREC != PREC_F32) done = conv_drain_dispatch<BN, BM, 64 * (NC + NP), (PREC == PREC_F16 ? 2 : (prec_half(PREC) ? 1 : 0))>(p, lds, ct0, pt0, tid, HoWo);
      if (!done) conv_epilogue_drain<BN, BM, 64 * (NC + NP)>(p, lds, ct0, pt0, tid, HoWo, ks);
    }
    return;
  }

  // ================= consumer =================
  // the MFMA waves ahead of the DMA-issuing producers in the CU's issue arbitration (TA_CONV_PRIO, A/B; s_setprio ignores EXEC)
  if (p.cons_prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (p.cons_prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (p.cons_prio == 3) __builtin_amdgcn_s_setprio(3);
  const int cm = wave / CN, cn = wave % CN;
  int stage = 0;                                    // stage of the slab being consumed
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int frow = lane & 31;
  const int fsw = (frow >> 1) & 7;
  const int kg = lane >> 5;
  const int a_row0 = cm * 64 + frow;
  const int b_row0 = BN + cn * 64 + frow;
  struct Frag {                                     // one k-step (16) of a slab
    bf16x8 ah[2], al[2], bh[2], bl[2];              // bf16 modes: operands pre-split in LDS
    f32x4 a32[2][2], b32[2][2];                     // f32 mode: lane (row, kg) holds k = 16 kg + 8 t + 0..7
  };
  const int fcb = kg * 4;
  auto load = [&](Frag& f, const float* st, int t) {
    if constexpr (PREC == PREC_F32) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int pc = ((fcb + 2 * t + g) ^ fsw) * 4;
#pragma unroll
        for (int a = 0; a < 2; ++a) f.a32[a][g] = *(const f32x4*)(st + (a_row0 + a * 32) * 32 + pc);
#pragma unroll
        for (int b = 0; b < 2; ++b) f.b32[b][g] = *(const f32x4*)(st + (b_row0 + b * 32) * 32 + pc);
      }
      return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      f.ah[a] = *(const bf16x8*)(st + (a_row0 + a * 32) * 32 + ((2 * kg + t) ^ fsw) * 4);
      if constexpr (prec_x3(PREC) || prec_x2(PREC) || PREC == PREC_F16) f.al[a] = *(const bf16x8*)(st + (a_row0 + a * 32) * 32 + ((4 + 2 * kg + t) ^ fsw) * 4);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      f.bh[b] = *(const bf16x8*)(st + (b_row0 + b * 32) * 32 + ((2 * kg + t) ^ fsw) * 4);
      if constexpr (prec_x3(PREC) || PREC == PREC_F16) f.bl[b] = *(const bf16x8*)(st + (b_row0 + b * 32) * 32 + ((4 + 2 * kg + t) ^ fsw) * 4);
    }
  };
  auto mma = [&](const Frag& f) {
    if constexpr (PREC == PREC_F32) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a32[a][g][e], f.b32[b][g][e], acc[a][b], 0, 0, 0);
      return;
    }
    if constexpr (prec_x3(PREC) || prec_x2(PREC)) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = ta_mfma16<PREC>(f.al[a], f.bh[b], acc[a][b]);
    }
    if constexpr (prec_x3(PREC)) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = ta_mfma16<PREC>(f.ah[a], f.bl[b], acc[a][b]);
    }
    if constexpr (PREC == PREC_F16) {                 // a row is 64 channels of plain halfs: its second 64 bytes are 32 more of K
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = ta_mfma16<PREC>(f.al[a], f.bl[b], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = ta_mfma16<PREC>(f.ah[a], f.bh[b], acc[a][b]);
  };
  constexpr int NREAD = PREC == PREC_BF16 ? 4 : (prec_x2(PREC) ? 6 : 8);          // ds_read_b128 per k-step
  constexpr int NMMA = PREC == PREC_F32 ? 32 : (prec_x3(PREC) ? 12 : ((PREC == PREC_F16 || prec_x2(PREC)) ? 8 : 4));         // MFMAs per k-step
  // pin "reads first, one per MFMA slot, then the remaining MFMAs": hipcc otherwise sinks the reads next to their
  // use to save registers and exposes the LDS latency in front of every group of MFMAs
  auto pin = [&]() {
#pragma unroll
    for (int i = 0; i < NREAD; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NMMA - NREAD, 0);
  };
  Frag F0, F1;
  if (wave == 0) TA_STAMP(1);                       // consumer: set up, waiting for slab 0
  __builtin_amdgcn_s_barrier();                     // B_0: slab 0 visible
  asm volatile("" ::: "memory");
  if (wave == 0) TA_STAMP(2);                       // consumer: slab 0 landed
  if constexpr (STAGES == 2) {
    // one fragment set: the other workgroup's consumer on this SIMD covers the read latency
    for (int s = 0; s < S; ++s) {
      const float* st = lds + (s & 1) * STAGE;
      load(F0, st, 0);
      mma(F0);
      load(F0, st, 1);
      mma(F0);
      if (s + 1 < S) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // B_{s+1}
        asm volatile("" ::: "memory");
      }
    }
  } else {
  load(F0, lds + stage * STAGE, 0);
  for (int s = 0; s + 1 < S; ++s) {                 // branch-free body; the last slab is peeled below
    const float* st = lds + stage * STAGE;
    stage = stage + 1 == STAGES ? 0 : stage + 1;
    __builtin_amdgcn_sched_barrier(0);
    load(F1, st, 1);                                // second k-step of slab s under the MFMAs of the first
    mma(F0);
    pin();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every fragment of slab s is in registers
    __builtin_amdgcn_s_barrier();                          // B_{s+1}: slab s+1 visible, stage of slab s handed back
    asm volatile("" ::: "memory");
    load(F0, lds + stage * STAGE, 0);               // first k-step of slab s+1 under the MFMAs of the second
    mma(F1);
    pin();
    __builtin_amdgcn_sched_barrier(0);
  }
  load(F1, lds + stage * STAGE, 1);
  stage = stage + 1 == STAGES ? 0 : stage + 1;
  mma(F0);
  mma(F1);
  }
  if (wave == 0) TA_STAMP(3);                       // consumer: main loop done (last MFMAs issued)
  {
    __builtin_amdgcn_s_barrier();                   // E0: every consumer has its last fragments: the ring can be reused
    asm volatile("" ::: "memory");
    if (wave == 0) TA_STAMP(5);                     // consumer: past E0
    conv_epilogue_park<BN>(acc, lds, cm, cn, lane);
    if (wave == 0) TA_STAMP(6);                     // consumer: accumulators parked (LDS writes issued)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // ... and executed: a raw s_barrier does not wait for them
    __builtin_amdgcn_s_barrier();                   // E1: tile parked
    asm volatile("" ::: "memory");
    if (wave == 0) TA_STAMP(7);                     // consumer: past E1
    bool done = false;
    if constexpr (PREC != PREC_F32) done = conv_drain_dispatch<BN, BM, 64 * (NC + NP), (PREC == PREC_F16 ? 2 : (prec_half(PREC) ? 1 : 0))>(p, lds, ct0, pt0, tid, HoWo);
    if (!done) conv_epilogue_drain<BN, BM, 64 * (NC + NP)>(p, lds, ct0, pt0, tid, HoWo, ks);
  }
  if (wave == 0) TA_STAMP(4);                       // consumer: epilogue stores issued
}

// ---- split-role kernel with a WINDOW-RESIDENT pixel operand (3x3 / 7x7 stride-1 convs on pre-split half-float tensors) --------
// conv_igemm_split streams both operands per K slab: over the kh x kw taps of one channel block the same input pixels are
// fetched kh x kw times from L2 into LDS (each time shifted by one tap).  Here the pixel operand of a channel block is loaded
// ONCE: the tile's BM pixels are consecutive interior pixels in raster order, so every tap of every one of them lies inside
// one contiguous run of the padded tensor -- from the first pixel's tap (0, 0) to the last pixel's tap (kh-1, kw-1), halo
// rows and, where a tile crosses into the next image, the halo rows between the images included.  That run (<= PR pixel rows
// of 128 bytes: one 32-channel block, [hi x32 | lo x32]) is the PATCH.  Producers DMA patch cb + 1 into the second patch
// buffer while the kh x kw slabs of block cb are consumed; per slab only the weight rows stream (BN x 128 bytes instead of
// (BN + BM) x 128).  A consumer lane keeps the patch row of its two pixels and reads tap (ky, kx) at row + ky * Wp + kx -- the
// same XOR swizzle on the row index, so the fragment reads stay conflict-free (16 consecutive rows per quarter wave).
// K order, MFMA order and epilogue are conv_igemm_split's: a layer's bits do not depend on which of the two kernels runs it.
// L2 -> LDS bytes per tile and channel block: kh kw BN 128 + ~1.5 BM 128 instead of kh kw (BN + BM) 128 (3x3, 128 x 128:
// 172 KiB instead of 288; the embedder's two-product mode is bound by exactly this stream).
template <int CM, int CN, int PREC, int PR>
__global__ __launch_bounds__(64 * (CM * CN + 4), (CM * CN + 4) / 4) void conv_igemm_win(const ta_conv_launch p) {
  static_assert(prec_half(PREC) && PREC != PREC_F16, "pre-split half-float tensors (f16x3 / f16x2)");
  static_assert(CM * CN == 4 || CM * CN == 8, "consumer grid: 1x4 (64 cout x 256 px), 2x2 (128 x 128) or 2x4 (128 x 256)");
  static_assert(PR % 32 == 0, "whole DMA instructions per producer wave");
  constexpr int NC = CM * CN, NP = 4;
  constexpr int BN = CM * 64, BM = CN * 64;
  constexpr int QA = BN / 8 / NP;                // weight-row DMA instructions per producer wave per slab
  constexpr int NPW = PR / 8 / NP;               // patch DMA instructions per producer wave per channel block
  constexpr int A_STAGE = BN * 32;               // floats per weight stage (3 stages)
  constexpr int PATCH = PR * 32;                 // floats per patch buffer (2 buffers)
  static_assert(QA + NPW < 64, "s_waitcnt vmcnt is a 6-bit count");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const patch0 = lds + 3 * A_STAGE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_ct = p.coutp / BN;
  const int n_pt = (p.M + BM - 1) / BM;
  const int bid = blockIdx.x;
  const int grp = bid >> 3, xcd = bid & 7;
  const int gq = ta_div_r(grp, n_ct, p.r_nct, p.fast_div);
  const int pt = ta_xcd_tile(n_pt, xcd, gq);
  if (pt < 0) return;
  const int ct0 = (grp - gq * n_ct) * BN;
  const int pt0 = pt * BM;
  const int HoWo = p.Ho * p.Wo;
  const int S = p.n_slabs;
  const int T = p.k_w * p.k_h;                   // slabs per channel block
  if (wave == 0) TA_STAMP(0);                       // consumer entry (tile decoded)
  if (wave == NC) TA_STAMP(8);                      // producer entry
  // padded-raster index of a pixel's tap (0, 0) relative to the tile's first pixel: the patch row it reads at that tap
  const ta_pixel_walk walk(p, pt0, HoWo);
  const int wp = p.win_wp, wimg = p.win_img;    // pixels per padded row / per padded image (launcher)

  if (wave >= NC) {
    // ================= producer =================
    const int pw = wave - NC;
    const int pchunk = lane & 7;
    const int lchunk = pchunk ^ ((4 * (pw & 1) + (lane >> 4)) & 7);
    const char* a_base = (const char*)p.w;
    const size_t a_slab_bytes = (size_t)p.coutp * 128;
    unsigned a_off[QA];
#pragma unroll
    for (int q = 0; q < QA; ++q) a_off[q] = (unsigned)(((ct0 + (q * NP + pw) * 8 + (lane >> 3)) * 32 + lchunk * 4) * 4);
    ta_k_walk wa(p, 0);
    auto issue_a = [&](int stage) {
#pragma unroll
      for (int q = 0; q < QA; ++q) ta_dma16(a_base + (size_t)wa.a_slab * a_slab_bytes, a_off[q], lds + stage * A_STAGE + (q * NP + pw) * 256);
      wa.advance();
    };
    issue_a(0);                                     // needs no pixel arithmetic: moving first
    // the patch: rows 0 .. n_patch - 1 of the padded tensor from the first pixel's tap (0, 0) on
    const int last = (p.M - pt0 < BM ? p.M - pt0 : BM) - 1;
    int img1, y1, x1;
    walk.at(last, img1, y1, x1);
    const int n_patch = (img1 - walk.img0) * wimg + (y1 - walk.y0) * wp + (x1 - walk.x0) + (p.k_h - 1) * wp + p.k_w;
    const int in_ch = p.in_ch_off + (p.group_cout ? (ct0 / p.group_cout) * p.group_cin : 0);
    const size_t off0 = (size_t)walk.img0 * p.in_img + (size_t)walk.y0 * p.in_row + (size_t)walk.x0 * p.in_pix + p.in_off0 + in_ch;
    const char* b_base = (const char*)(p.in + off0);
    unsigned p_off[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int r = (i * NP + pw) * 8 + (lane >> 3);                      // LDS row; rows past the patch re-read its last row
      const int rr = r < n_patch ? r : n_patch - 1;
      p_off[i] = (unsigned)rr * (unsigned)(p.in_pix * 4) + (unsigned)(lchunk * 16);
    }
    auto issue_patch = [&](int cb) {
      float* dst = patch0 + (cb & 1) * PATCH;
#pragma unroll
      for (int i = 0; i < NPW; ++i) ta_dma16(b_base + (size_t)cb * 128, p_off[i], dst + (i * NP + pw) * 256);
    };
    if (wave == NC) TA_STAMP(9);                    // producer: addresses ready, first weight rows issued
    issue_patch(0);
    if (S > 1) issue_a(1);
    if (wave == NC) TA_STAMP(10);                   // producer: patch 0 + second weight slab issued
    int stage = 2, t = 0, cb = 0;
    bool patch_behind = false;                      // a patch was issued right after the previous barrier
    for (int g = 0; g < S; ++g) {
      // vmcnt counts in issue order: [A0 P0 A1], then per barrier [P(cb+1) if tap 0] A(g+2).  Slab g's weight rows must have
      // landed; what may stay in flight is the next slab's rows and a patch issued behind slab g's rows
      const bool more = g + 1 < S;
      if (patch_behind) {
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QA + NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
      } else {
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                 // B_g: slab g (and, at tap 0, its patch) landed; consumers drained slab g-1
      asm volatile("" ::: "memory");
      patch_behind = false;
      if (t == 0 && cb + 1 < p.k_cblocks) {         // the other patch buffer was last read by block cb-1: free since this barrier
        issue_patch(cb + 1);
        patch_behind = true;
      }
      if (g + 2 < S) {
        issue_a(stage);
        stage = stage == 2 ? 0 : stage + 1;
      }
      if (++t == T) {
        t = 0;
        ++cb;
      }
    }
    // (a patch is never left in flight here: the last block issues none)
    {
      __builtin_amdgcn_s_barrier();                 // E0
      __builtin_amdgcn_s_barrier();                 // E1
      asm volatile("" ::: "memory");
      if (!conv_drain_dispatch<BN, BM, 64 * (NC + NP), 1>(p, lds, ct0, pt0, tid, HoWo))
        conv_epilogue_drain<BN, BM, 64 * (NC + NP)>(p, lds, ct0, pt0, tid, HoWo, 0);
    }
    return;
  }

  // ================= consumer =================
  // the MFMA waves ahead of the DMA-issuing producers in the CU's issue arbitration (TA_CONV_PRIO, A/B; s_setprio ignores EXEC)
  if (p.cons_prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (p.cons_prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (p.cons_prio == 3) __builtin_amdgcn_s_setprio(3);
  const int cm = wave / CN, cn = wave % CN;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int frow = lane & 31;
  const int fsw = (frow >> 1) & 7;
  const int kg = lane >> 5;
  const int a_row0 = cm * 64 + frow;
  int rB[2];                                        // patch row of this lane's two pixels at tap (0, 0)
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int d = cn * 64 + b * 32 + frow;
    int img, y, x;
    walk.at(pt0 + d < p.M ? d : 0, img, y, x);     // pixels past M: the tile's first pixel (their stores are masked)
    rB[b] = (img - walk.img0) * wimg + (y - walk.y0) * wp + (x - walk.x0);
  }
  struct Frag {
    bf16x8 ah[2], al[2], bh[2], bl[2];
  };
  unsigned pb[2];                                   // byte offset (inside the patch buffers) of this slab's B rows, k-step 0, hi words
  auto baddr = [&](int tap_off, int buf) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const unsigned row = (unsigned)(rB[b] + tap_off);
      pb[b] = (unsigned)buf * (unsigned)(PATCH * 4) + row * 128u + ((((unsigned)(2 * kg)) ^ ((row >> 1) & 7u)) << 4);
    }
  };
  auto load = [&](Frag& f, const float* st, int t) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      f.ah[a] = *(const bf16x8*)(st + (a_row0 + a * 32) * 32 + ((2 * kg + t) ^ fsw) * 4);
      f.al[a] = *(const bf16x8*)(st + (a_row0 + a * 32) * 32 + ((4 + 2 * kg + t) ^ fsw) * 4);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const char* q = (const char*)patch0 + (pb[b] ^ (unsigned)(t << 4));      // chunk 2 kg + t: bit 0 of the chunk index
      f.bh[b] = *(const bf16x8*)q;
      if constexpr (prec_x3(PREC)) f.bl[b] = *(const bf16x8*)((const char*)patch0 + ((pb[b] ^ (unsigned)(t << 4)) ^ 64u));   // + 4 chunks: the lo words
    }
  };
  auto mma = [&](const Frag& f) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = ta_mfma16<PREC>(f.al[a], f.bh[b], acc[a][b]);
    if constexpr (prec_x3(PREC)) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = ta_mfma16<PREC>(f.ah[a], f.bl[b], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = ta_mfma16<PREC>(f.ah[a], f.bh[b], acc[a][b]);
  };
  constexpr int NREAD = prec_x3(PREC) ? 8 : 6;
  constexpr int NMMA = prec_x3(PREC) ? 12 : 8;
  auto pin = [&]() {
#pragma unroll
    for (int i = 0; i < NREAD; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NMMA - NREAD, 0);
  };
  // K walk of the consumer side (scalar): tap offset inside the patch and the patch buffer of the slab being read
  int kx = 0, ky = 0, cb = 0, tap_off = 0, stage = 0;
  Frag F0, F1;
  baddr(0, 0);
  if (wave == 0) TA_STAMP(1);                       // consumer: set up, waiting for slab 0
  __builtin_amdgcn_s_barrier();                     // B_0
  asm volatile("" ::: "memory");
  if (wave == 0) TA_STAMP(2);                       // consumer: slab 0 + patch 0 landed
  load(F0, lds, 0);
  for (int g = 0; g + 1 < S; ++g) {
    const float* st = lds + stage * A_STAGE;
    stage = stage == 2 ? 0 : stage + 1;
    __builtin_amdgcn_sched_barrier(0);
    load(F1, st, 1);
    mma(F0);
    pin();
    __builtin_amdgcn_sched_barrier(0);
    // the next slab's tap (scalar walk) and its B addresses, while the reads of this slab return
    if (++kx == p.k_w) {
      kx = 0;
      tap_off += wp - p.k_w;
      if (++ky == p.k_h) {
        ky = 0;
        ++cb;
        tap_off = -1;
      }
    }
    ++tap_off;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every fragment of slab g is in registers
    baddr(tap_off, cb & 1);
    __builtin_amdgcn_s_barrier();                          // B_{g+1}
    asm volatile("" ::: "memory");
    load(F0, lds + stage * A_STAGE, 0);
    mma(F1);
    pin();
    __builtin_amdgcn_sched_barrier(0);
  }
  load(F1, lds + stage * A_STAGE, 1);
  mma(F0);
  mma(F1);
  if (wave == 0) TA_STAMP(3);                       // consumer: main loop done (last MFMAs issued)
  {
    __builtin_amdgcn_s_barrier();                   // E0: every consumer has its last fragments: ring and patches can be reused
    asm volatile("" ::: "memory");
    if (wave == 0) TA_STAMP(5);
    conv_epilogue_park<BN>(acc, lds, cm, cn, lane);
    if (wave == 0) TA_STAMP(6);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // E1: tile parked
    asm volatile("" ::: "memory");
    if (wave == 0) TA_STAMP(7);
    if (!conv_drain_dispatch<BN, BM, 64 * (NC + NP), 1>(p, lds, ct0, pt0, tid, HoWo))
      conv_epilogue_drain<BN, BM, 64 * (NC + NP)>(p, lds, ct0, pt0, tid, HoWo, 0);
  }
  if (wave == 0) TA_STAMP(4);                       // consumer: epilogue stores issued
}

// Second pass of a K-split conv: out = act(sum_k partial[k] + bias), ranges added in ascending order (deterministic).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ta_conv_launch p) {
  const int c4 = p.cout >> 2;
  const int total = p.M * c4;
  const int HoWo = p.Ho * p.Wo;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int pix = i / c4, co = (i - pix * c4) * 4;
    const int img = pix / HoWo;
    const int rem = pix - img * HoWo;
    const int y = rem / p.Wo, x = rem - y * p.Wo;
    f32x4 v = *(const f32x4*)(p.bias + co);
    const f32x4 us = *(const f32x4*)((p.bias + p.coutp) + co);
    if (p.bias9) {
      const int cls = ta_border_class(y, x, p.Ho, p.Wo);
      if (cls != TA_INTERIOR) v = *(const f32x4*)(p.bias9 + (size_t)cls * p.coutp + co);
    }
    for (int k = 0; k < p.k_split; ++k) {
      const f32x4 t = *(const f32x4*)(p.partial + ((size_t)k * p.M + pix) * p.coutp + co);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(t[e], us[e], v[e]);      // us == 1: v + t
    }
    if (p.act == TA_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ta_relu(v[e]);
    } else if (p.act == TA_ACT_PRELU) {
      const f32x4 sl = *(const f32x4*)(p.prelu + co);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * sl[e];
    }
    if (p.res) {                                     // the same order as the one-pass epilogues: activation, + shortcut, store, second output
      const int ry = p.res_up2 ? (y >> 1) : y, rx = p.res_up2 ? (x >> 1) : x;
      const f32x4 r = ta_ld4(p.res + (size_t)img * p.res_img + (size_t)ry * p.res_row + (size_t)rx * p.res_pix + p.res_off0, p.res_ch + co, p.res_fmt);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += r[e];
    }
    ta_st4(p.out + (size_t)img * p.out_img + (size_t)y * p.out_row + (size_t)x * p.out_pix + p.out_off0, p.out_ch + co,
           p.out_fmt, v);
    unsigned amax = ta_amax4(0u, v);
    if (p.out2) {
      const f32x4 sc = *(const f32x4*)(p.scale2 + co), sh = *(const f32x4*)(p.shift2 + co);
      f32x4 z;
#pragma unroll
      for (int e = 0; e < 4; ++e) z[e] = v[e] * sc[e] + sh[e];
      ta_st4(p.out2 + (size_t)img * p.o2_img + (size_t)y * p.o2_row + (size_t)x * p.o2_pix + p.o2_off0, p.o2_ch + co, p.o2_fmt, z);
      amax = ta_amax4(amax, z);
    }
    if (p.range_check) ta_range_report(p, amax);
  }
}

template <int CM, int CN, int NP, int PREC, int STAGES>
static int launch_split(ta_ctx* ctx, const ta_conv_launch& p) {
  constexpr int BN = CM * 64, BM = CN * 64;
  const int n_ct = p.coutp / BN;
  const int n_pt = (p.M + BM - 1) / BM;
  const int groups = ((n_pt + 7) / 8) * n_ct;
  const size_t lds_bytes = (size_t)STAGES * (BN + BM) * 32 * sizeof(float);
  auto kern = conv_igemm_split<CM, CN, NP, PREC, STAGES>;
  TA_SET_LDS_ATTR(ctx, kern, lds_bytes);
  ta_conv_launch q = p;
  const long long grid = (long long)groups * 8 * p.k_split;
  q.r_nct = 1.0f / (float)n_ct;
  q.r_tile_blocks = 1.0f / (float)(groups * 8);
  q.r_Wo = 1.0f / (float)p.Wo;
  q.r_Ho = 1.0f / (float)p.Ho;
  q.r_HoWo = 1.0f / (float)(p.Ho * p.Wo);
  {
    // lean epilogue: split-format tensors whose byte offsets fit 32 bits, channel slices on 8-channel boundaries
    static const bool no_fast_drain = getenv("TA_CONV_NO_FASTDRAIN") != nullptr;      // tools: A/B
    const long long n_img = p.Ho * p.Wo > 0 ? ((long long)p.M + p.Ho * p.Wo - 1) / (p.Ho * p.Wo) : 0;
    auto fits = [&](long long img_stride, int off0) { return ((n_img + 1) * img_stride + off0) * 4 < (1LL << 32); };
    constexpr int SPLIT_FMT = PREC == PREC_F16 ? TA_FMT_F16 : (prec_half(PREC) ? TA_FMT_SPLIT16 : TA_FMT_SPLIT);
    bool ok = !no_fast_drain && PREC != PREC_F32 && p.k_split == 1 && !p.direct_epilogue && p.out_fmt == SPLIT_FMT &&
              (p.cout & 7) == 0 && ((p.out_ch | p.res_ch | p.o2_ch) & 7) == 0 && fits(p.out_img, p.out_off0);
    if (p.res) ok = ok && p.res_fmt == SPLIT_FMT && fits(p.res_img, p.res_off0);
    if (p.out2) ok = ok && p.o2_fmt == SPLIT_FMT && fits(p.o2_img, p.o2_off0);
    q.fast_drain = ok ? 1 : 0;
    {
      static const int prio = getenv("TA_CONV_PRIO") ? atoi(getenv("TA_CONV_PRIO")) & 3 : 0;     // tools: A/B of the consumer waves' issue priority
      q.cons_prio = prio;
    }
    if (ok) ctx->conv_counts[TA_CV_COUNT - 1] += 1;     // slot 15: launches whose epilogue ran the specialised drain
  }
  static const bool late_b = getenv("TA_CONV_LATE_B") != nullptr;             // tools: A/B of the early slab-0 pixel-row DMAs
  if (late_b) q.probe |= 4;
  static const bool no_fast_div = getenv("TA_CONV_NO_FASTDIV") != nullptr;    // tools: A/B of the division-free set-up
  q.fast_div = (!no_fast_div && grid < (1 << 24) && (long long)p.M + BM < (1 << 24)) ? 1 : 0;
  {
    static char name[64];
    if (!name[0]) snprintf(name, sizeof(name), "conv_igemm_split<%d,%d,%d,%d,%d>", CM, CN, NP, PREC, STAGES);
    ctx->note_kernel(name);
  }
  hipLaunchKernelGGL(kern, dim3(groups * 8 * p.k_split), dim3(64 * (CM * CN + NP)), lds_bytes, ctx->stream, q);
  TA_HIP(ctx, hipGetLastError());
  if (p.k_split > 1) {
    const int total = p.M * (p.cout >> 2);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, p);
    TA_HIP(ctx, hipGetLastError());
  }
  return TA_OK;
}

// patch rows a BM-pixel tile of this launch can need at most (conv_igemm_win): BM - 1 raster steps, each output row crossed adds
// the two halo columns, each image crossed the halo rows between the images, plus the taps of the last pixel
static int win_patch_rows(const ta_conv_launch& p, int BM) {
  if (p.Wo <= 0 || p.Ho <= 0 || p.win_wp <= 0) return 1 << 30;
  const long long n = BM - 1, hp = p.win_img / p.win_wp;
  const long long rows = n + ((n + p.Wo - 1) / p.Wo) * (p.win_wp - p.Wo) + ((n + (long long)p.Ho * p.Wo - 1) / ((long long)p.Ho * p.Wo)) * (hp - p.Ho) * p.win_wp +
                         (long long)(p.k_h - 1) * p.win_wp + p.k_w;
  return rows > (1 << 30) ? (1 << 30) : (int)rows;
}
// what the window kernels are instantiated with: patch capacity per tile shape (LDS: 3 weight stages + 2 patches <= 160 KiB)
#define TA_WIN_PR_2x2 384
#define TA_WIN_PR_2x4 448
#define TA_WIN_PR_1x4 512

template <int CM, int CN, int PREC, int PR>
static int launch_win(ta_ctx* ctx, const ta_conv_launch& p) {
  constexpr int BN = CM * 64, BM = CN * 64;
  const int n_ct = p.coutp / BN;
  const int n_pt = (p.M + BM - 1) / BM;
  const int groups = ((n_pt + 7) / 8) * n_ct;
  constexpr size_t ring = (size_t)(3 * BN + 2 * PR) * 128, park = (size_t)BM * BN * 4;
  constexpr size_t lds_bytes = ring > park ? ring : park;
  static_assert(lds_bytes <= 160 * 1024, "one workgroup's LDS");
  auto kern = conv_igemm_win<CM, CN, PREC, PR>;
  TA_SET_LDS_ATTR(ctx, kern, lds_bytes);
  ta_conv_launch q = p;
  const long long grid = (long long)groups * 8;
  q.r_nct = 1.0f / (float)n_ct;
  q.r_tile_blocks = 1.0f / (float)(groups * 8);
  q.r_Wo = 1.0f / (float)p.Wo;
  q.r_Ho = 1.0f / (float)p.Ho;
  q.r_HoWo = 1.0f / (float)(p.Ho * p.Wo);
  {
    static const bool no_fast_drain = getenv("TA_CONV_NO_FASTDRAIN") != nullptr;
    const long long n_img = ((long long)p.M + p.Ho * p.Wo - 1) / (p.Ho * p.Wo);
    auto fits = [&](long long img_stride, int off0) { return ((n_img + 1) * img_stride + off0) * 4 < (1LL << 32); };
    bool ok = !no_fast_drain && p.out_fmt == TA_FMT_SPLIT16 && (p.cout & 7) == 0 && ((p.out_ch | p.res_ch | p.o2_ch) & 7) == 0 && fits(p.out_img, p.out_off0);
    if (p.res) ok = ok && p.res_fmt == TA_FMT_SPLIT16 && fits(p.res_img, p.res_off0);
    if (p.out2) ok = ok && p.o2_fmt == TA_FMT_SPLIT16 && fits(p.o2_img, p.o2_off0);
    q.fast_drain = ok ? 1 : 0;
    {
      static const int prio = getenv("TA_CONV_PRIO") ? atoi(getenv("TA_CONV_PRIO")) & 3 : 0;     // tools: A/B of the consumer waves' issue priority
      q.cons_prio = prio;
    }
    if (ok) ctx->conv_counts[TA_CV_COUNT - 1] += 1;
  }
  q.fast_div = (grid < (1 << 24) && (long long)p.M + BM < (1 << 24)) ? 1 : 0;
  q.k_split = 1;
  {
    static char name[64];
    if (!name[0]) snprintf(name, sizeof(name), "conv_igemm_win<%d,%d,%d,%d>", CM, CN, PREC, PR);
    ctx->note_kernel(name);
  }
  hipLaunchKernelGGL(kern, dim3(groups * 8), dim3(64 * (CM * CN + 4)), lds_bytes, ctx->stream, q);
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

template <int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES, int PREC>
static int launch_cfg(ta_ctx* ctx, const ta_conv_launch& p) {
  constexpr int BN = WAVES_M * WM_TILES * 32;
  constexpr int BM = WAVES_N * WN_TILES * 32;
  const int n_ct = p.coutp / BN;
  const int n_pt = (p.M + BM - 1) / BM;
  const int groups = ((n_pt + 7) / 8) * n_ct;
  const size_t lds_bytes = 2 * (size_t)(BN + BM) * 32 * sizeof(float);
  // two instances: the usual one drains through LDS only; channel slices off the 8-channel boundaries (or the A/B switch
  // TA_CONV_DIRECT_EPILOGUE) take the one that also carries the direct epilogue -- and pays for it in registers
  const bool staged = ((p.out_ch | p.res_ch | p.o2_ch | p.direct_epilogue) & 7) == 0 && (p.cout & 3) == 0;
  {
    static char name[2][64];
    if (!name[0][0])
      for (int d = 0; d < 2; ++d)
        snprintf(name[d], sizeof(name[d]), "conv_igemm<%d,%d,%d,%d,%d,%s>", WAVES_M, WAVES_N, WM_TILES, WN_TILES, PREC, d ? "true" : "false");
    ctx->note_kernel(name[staged ? 0 : 1]);
  }
  if (staged) {
    auto kern = conv_igemm<WAVES_M, WAVES_N, WM_TILES, WN_TILES, PREC, false>;
    TA_SET_LDS_ATTR(ctx, kern, lds_bytes);
    hipLaunchKernelGGL(kern, dim3(groups * 8), dim3(256), lds_bytes, ctx->stream, p);
  } else {
    auto kern = conv_igemm<WAVES_M, WAVES_N, WM_TILES, WN_TILES, PREC, true>;
    TA_SET_LDS_ATTR(ctx, kern, lds_bytes);
    hipLaunchKernelGGL(kern, dim3(groups * 8), dim3(256), lds_bytes, ctx->stream, p);
  }
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

template <int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES, int PREC, int STAGES, bool BSPLIT>
static int launch_pipe(ta_ctx* ctx, const ta_conv_launch& p) {
  constexpr int BN = WAVES_M * WM_TILES * 32;
  constexpr int BM = WAVES_N * WN_TILES * 32;
  const int n_ct = p.coutp / BN;
  const int n_pt = (p.M + BM - 1) / BM;
  const int groups = ((n_pt + 7) / 8) * n_ct;
  const size_t lds_bytes = (size_t)STAGES * (BN + BM) * 32 * sizeof(float);
  auto kern = conv_igemm_pipe<WAVES_M, WAVES_N, WM_TILES, WN_TILES, PREC, STAGES, BSPLIT>;
  {
    static char name[80];
    if (!name[0]) snprintf(name, sizeof(name), "conv_igemm_pipe<%d,%d,%d,%d,%d,%d,%s>", WAVES_M, WAVES_N, WM_TILES, WN_TILES, PREC, STAGES, BSPLIT ? "true" : "false");
    ctx->note_kernel(name);
  }
  TA_SET_LDS_ATTR(ctx, kern, lds_bytes);
  hipLaunchKernelGGL(kern, dim3(groups * 8), dim3(256), lds_bytes, ctx->stream, p);
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

// ---------------------------------------------------------------------------------------------------
// rf_dwpw_kernel (round 6): the SAME [depthwise 3x3 -> 1x1] block as conv_dwpw above, written for the one shape of work the
// detector's deep base has -- split-half (f16x3) 1x1, input channels a multiple of 32, output channels a multiple of 8, bias +
// ReLU epilogue -- instead of instantiated from the generic tile machinery.  conv_dwpw spends ~15 M wave-instructions on a
// 40 x 40 x 128 block whose arithmetic needs ~1.4 M (profiles/r06_dwpw_*): run-time epilogue flags, 64-bit addressing, the
// pixel operand re-converted by every wave that consumes it, ten global loads of depthwise weights per slab.  Here:
//   * taps are buffer loads: one 32-bit byte offset per pixel row, tap and slab offsets in the scalar offset -- no vector address math;
//   * the depthwise weights of ALL channels sit in LDS ([9 taps + bias][C]), loaded once per workgroup;
//   * a row is split into half floats ONCE, by the thread that computed it, and stored as the [hi x32 | lo x32] image the weight
//     rows already have: every consumer wave reads ready fragments (ds_read_b128), no VALU between LDS and MFMA;
//   * the drain's bias / un-scale vectors are loaded before the K loop; the drain is fma + max + store with 32-bit offsets.
// Same products in the same order as conv_dwpw (al*bh, ah*bl, ah*bh per k-step; bias + fmaf chain in (ky, kx) order; fmaf(acc, us,
// bias) in the drain): a block's output bits do not depend on which of the two kernels ran it (tests pin both).
template <int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES>
__global__ __launch_bounds__(256, 2) void rf_dwpw_kernel(const ta_conv_launch p) {
  constexpr int BN = WAVES_M * WM_TILES * 32, BM = WAVES_N * WN_TILES * 32;
  constexpr int QA = BN / 32, RP = BM / 32, STAGE = (BN + BM) * 32;
  static_assert(WAVES_M * WAVES_N == 4 && (RP == 2 || RP == 4), "4 waves; 64 or 128 pixels per tile");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int n_ct = p.coutp / BN;
  const int bid = blockIdx.x;
  const int grp = bid >> 3, xcd = bid & 7;
  const int ct = grp % n_ct;
  const int n_pt = (p.M + BM - 1) / BM;
  const int pt = ta_xcd_tile(n_pt, xcd, grp / n_ct);
  if (pt < 0) return;
  const int ct0 = ct * BN, pt0 = pt * BM;
  const int HoWo = p.Ho * p.Wo;
  const int C = p.dw_c, S = p.n_slabs;

  // drain constants of this lane's 8 output channels, and the depthwise weight table: issued first
  f32x4 pre[4];                                      // bias (2 x 4 channels), un-scale (2 x 4): both padded to coutp
  {
    const int pco = ct0 + 8 * (tid % (BN / 8));
    pre[0] = *(const f32x4*)(p.bias + pco);
    pre[1] = *(const f32x4*)(p.bias + pco + 4);
    pre[2] = *(const f32x4*)(p.bias + p.coutp + pco);
    pre[3] = *(const f32x4*)(p.bias + p.coutp + pco + 4);
  }
  float* wl = lds + 2 * STAGE;                       // [10][C]
  const int n_gran = 10 * C / 4;
  constexpr int WG = 3;
  f32x4 wtmp[WG];
#pragma unroll
  for (int i = 0; i < WG; ++i) {
    const int g = tid + 256 * i;
    wtmp[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g < n_gran) wtmp[i] = g * 4 < 9 * C ? *(const f32x4*)(p.dw_w + g * 4) : *(const f32x4*)(p.dw_bias + (g * 4 - 9 * C));
  }

  // weight rows of a slab: LDS DMA as in conv_dwpw
  const int pchunk = lane & 7;
  const int lchunk = pchunk ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  const char* a_src[QA];
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    const int row = (q * 4 + wave) * 8 + (lane >> 3);
    a_src[q] = (const char*)(p.w + ((size_t)(ct0 + row)) * 32 + lchunk * 4);
  }
  const size_t a_slab_bytes = (size_t)p.coutp * 128;
  auto dma_w = [&](int s, int stage) {
    float* base = lds + stage * STAGE;
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int t = q * 4 + wave;
      __builtin_amdgcn_global_load_lds(GLB_PTR(a_src[q] + (size_t)s * a_slab_bytes), LDS_PTR(base + t * 256), 16, 0, 0);
    }
  };

  // pixel rows: thread -> 4 channels c4 of the slab, rows (tid >> 3) + 32 j; a row's taps are one byte offset + scalar offsets
  const int c4 = tid & 7;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, -1, 0x00020000);
  int voff[RP];
#pragma unroll
  for (int j = 0; j < RP; ++j) {
    const int row = (tid >> 3) + 32 * j;
    int pix = pt0 + row;
    if (pix >= p.M) pix = 0;                        // clamp: the store is masked
    const int img = pix / HoWo;
    const int rem = pix - img * HoWo;
    const int y = rem / p.Wo;
    const int x = rem - y * p.Wo;
    voff[j] = 4 * (img * p.in_img + y * p.dw_stride * p.in_row + x * p.dw_stride * p.in_pix + p.in_off0) + 16 * c4;
  }
  const int row_b = 4 * p.in_row, pix_b = 4 * p.in_pix;

  unsigned dw_amax = 0;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // depthwise rows of slab s (two pixel rows of this thread at a time: 18 loads in flight), split, into the stage
  auto produce = [&](int s, int stage) {
    float* base = lds + stage * STAGE;
    const int ch = s * 32 + c4 * 4;
    f32x4 w9[9], bias;
#pragma unroll
    for (int t = 0; t < 9; ++t) w9[t] = *(const f32x4*)(wl + t * C + ch);
    bias = *(const f32x4*)(wl + 9 * C + ch);
#pragma unroll
    for (int j0 = 0; j0 < RP; j0 += 2) {
      f32x4 v[2][9];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#ifdef TA_CONV_TRACE      // debug build only (TA_DWPW_PROBE bit 0: no tap loads; timing ablation, WRONG results)
            v[jj][ky * 3 + kx] = (p.probe & 1) ? f32x4{1.f, 1.f, 1.f, 1.f} : __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[j0 + jj], ky * row_b + kx * pix_b + s * 128, 0));
#else
            v[jj][ky * 3 + kx] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[j0 + jj], ky * row_b + kx * pix_b + s * 128, 0));
#endif
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int row = (tid >> 3) + 32 * (j0 + jj);
        f32x4 acc = bias;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(v[jj][t][e], w9[t][e], acc[e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = ta_relu(acc[e]);
        dw_amax = ta_amax4(dw_amax, acc);            // these rows are split into half floats: range-checked like a stored tensor
        unsigned h0, h1, l0, l1;
        ta_pack2<true>(acc[0], acc[1], h0, l0);
        ta_pack2<true>(acc[2], acc[3], h1, l1);
        const int sw = (row >> 1) & 7;
        float* r = base + (BN + row) * 32 + (c4 & 1) * 2;
        *(uint2*)(r + (((c4 >> 1)) ^ sw) * 4) = make_uint2(h0, h1);
        *(uint2*)(r + ((4 + (c4 >> 1)) ^ sw) * 4) = make_uint2(l0, l1);
      }
    }
  };

  f32x16 acc[WM_TILES][WN_TILES];
#pragma unroll
  for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
    for (int b = 0; b < WN_TILES; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int frow = lane & 31;
  const int fsw = (frow >> 1) & 7;
  const int kg = lane >> 5;
  const int a_row0 = wm * WM_TILES * 32 + frow;
  const int b_row0 = BN + wn * WN_TILES * 32 + frow;
  auto mma = [&](int s) {
    const float* st = lds + (s & 1) * STAGE;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16x8 ah[WM_TILES], al[WM_TILES], bh[WN_TILES], bl[WN_TILES];
#pragma unroll
      for (int a = 0; a < WM_TILES; ++a) {
        ah[a] = *(const bf16x8*)(st + (a_row0 + a * 32) * 32 + ((2 * kg + t) ^ fsw) * 4);
        al[a] = *(const bf16x8*)(st + (a_row0 + a * 32) * 32 + ((4 + 2 * kg + t) ^ fsw) * 4);
      }
#pragma unroll
      for (int b = 0; b < WN_TILES; ++b) {
        bh[b] = *(const bf16x8*)(st + (b_row0 + b * 32) * 32 + ((2 * kg + t) ^ fsw) * 4);
        bl[b] = *(const bf16x8*)(st + (b_row0 + b * 32) * 32 + ((4 + 2 * kg + t) ^ fsw) * 4);
      }
#pragma unroll
      for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
        for (int b = 0; b < WN_TILES; ++b) acc[a][b] = ta_mfma16<PREC_F16X3>(al[a], bh[b], acc[a][b]);
#pragma unroll
      for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
        for (int b = 0; b < WN_TILES; ++b) acc[a][b] = ta_mfma16<PREC_F16X3>(ah[a], bl[b], acc[a][b]);
#pragma unroll
      for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
        for (int b = 0; b < WN_TILES; ++b) acc[a][b] = ta_mfma16<PREC_F16X3>(ah[a], bh[b], acc[a][b]);
    }
  };

  dma_w(0, 0);
#pragma unroll
  for (int i = 0; i < WG; ++i)
    if (tid + 256 * i < n_gran) *(f32x4*)(wl + (tid + 256 * i) * 4) = wtmp[i];
  for (int g = tid + 256 * WG; g < n_gran; g += 256)   // (more than 768 granules: 300+ input channels)
    *(f32x4*)(wl + g * 4) = g * 4 < 9 * C ? *(const f32x4*)(p.dw_w + g * 4) : *(const f32x4*)(p.dw_bias + (g * 4 - 9 * C));
  __syncthreads();                                   // the weight table is complete
  produce(0, 0);
  for (int s = 0; s < S; ++s) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                 // slab s complete (weights landed, rows written); the other stage is free
    if (s + 1 < S) {
#ifdef TA_CONV_TRACE      // debug build only (TA_DWPW_PROBE bit 2: weight rows DMA'd for slab 0 only; bit 3: no MFMAs)
      if (!(p.probe & 4))
#endif
      dma_w(s + 1, (s + 1) & 1);
      produce(s + 1, (s + 1) & 1);
    }
#ifdef TA_CONV_TRACE
    if (!(p.probe & 8))
#endif
    mma(s);
  }
  if (dw_amax > TA_F16_MAX_BITS) *p.range_flag = 1;
  if (p.amax_index >= 0 && dw_amax) atomicMax((unsigned*)p.range_flag + TA_AMAX_SLOT0 + 2 * p.amax_index + 1, dw_amax);

  // ---- epilogue: park the raw tile [pixel][cout] in the ring, drain one lane per (pixel, 8 channels) ----------------------
  constexpr int NCH = BN / 4, G = BN / 8, RPI = 256 / G;
  __syncthreads();
#pragma unroll
  for (int b = 0; b < WN_TILES; ++b) {
    const int row = (wn * WN_TILES + b) * 32 + (lane & 31);
#pragma unroll
    for (int a = 0; a < WM_TILES; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = ((wm * WM_TILES + a) * 32 + 8 * j + 4 * (lane >> 5)) >> 2;
        *(f32x4*)(lds + (row * NCH + (c ^ (row & (NCH - 1)))) * 4) =
            f32x4{acc[a][b][4 * j], acc[a][b][4 * j + 1], acc[a][b][4 * j + 2], acc[a][b][4 * j + 3]};
      }
  }
  __syncthreads();
  const int k8 = tid % G, r0 = tid / G;
  const int co = ct0 + 8 * k8;
  unsigned amax = 0;
  if (co < p.cout) {                                 // cout % 8 == 0: a lane is inside or outside with all 8 channels
    const bool split = p.out_fmt == TA_FMT_SPLIT16;  // uniform
    const bool chk = p.range_check != 0;
    char* const ob = (char*)p.out + (split ? ta_split_chan(p.out_ch + co) : 4u * (unsigned)(p.out_ch + co));
    int pix = pt0 + r0;
    int img = pix / HoWo;
    int rem = pix - img * HoWo;
    int y = rem / p.Wo;
    int x = rem - y * p.Wo;
    for (int row = r0; row < BM; row += RPI, pix += RPI) {
      if (pix >= p.M) break;
      const int sw = row & (NCH - 1);
      float v[8];
      *(f32x4*)v = *(const f32x4*)(lds + (row * NCH + ((2 * k8) ^ sw)) * 4);
      *(f32x4*)(v + 4) = *(const f32x4*)(lds + (row * NCH + ((2 * k8 + 1) ^ sw)) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = ta_relu(__builtin_fmaf(v[e], pre[2][e], pre[0][e]));
        v[4 + e] = ta_relu(__builtin_fmaf(v[4 + e], pre[3][e], pre[1][e]));
      }
      char* o = ob + 4u * (unsigned)(img * p.out_img + y * p.out_row + x * p.out_pix + p.out_off0);
#ifdef TA_CONV_TRACE      // debug build only (TA_DWPW_PROBE bit 1: no output stores)
      if ((p.probe & 2) && v[0] != 12345.f) {
        amax = max(amax, ta_absbits(v[0]));
        continue;
      }
#endif
      if (split) {
        unsigned am = 0;
        ta_split_store8<1>(o, v, am);
        amax = max(amax, am);
      } else {
        *(f32x4*)o = *(const f32x4*)v;
        *(f32x4*)(o + 16) = *(const f32x4*)(v + 4);
        if (chk) {
#pragma unroll
          for (int e = 0; e < 8; ++e) amax = max(amax, ta_absbits(v[e]));
        }
      }
      x += RPI;                                      // next pass: RPI pixels further in raster order
      while (x >= p.Wo) {
        x -= p.Wo;
        if (++y == p.Ho) {
          y = 0;
          ++img;
        }
      }
    }
  }
  ta_range_report(p, amax);
}

template <int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES>
static int launch_rf_dwpw(ta_ctx* ctx, const ta_conv_launch& p) {
  constexpr int BN = WAVES_M * WM_TILES * 32, BM = WAVES_N * WN_TILES * 32;
  const int n_ct = p.coutp / BN;
  const int n_pt = (p.M + BM - 1) / BM;
  const int groups = ((n_pt + 7) / 8) * n_ct;
  const size_t lds_bytes = 2 * (size_t)(BN + BM) * 32 * sizeof(float) + (size_t)40 * p.dw_c;   // two stages + the depthwise weight table [10][C]
  auto kern = rf_dwpw_kernel<WAVES_M, WAVES_N, WM_TILES, WN_TILES>;
  {
    static char name[64];
    if (!name[0]) snprintf(name, sizeof(name), "rf_dwpw_kernel<%d,%d,%d,%d>", WAVES_M, WAVES_N, WM_TILES, WN_TILES);
    ctx->note_kernel(name);
  }
  TA_SET_LDS_ATTR(ctx, kern, lds_bytes);
  hipLaunchKernelGGL(kern, dim3(groups * 8), dim3(256), lds_bytes, ctx->stream, p);
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

template <int WAVES_M, int WAVES_N, int WM_TILES, int WN_TILES, int PREC>
static int launch_dwpw_cfg(ta_ctx* ctx, const ta_conv_launch& p) {
  constexpr int BN = WAVES_M * WM_TILES * 32;
  constexpr int BM = WAVES_N * WN_TILES * 32;
  const int n_ct = p.coutp / BN;
  const int n_pt = (p.M + BM - 1) / BM;
  const int groups = ((n_pt + 7) / 8) * n_ct;
  const size_t lds_bytes = 2 * (size_t)(BN + BM) * 32 * sizeof(float);
  auto kern = conv_dwpw<WAVES_M, WAVES_N, WM_TILES, WN_TILES, PREC>;
  {
    static char name[64];
    if (!name[0]) snprintf(name, sizeof(name), "conv_dwpw<%d,%d,%d,%d,%d>", WAVES_M, WAVES_N, WM_TILES, WN_TILES, PREC);
    ctx->note_kernel(name);
  }
  TA_SET_LDS_ATTR(ctx, kern, lds_bytes);
  hipLaunchKernelGGL(kern, dim3(groups * 8), dim3(256), lds_bytes, ctx->stream, p);
  TA_HIP(ctx, hipGetLastError());
  return TA_OK;
}

int ta_launch_dwpw(ta_ctx* ctx, const ta_conv_launch& p_in, double flops) {
  if (p_in.M <= 0) return TA_OK;
  ta_conv_launch p = p_in;
  p.range_flag = ctx->range_flag;
  if ((p.prec != PREC_F32 && p.prec != PREC_F16X3) || p.in_fmt != TA_FMT_F32 || p.coutp % 32 || p.cout % 4 || (p.out_ch & 7) || p.dw_c % 4 || !p.dw_w ||
      !p.dw_bias || p.n_slabs * 32 < p.dw_c)
    return ta_fail(ctx, TA_E_INVALID, "dw+pw: needs the f32 or f16x3 mode, float32 input activations and 4-aligned channels");
  ta_prof_scope scope(ctx, 0, flops);
  ctx->cur_flops = flops;
#ifdef TA_CONV_TRACE
  static const int dw_probe = getenv("TA_DWPW_PROBE") ? atoi(getenv("TA_DWPW_PROBE")) & 15 : 0;   // debug build: timing ablations (conv_dwpw reads bits 0..1 as a number, rf_dwpw_kernel bits 0..3 as flags)
  p.probe = dw_probe;
#endif
  if (p.prec == PREC_F16X3) {
    // the lean kernel wherever the block has the shape it is written for (every f16x3 block of the detector's base); TA_DWPW_GENERIC: A/B
    const bool generic_only = getenv("TA_DWPW_GENERIC") != nullptr;      // read per launch: the parity test flips it inside one process
    const bool lean_ok = !generic_only && p.dw_c % 32 == 0 && p.dw_c == p.n_slabs * 32 && p.dw_c <= 1024 && p.cout % 8 == 0 && p.act == TA_ACT_RELU &&
                         (p.out_fmt == TA_FMT_F32 || p.out_fmt == TA_FMT_SPLIT16);
    if (lean_ok) {
      if (p.coutp % 128 == 0) return launch_rf_dwpw<2, 2, 2, 1>(ctx, p);
      if (p.coutp % 64 == 0) return launch_rf_dwpw<2, 2, 1, 1>(ctx, p);
      return launch_rf_dwpw<1, 4, 1, 1>(ctx, p);
    }
    if (p.coutp % 128 == 0) return launch_dwpw_cfg<2, 2, 2, 2, PREC_F16X3>(ctx, p);
    if (p.coutp % 64 == 0) return launch_dwpw_cfg<1, 4, 2, 1, PREC_F16X3>(ctx, p);
    return launch_dwpw_cfg<1, 4, 1, 1, PREC_F16X3>(ctx, p);
  }
  if (p.coutp % 128 == 0) return launch_dwpw_cfg<2, 2, 2, 2, PREC_F32>(ctx, p);
  if (p.coutp % 64 == 0) return launch_dwpw_cfg<1, 4, 2, 1, PREC_F32>(ctx, p);
  return launch_dwpw_cfg<1, 4, 1, 1, PREC_F32>(ctx, p);
}

// ---- kernel selection ---------------------------------------------------------------------------------------------
// Which variants can run this conv at all (a forced variant that cannot is an error, never a silent substitution).
static bool variant_eligible(int v, const ta_conv_launch& p) {
  const bool split_in = p.in_fmt == ta_split_fmt_of(p.prec);   // what the split-role kernel reads: float32 in f32 mode, else the mode's pre-split format
  const bool deep = p.uniform_k && (p.n_slabs >= 2 || p.prec == PREC_F16);   // f16 mode: a 1x1 conv over 64 channels is ONE slab of 64
  // the split-role kernel has the LDS-staged epilogue only: every channel slice on an 8-channel boundary
  const bool staged = ((p.out_ch | p.res_ch | p.o2_ch) & 7) == 0 && (p.cout & 3) == 0;
  switch (v) {
    case TA_CV_GENERIC: return p.in_fmt == TA_FMT_F32 && !p.group_cout;
    case TA_CV_PIPE64: return deep && staged && p.coutp % 64 == 0 && !p.group_cout && (p.in_fmt == TA_FMT_F32 || (split_in && p.prec != PREC_F16));
    case TA_CV_PIPE128: return deep && staged && p.coutp % 128 == 0 && !p.group_cout && p.in_fmt == TA_FMT_F32;
    case TA_CV_SPLIT_2x2:
    case TA_CV_SPLIT_2x2_P8:
    case TA_CV_SPLIT_2x4: return deep && split_in && staged && p.coutp % 128 == 0;
    case TA_CV_SPLIT_1x4: return deep && split_in && staged && p.coutp % 64 == 0 && (!p.group_cout || p.group_cout % 64 == 0);
    case TA_CV_SPLIT_2x2_W2: return deep && split_in && staged && p.coutp % 128 == 0 && p.k_split <= 1;
    case TA_CV_SPLIT_1x4_W2: return deep && split_in && staged && p.coutp % 64 == 0 && (!p.group_cout || p.group_cout % 64 == 0) && p.k_split <= 1;
    case TA_CV_WIN_2x2:
    case TA_CV_WIN_2x4:
    case TA_CV_WIN_1x4: {
      // window-resident pixel operand: stride-1 convs with >= 4 taps on pre-split half-float tensors (f16x3 / f16x2), no fused
      // pool, no K ranges, and a patch that fits the instantiated capacity
      const bool base = deep && split_in && staged && (p.prec == PREC_F16X3 || p.prec == PREC_F16X2) && p.stride == 1 && !p.pool && p.k_split <= 1 &&
                        p.k_w * p.k_h >= 4 && p.n_slabs == p.k_w * p.k_h * p.k_cblocks;
      if (!base) return false;
      if (v == TA_CV_WIN_1x4) return p.coutp % 64 == 0 && (!p.group_cout || p.group_cout % 64 == 0) && win_patch_rows(p, 256) <= TA_WIN_PR_1x4;
      if (p.coutp % 128) return false;
      return v == TA_CV_WIN_2x2 ? win_patch_rows(p, 128) <= TA_WIN_PR_2x2 : win_patch_rows(p, 256) <= TA_WIN_PR_2x4;
    }
  }
  return false;
}

// The automatic choice.
static int choose_variant_streamed(const ta_conv_launch& p);
static int choose_variant(const ta_conv_launch& p) {
  const int v = choose_variant_streamed(p);
  // the same tile with the pixel operand window-resident, wherever the layer qualifies (stride-1 3x3 / 7x7 on pre-split half-float
  // tensors whose patch fits): same bits, fewer L2 -> LDS bytes.  TA_CONV_NO_WIN: A/B switch for tools
  static const bool no_win = getenv("TA_CONV_NO_WIN") != nullptr;
  if (no_win) return v;
  const int w = v == TA_CV_SPLIT_2x2 ? TA_CV_WIN_2x2 : (v == TA_CV_SPLIT_2x4 ? TA_CV_WIN_2x4 : (v == TA_CV_SPLIT_1x4 ? TA_CV_WIN_1x4 : 0));
  return (w && variant_eligible(w, p)) ? w : v;
}
static int choose_variant_streamed(const ta_conv_launch& p) {
  if (p.uniform_k && (p.n_slabs >= 2 || p.prec == PREC_F16)) {
    if (variant_eligible(TA_CV_SPLIT_2x2, p)) {
      if (p.prec != PREC_F32 && p.k_split == 1) {
        // 128 x 256 tiles (8 consumer waves) stream 25 % fewer DMA bytes per FLOP and measure ~8 % faster per tile
        // pair, but a CU holds one workgroup either way: take them when they do not cost a round of the 256 CUs
        const int n_ct = p.coutp / 128;
        const int t1 = ((p.M + 127) / 128) * n_ct, t2 = ((p.M + 255) / 256) * n_ct;
        const int r1 = (t1 + 255) / 256, r2 = (t2 + 255) / 256;
        if (r2 * 184 < r1 * 100) return TA_CV_SPLIT_2x4;
      }
      return TA_CV_SPLIT_2x2;
    }
    if (variant_eligible(TA_CV_SPLIT_1x4, p)) {
      // 64-channel layers with a short K (18 slabs: conv1_2 of the pose network, 544 us per step) and many tiles: two workgroups
      // per CU on a 2-stage ring hide one tile's fixed cost under the other's loop (+6 ... 10 % on those shapes, tools/conv_bench.py)
      static const bool no_w2 = getenv("TA_CONV_NO_W2") != nullptr;
      static const int w2_tiles = getenv("TA_CONV_W2_TILES") ? atoi(getenv("TA_CONV_W2_TILES")) : 1024;     // tools: the threshold
      static const int w2_slabs = getenv("TA_CONV_W2_SLABS") ? atoi(getenv("TA_CONV_W2_SLABS")) : 18;
      if (!no_w2 && p.prec != PREC_F32 && p.n_slabs <= w2_slabs && (p.M + 255) / 256 * (p.coutp / 64) >= w2_tiles && variant_eligible(TA_CV_SPLIT_1x4_W2, p))
        return TA_CV_SPLIT_1x4_W2;
      return TA_CV_SPLIT_1x4;
    }
    if (variant_eligible(TA_CV_PIPE64, p)) return TA_CV_PIPE64;
  }
  return TA_CV_GENERIC;
}

template <int PREC>
static int launch_variant(ta_ctx* ctx, int v, const ta_conv_launch& p) {
  switch (v) {
    case TA_CV_GENERIC:
      if (p.coutp % 128 == 0) return launch_cfg<2, 2, 2, 2, PREC>(ctx, p);
      if (p.coutp % 64 == 0) return launch_cfg<1, 4, 2, 1, PREC>(ctx, p);
      return launch_cfg<1, 4, 1, 1, PREC>(ctx, p);
    case TA_CV_PIPE64:
      if constexpr (PREC != PREC_F32) {
        if (p.in_fmt != TA_FMT_F32) return launch_pipe<1, 4, 2, 1, PREC, 3, true>(ctx, p);
      }
      return launch_pipe<1, 4, 2, 1, PREC, 3, false>(ctx, p);
    case TA_CV_PIPE128: return launch_pipe<2, 2, 2, 2, PREC, 3, false>(ctx, p);
    case TA_CV_SPLIT_2x2: return launch_split<2, 2, 4, PREC, 3>(ctx, p);
    case TA_CV_SPLIT_2x2_P8: return launch_split<2, 2, 8, PREC, 3>(ctx, p);
    case TA_CV_SPLIT_2x4: return launch_split<2, 4, 4, PREC, 3>(ctx, p);
    case TA_CV_SPLIT_1x4: return launch_split<1, 4, 4, PREC, 3>(ctx, p);
    case TA_CV_SPLIT_1x4_W2: return launch_split<1, 4, 4, PREC, 2>(ctx, p);
    case TA_CV_SPLIT_2x2_W2: return launch_split<2, 2, 4, PREC, 2>(ctx, p);
    case TA_CV_WIN_2x2:
    case TA_CV_WIN_2x4:
    case TA_CV_WIN_1x4:
      if constexpr (PREC == PREC_F16X3 || PREC == PREC_F16X2) {
        if (v == TA_CV_WIN_2x2) return launch_win<2, 2, PREC, TA_WIN_PR_2x2>(ctx, p);
        if (v == TA_CV_WIN_2x4) return launch_win<2, 4, PREC, TA_WIN_PR_2x4>(ctx, p);
        return launch_win<1, 4, PREC, TA_WIN_PR_1x4>(ctx, p);
      }
      break;
  }
  return ta_fail(ctx, TA_E_INVALID, "conv: unknown kernel variant %d", v);
}

int ta_launch_conv(ta_ctx* ctx, const ta_conv_launch& p_in, double flops) {
  if (p_in.M <= 0) return TA_OK;
  static const int direct = getenv("TA_CONV_DIRECT_EPILOGUE") ? 1 : 0;      // A/B switch: accumulators straight to global
  static const int no_ksplit = getenv("TA_CONV_NO_KSPLIT") ? 1 : 0;         // A/B switch
  ta_conv_launch p = p_in;
  p.direct_epilogue = direct;
  p.probe = ctx->conv_probe;
  if (p.k_split < 1 || !p.partial || no_ksplit) p.k_split = 1;
  if (p.coutp % 32 != 0 || p.cout % 4 != 0) return ta_fail(ctx, TA_E_INVALID, "conv: bad cout padding");
  if (p.prec != PREC_F32 && p.prec != PREC_BF16X3 && p.prec != PREC_BF16 && p.prec != PREC_F16X3 && p.prec != PREC_F16 && p.prec != PREC_F16X2)
    return ta_fail(ctx, TA_E_INVALID, "conv: unknown precision mode %d", p.prec);
  if (p.in_fmt != TA_FMT_F32 && p.in_fmt != ta_split_fmt_of(p.prec))
    return ta_fail(ctx, TA_E_INVALID, "conv: input tensor format %d does not belong to precision mode %d", p.in_fmt, p.prec);
  p.range_flag = ctx->range_flag;
  int v = p.variant;
  if (v != TA_CV_AUTO) {
    if (!variant_eligible(v, p))
      return ta_fail(ctx, TA_E_INVALID, "conv: forced kernel variant %d cannot run this layer (cin-uniform %d, slabs %d, coutp %d, "
                     "input format %d, groups %d)", v, p.uniform_k, p.n_slabs, p.coutp, p.in_fmt, p.group_cout ? 1 : 0);
  } else {
    auto is_split_v = [](int x) { return x == TA_CV_SPLIT_2x2 || x == TA_CV_SPLIT_2x2_P8 || x == TA_CV_SPLIT_2x4 || x == TA_CV_SPLIT_1x4 || x == TA_CV_SPLIT_1x4_W2 || x == TA_CV_SPLIT_2x2_W2; };   // (the window kernels have no fused pool)
    const bool prefer = ctx->conv_force && variant_eligible(ctx->conv_force, p) && (!p.pool || is_split_v(ctx->conv_force));
    v = prefer ? ctx->conv_force : choose_variant(p);
    if (!variant_eligible(v, p)) return ta_fail(ctx, TA_E_INVALID, "conv: pre-split input reached a kernel that cannot read it");
  }
  const bool is_win = v == TA_CV_WIN_2x2 || v == TA_CV_WIN_2x4 || v == TA_CV_WIN_1x4;
  const bool is_split = v == TA_CV_SPLIT_2x2 || v == TA_CV_SPLIT_2x2_P8 || v == TA_CV_SPLIT_2x4 || v == TA_CV_SPLIT_1x4 || v == TA_CV_SPLIT_1x4_W2 || v == TA_CV_SPLIT_2x2_W2;
  if (!is_split || v == TA_CV_SPLIT_1x4_W2 || v == TA_CV_SPLIT_2x2_W2) p.k_split = 1;
  (void)is_win;
  if (p.pool) {                                      // only the split-role kernel's LDS-staged epilogue knows 2x2 windows
    if (!is_split || p.res || p.out2 || (p.out_ch & 7) || (p.M & 3) || (p.act != TA_ACT_RELU && p.act != TA_ACT_NONE))
      return ta_fail(ctx, TA_E_INVALID, "conv: the fused max-pool needs the split-role kernel and a plain epilogue (variant %d)", v);
    p.k_split = 1;
    p.direct_epilogue = 0;
  }                                              // only the split-role kernel knows K ranges
  ctx->conv_counts[v] += 1;
  ctx->cur_flops = flops;
  ta_prof_scope scope(ctx, 0, flops);
  switch (p.prec) {
    case PREC_F32: return launch_variant<PREC_F32>(ctx, v, p);
    case PREC_BF16X3: return launch_variant<PREC_BF16X3>(ctx, v, p);
    case PREC_F16X3: return launch_variant<PREC_F16X3>(ctx, v, p);
    case PREC_F16: return launch_variant<PREC_F16>(ctx, v, p);
    case PREC_F16X2: return launch_variant<PREC_F16X2>(ctx, v, p);
    default: return launch_variant<PREC_BF16>(ctx, v, p);
  }
}

extern "C" {

int ta_debug_conv_variant(ta_ctx* ctx, int variant) {
  if (!ctx || variant < 0 || variant >= TA_CV_COUNT) return TA_E_INVALID;
  ctx->conv_force = variant;
  return TA_OK;
}

int ta_debug_kernel_work(ta_ctx* ctx, char* csv, size_t capacity, int reset) {
  if (!ctx || (!csv && capacity)) return TA_E_INVALID;
  std::string out;
  for (const auto& kv : ctx->kernel_work) {
    char line[200];
    const auto t = ctx->kernel_ms.find(kv.first);
    snprintf(line, sizeof(line), "%s;%lld;%.6e;%.6f\n", kv.first.c_str(), (long long)kv.second.first, kv.second.second,
             t == ctx->kernel_ms.end() ? 0.0 : t->second);
    out += line;
  }
  if (reset) {
    ta_drain_profile(ctx);                           // pending events point at the keys: read them out first
    ctx->kernel_work.clear();
    ctx->kernel_ms.clear();
  }
  if (out.size() + 1 > capacity) return ta_fail(ctx, TA_E_CAPACITY, "kernel_work: %zu bytes needed", out.size() + 1);
  memcpy(csv, out.c_str(), out.size() + 1);
  return TA_OK;
}

int ta_debug_conv_counts(ta_ctx* ctx, int64_t* counts16, int reset) {
  if (!ctx) return TA_E_INVALID;
  for (int i = 0; i < TA_CV_COUNT; ++i) {
    if (counts16) counts16[i] = ctx->conv_counts[i];
    if (reset) ctx->conv_counts[i] = 0;
  }
  return TA_OK;
}

}  // extern "C"

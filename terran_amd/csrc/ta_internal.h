// Internal declarations shared by the translation units of libterran_amd.so.
// gfx950 only: no CUDA/other-arch paths anywhere in this library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/terran_amd.h"

// ---------------------------------------------------------------------------------------------
// Packed model blob (written by terran_amd/pack.py, parsed by net.cpp).  All little-endian PODs.
// ---------------------------------------------------------------------------------------------
#define TA_BLOB_MAGIC 0x314D4154u /* "TAM1" */

enum {
  TA_OP_CONV = 1,
  TA_OP_DWCONV = 2,
  TA_OP_MAXPOOL = 3,
  TA_OP_COPYCH = 4,
  TA_OP_RFSTEM = 5,   // RetinaFace front: uint8 frames -> conv3x3 s2 (3 -> 8) -> dw3x3 (8) -> 1x1 (8 -> 16), all + BN + ReLU, one kernel; cout == 32: + the next block (dw3x3 s2 (16) -> 1x1 (16 -> 32)) in the same kernel
  TA_OP_DWPW = 6      // depthwise 3x3 (stride 1 / 2) + BN + ReLU fused into the following 1x1 conv + BN + ReLU (exact-f32 MFMA)
};
enum { TA_ACT_NONE = 0, TA_ACT_RELU = 1, TA_ACT_PRELU = 2 };

struct ta_blob_header {
  uint32_t magic;
  uint32_t version;
  int32_t kind;        // TA_MODEL_*
  int32_t n_tensors;
  int32_t n_ops;
  int32_t input_tensor;   // tensor id the pre-processing kernel fills
  int32_t n_outputs;      // named outputs (ids in `outputs`)
  int32_t outputs[16];
  int64_t tensors_off, ops_off, weights_off, weights_bytes;
};

// Activation tensor: NHWC float32 with a physical zero halo of `halo` pixels on every side.
struct ta_tensor_desc {
  int32_t channels;   // C_total (multiple of 4)
  int32_t halo;
  int32_t alias_of;   // -1, or tensor id whose memory is viewed as (N,1,1,channels) (requires halo 0);
                      // -2: shape only, never materialised (the float input of a program whose first op reads the frames)
  int32_t fmt;        // TA_FMT_F32 / TA_FMT_SPLIT (act_format.h)
  int32_t unscale_off; // -1, or the byte offset (weights region) of [channels] floats 2^-a[c]: channel c of the tensor is STORED times
                       // 2^a[c] (f16x3 / f16 programs: keeps every channel's half-float words normal and away from 65504; chosen by the
                       // packer from the expected magnitudes, terran_amd/pack.py tensor_scales).  Consumers have 2^-a[c] folded into
                       // their weight columns, producers 2^a[co] into their per-channel epilogue vectors; debug taps
                       // (ta_model_read_tensor) and the pose post-processing multiply what they read with this vector
};

struct ta_op_desc {
  int32_t type;
  int32_t in, out;             // tensor ids
  int32_t in_ch_off, cin;      // input channel slice; cin multiple of 4
  int32_t out_ch_off, cout;    // output channel slice; cout multiple of 4
  int32_t coutp;               // cout padded to a multiple of 32 (packed weight rows)
  int32_t kh, kw, stride, pad;
  int32_t act;
  int32_t res, res_ch_off, res_up2;   // residual tensor (-1 none); res_up2: read residual at (y/2, x/2)
  int32_t out2, out2_ch_off;          // second output = out*scale2 + shift2 (-1 none)
  int32_t n_slabs;                    // K slabs of 32 floats (8 chunks of 4 channels)
  int32_t prec;                       // 0 = f32 MFMA, 1 = bf16x3 split (16 bits), 2 = bf16 (throughput), 3 = f16x3 split (22 bits), 4 = f16 (11 bits: tolerance mode), 5 = f16x2 (f16x3's tensors and weights, activations enter the products as their hi half: tolerance mode)
  int32_t groups;                     // grouped conv: `cin` is per group, group g reads channels in_ch_off + g*cin
  int32_t variant;                    // bits 0..7: 0 = automatic, else the TA_CV_* kernel variant this conv MUST run on (tests);
                                      // bits 8..15: K-split factor fixed by the packer for this layer (0 = the library's rule);
                                      // bit 16: `scale2_off` holds a border-class bias table [16][coutp] (3x3, stride 1, pad 1, no
                                      // second output): a per-channel affine of the INPUT is folded into the weights;
                                      // bits 17..18: lane -- 1 / 2 = the op runs on side stream 1 / 2 (ta_model_run_ops)
  int32_t pool;                       // 1: a 2x2 / 2 max-pool (floor) is fused into the epilogue; `out` has the pooled size
  int32_t wscale_log2;                // reserved (0): blob versions <= 7 kept one weight exponent per layer here
  int64_t w_off, bias_off, prelu_off, scale2_off, shift2_off;   // byte offsets in weights region, -1 none
  int64_t wus_off;                    // conv / dw+pw: [coutp] floats, the power of two the raw sums of output channel co are multiplied
                                      // with before the bias is added: 2^(a_out[co] - s[co]) -- s[co] the exponent row co of the packed
                                      // half-float weights carries (their lo halves stay normal; the input channels' activation exponents
                                      // are folded into the weight columns), a_out[co] the activation exponent of the channel written;
                                      // bias, border-class biases and the second output's affine are packed times 2^a_out[co] already.
                                      // All ones outside the f16x3 / f16 programs
  double macs_per_pixel;              // algorithmic MACs per output pixel (true, unpadded dims)
};

static_assert(sizeof(ta_blob_header) == 128 && sizeof(ta_tensor_desc) == 20 && sizeof(ta_op_desc) == 152,
              "blob layout is shared with terran_amd/pack.py (HEADER_DT / TENSOR_DT / OP_DT)");

// ---------------------------------------------------------------------------------------------
// Context
// ---------------------------------------------------------------------------------------------
struct ta_prof_class {
  double ms = 0;
  int64_t launches = 0;
  double work = 0;
};

struct ta_pending_event {
  hipEvent_t a, b;
  int klass;
  const std::string* kernel;   // the dense-conv kernel instance launched inside the scope (nullptr: none); a key of ta_ctx::kernel_work
};

struct ta_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  std::string err;
  bool profiling = false;
  ta_prof_class prof[4];
  std::vector<ta_pending_event> pending;
  std::vector<hipEvent_t> event_pool;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  // grow-only device scratch + pinned host staging
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  // released frame buffers, kept for the next batch of the same size: hipFree waits for the whole device (every
  // stream of the process), which serialises the per-GPU host threads once per resize otherwise
  std::mutex frame_cache_mu;                       // frames may be released from a GC thread
  std::vector<std::pair<size_t, void*>> frame_cache;
  size_t frame_cache_bytes = 0;
  int64_t pose_peaks = 0, pose_connections = 0;    // statistics of the last OpenPose grouping on this context
  float* pose_wphase = nullptr;                    // x8 bicubic phase weights on the device (uploaded once per context)
  // where the last grouping left its per-stage results in the scratch block (ta_openpose_debug_read); n = 0: none
  struct {
    int n = 0, maxp = 0;
    const int *peak_cnt = nullptr, *peak_yx = nullptr, *conn_cnt = nullptr, *conn_ij = nullptr;
    const float *peak_sc = nullptr, *conn_sc = nullptr;
    // images of that grouping that were re-run with larger lists (openpose_post.hip: op_rerun_image): their taps live in
    // `mem`, a device block of their own that stays until the next grouping on the context
    struct over_t {
      int img, maxp;
      void* mem;
      const int *peak_cnt, *peak_yx, *conn_cnt, *conn_ij;
      const float *peak_sc, *conn_sc;
    };
    std::vector<over_t> over;
  } pose_dbg;
  // side streams of the op programs (ta_op_desc.variant bits 17..18 = lane 1..2): independent branches of a graph --
  // RetinaFace's context modules + heads of the stride-32 / 16 levels, 4 small launches each -- run beside the main
  // stream's ops instead of in front of them (fork / join with events, ta_model_run_ops).  Created on first use.
  hipStream_t side_stream[2] = {nullptr, nullptr};
  hipEvent_t side_fork[2] = {nullptr, nullptr}, side_join[2] = {nullptr, nullptr};
  // conv kernel selection (ta_debug_conv_variant, or TA_CONV_PREFER for tools): 0 = automatic, else the TA_CV_* variant
  // every conv that variant CAN run is launched on (others stay automatic);
  // conv_counts[v] = launches per variant since the last ta_debug_conv_counts(reset)
  int conv_force = 0;
  int conv_probe = 0;                              // tools only (TA_CONV_PROBE): timing ablations of the split kernel
  int64_t conv_counts[16] = {0};
  // f16x3: raised (device side) by a conv epilogue that met |x| > 65504 while writing a half-split tensor; copied to
  // the pinned host word behind the results of a call (ta_range_enqueue) and turned into TA_E_RANGE (ta_range_check)
  int* range_flag = nullptr;                       // device: [0] the flag, [TA_AMAX_SLOT0 ..] the tools' amax slots (2 per op, TA_AMAX_OPS ops)
  int* range_flag_host = nullptr;
  const void* amax_owner = nullptr;                // tools: the ONE model of this context whose ops report into the amax slots (ta_model_debug_amax)
  // algorithmic FLOPs per dense-conv KERNEL INSTANCE ("conv_igemm_split<2,4,4,3,3>", ...) since the last reset
  // (ta_debug_kernel_work): joins a rocprofv3 per-kernel time table with the work each template instance did
  double cur_flops = 0;
  std::map<std::string, std::pair<int64_t, double>> kernel_work;
  std::map<std::string, double> kernel_ms;        // HIP-event time per instance, while profiling (the launches ta_prof_scope brackets)
  const std::string* cur_kernel = nullptr;        // instance of the launch in flight inside the current profiling scope
  void note_kernel(const char* name) {
    auto it = kernel_work.try_emplace(name, 0, 0.0).first;      // (node keys are stable: pending events keep the pointer)
    it->second.first += 1;
    it->second.second += cur_flops;
    cur_kernel = &it->first;
  }
};
void ta_pose_free_big(ta_ctx* ctx);  // frees pose_dbg.over
#define TA_AMAX_SLOT0 16
#define TA_AMAX_OPS 4096
int ta_range_enqueue(ta_ctx* ctx);   // async copy of the flag on the context's stream (before the call's final sync)
int ta_range_check(ta_ctx* ctx);     // after that sync: TA_OK, or TA_E_RANGE (and the flag is cleared for the next call)
int ta_range_finish(ta_ctx* ctx, int rc);   // end of a task entry point: TA_E_RANGE takes precedence over the post-processing's own result

// Conv kernel variants (ta_debug_conv_variant / ta_debug_conv_counts; include/terran_amd.h lists them)
enum {
  TA_CV_AUTO = 0,
  TA_CV_GENERIC = 1,      // conv_igemm<...>: K-offset table, any Cin, float32 activations, 2 LDS stages
  TA_CV_PIPE64 = 2,       // conv_igemm_pipe<1,4,2,1>: 64 cout x 128 px, every wave issues DMA + MFMA
  TA_CV_PIPE128 = 3,      // conv_igemm_pipe<2,2,2,2>: 128 x 128 (float32 activations only)
  TA_CV_SPLIT_2x2 = 4,    // conv_igemm_split<2,2,4>: 128 cout x 128 px, 4 consumer + 4 producer waves
  TA_CV_SPLIT_2x2_P8 = 5, // conv_igemm_split<2,2,8>: same tile, 8 producer waves
  TA_CV_SPLIT_2x4 = 6,    // conv_igemm_split<2,4,4>: 128 x 256, 8 consumer waves
  TA_CV_SPLIT_1x4 = 7,    // conv_igemm_split<1,4,4>: 64 cout x 256 px
  TA_CV_WIN_2x2 = 8,      // conv_igemm_win<2,2>: 128 x 128, the pixel operand of a channel block resident in LDS (stride-1 convs, >= 4 taps)
  TA_CV_WIN_2x4 = 9,      // conv_igemm_win<2,4>: 128 x 256
  TA_CV_WIN_1x4 = 10,     // conv_igemm_win<1,4>: 64 cout x 256 px
  TA_CV_SPLIT_1x4_W2 = 11,  // conv_igemm_split<1,4,4,.,2>: 64 x 256 on a 2-stage ring, TWO workgroups per CU (short-K layers)
  TA_CV_SPLIT_2x2_W2 = 12,  // conv_igemm_split<2,2,4,.,2>: 128 x 128, likewise
  TA_CV_COUNT = 16
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: set it once per (kernel, device), not once per process
#define TA_SET_LDS_ATTR(ctx, kern, bytes)                                                                         \
  do {                                                                                                            \
    static std::atomic<unsigned long long> _done{0};                                                              \
    const unsigned long long _bit = 1ull << ((ctx)->device & 63);                                                 \
    if (!(_done.load(std::memory_order_acquire) & _bit)) {                                                        \
      TA_HIP(ctx, hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
      _done.fetch_or(_bit, std::memory_order_release);                                                            \
    }                                                                                                             \
  } while (0)

int ta_fail(ta_ctx* ctx, int code, const char* fmt, ...);
// HIP's current device is per host thread: every ABI entry point binds the calling thread to the ctx's GPU.
static inline void ta_enter(const ta_ctx* ctx) {
  if (ctx) (void)hipSetDevice(ctx->device);
}
int ta_scratch(ta_ctx* ctx, size_t bytes, void** out);   // device scratch, valid until next call
int ta_pinned(ta_ctx* ctx, size_t bytes, void** out);    // pinned host staging

#define TA_HIP(ctx, expr)                                                                      \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return ta_fail((ctx), TA_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                     __FILE__, __LINE__);                                                      \
  } while (0)

#define TA_TRY(expr)            \
  do {                          \
    int _r = (expr);            \
    if (_r != TA_OK) return _r; \
  } while (0)

// profiling scope: records an event pair around a launch when ctx->profiling
struct ta_prof_scope {
  ta_ctx* ctx;
  int klass;
  double work;
  hipEvent_t a = nullptr, b = nullptr;
  ta_prof_scope(ta_ctx* c, int k, double w);
  ~ta_prof_scope();
};

void ta_drain_profile(ta_ctx* ctx);   // runtime.hip: waits for the stream and adds every pending event pair to the class / instance times

struct ta_frames {
  ta_ctx* ctx;
  int n, h, w;
  uint8_t* dev;
  size_t cap = 0;          // bytes allocated
};

// ---------------------------------------------------------------------------------------------
// Planned tensors / kernels
// ---------------------------------------------------------------------------------------------
struct ta_tensor {
  float* dev = nullptr;   // base of the padded allocation
  int n = 0, h = 0, w = 0, c = 0, halo = 0, fmt = 0;
  const float* unscale_dev = nullptr;   // [c] floats 2^-a[c] on the device, or nullptr: stored values = true values (ta_tensor_desc::unscale_off)
  const float* unscale_host = nullptr;  // the same vector on the host (owned by the model)
  bool owns = true;
  int hp() const { return h + 2 * halo; }
  int wp() const { return w + 2 * halo; }
  size_t elems() const { return (size_t)n * hp() * wp() * c; }            // allocation: 4 bytes per element in every format
  int cf() const { return fmt == 3 /* TA_FMT_F16 */ ? c / 2 : c; }           // float slots per pixel (what strides are counted in)
  // element offset of interior pixel (img, y, x), channel 0
  size_t off(int img, int y, int x) const { return (((size_t)img * hp() + y + halo) * wp() + x + halo) * cf(); }
};

struct ta_conv_launch {
  const float* in;
  const float* w;
  const int32_t* ktab;
  const float* bias;
  const float* prelu;
  float* out;
  const float* res;
  float* out2;
  const float* scale2;
  const float* shift2;
  int M, Ho, Wo, n_slabs, coutp, cout, act, stride, prec;
  int uniform_k, k_cblocks, k_w, k_h, in_ch_off;   // uniform_k: cin % 32 == 0, slabs walk (channel block, kx, ky) without a table
  // element strides / pixel offsets (channel offsets are separate: the split format is not linear in the channel)
  int in_img, in_row, in_pix, in_off0;
  int out_img, out_row, out_pix, out_off0, out_ch, out_fmt;
  int res_img, res_row, res_pix, res_off0, res_up2, res_ch, res_fmt;
  int o2_img, o2_row, o2_pix, o2_off0, o2_ch, o2_fmt;
  int in_fmt;
  int win_wp, win_img;                         // pixels per padded row / per padded image of the INPUT tensor (conv_igemm_win: patch rows)
  int direct_epilogue;                         // debug A/B: 1 = split kernel stores straight from the accumulators
  int group_cout, group_cin;                   // grouped conv: output channels / input channels per group (0 = dense)
  int k_split;                                 // > 1: K is cut in k_split ranges, one workgroup each; raw sums go to
  float* partial;                              //      partial[k][pixel][coutp] and splitk_reduce_kernel finishes the op
  int variant;                                 // TA_CV_* this launch must use (0 = choose); an ineligible one is an error
  // TA_OP_DWPW: the B operand is not DMA'd but computed -- depthwise 3x3 (+ bias, ReLU) of `in` at (y*dw_stride, x*dw_stride)
  const float* dw_w;                           // [9][dw_c]
  const float* dw_bias;                        // [dw_c]
  int dw_c, dw_stride;
  // pool = 1 (split-role kernel only): tile pixels are ordered quad by quad -- pixel t of the launch is position
  // (t >> 1 & 1, t & 1) of 2x2 window t >> 2, windows in raster order over the POOLED map (Ho x Wo here are the pooled
  // sizes, M = 4 N Ho Wo) -- and the epilogue stores the max of every window instead of the four pixels
  int pool;
  // filled by the launcher of the split-role kernel: reciprocals of the launch-uniform divisors, so that a workgroup's
  // set-up needs no integer division (each one is a ~300-cycle dependent instruction chain in front of the first DMA)
  float r_nct, r_tile_blocks, r_Wo, r_Ho, r_HoWo;
  int fast_div;                                // 1: every dividend of the set-up is < 2^24 (exact in float32)
  int cons_prio;                               // split-role kernels: s_setprio level of the consumer (MFMA) waves, 0..3 (0 = leave alone)
  int fast_drain;                              // 1: the lean epilogue applies (split-format tensors below 4 GB, cout % 8 == 0, no pool / K-split)
  int probe;                                   // tools only: bits 0..1: 1 = producers skip the pixel-row DMA after the ring is full,
                                               //             2 = no DMA at all after the ring is full (WRONG results);
                                               //             bit 2 (TA_CONV_LATE_B): slab 0's pixel-row DMAs after ALL addresses are computed
                                               // [bias .. bias + coutp) is followed by the per-channel un-scale vector wus[coutp] (ta_op_desc.wus_off
                                               // == bias_off + 4 coutp): one pointer for both keeps the kernel-argument block (SGPRs) small
  const float* bias9;                          // border-class biases [16][coutp] of a conv with a folded input affine (nullptr: none)
  int range_check;                             // 1: the program has half-float convs and some op reads this op's output -- every store is
                                               // range-checked whatever the op's own mode (a float32 tensor written by an exact-f32 op
                                               // may be split into half floats by its consumer)
  int amax_index;                              // tools (ta_model_debug_amax): -1, or the op's slot pair behind the flag word: atomicMax of the
                                               // bit pattern of the largest |x| stored (slot 2 i), of a dw+pw block's depthwise rows (2 i + 1)
  int* range_flag;                             // set to 1 by an epilogue that stores |x| > 65504, inf or NaN while range_check is on
};

// the pre-split activation format the conv kernels of arithmetic mode `prec` read (PREC_* of conv_igemm.hip)
static inline int ta_split_fmt_of(int prec) { return prec == 0 ? 0 /* TA_FMT_F32 */ : (prec == 4 ? 3 /* TA_FMT_F16 */ : ((prec == 3 || prec == 5) ? 2 /* TA_FMT_SPLIT16 */ : 1 /* TA_FMT_SPLIT */)); }

// K-splitting of a conv with a very long K and few output tiles (ArcFace's 25088 -> 512 FC: 784 slabs, 4..8 tiles of
// 128 x 128 at the batch sizes in use): K is cut in a FIXED number of ranges that depends on the layer only, never on
// the batch -- the summation order, and with it every output bit, must not change with the batch composition
// (sharding a batch over GPUs has to reproduce the unsharded result exactly).  Shared by the planner (workspace size)
// and the launcher.  1 = no split.
// `packed` = the factor the packer wrote for the op (ta_op_desc.variant bits 8..15; 0 = none): layers whose output is
// too small to fill the chip at any batch in use (ArcFace stage 4: 7 x 7 maps, 100 tiles at 64 crops) get a fixed 2.
static inline int ta_conv_ksplit(int coutp, int n_slabs, bool eligible, int packed = 0) {
  if (!eligible || coutp % 128) return 1;
  if (packed > 1) return packed <= n_slabs ? packed : 1;
  if (n_slabs < 512) return 1;
  return 32;
}

int ta_launch_conv(ta_ctx* ctx, const ta_conv_launch& p, double flops);
int ta_launch_dwpw(ta_ctx* ctx, const ta_conv_launch& p, double flops);
struct ta_frames;
// RetinaFace front kernel: frames (uint8 RGB) -> 16-channel float tensor at half resolution; 448 packed floats in HOST memory
int ta_launch_rfstem(ta_ctx* ctx, const uint8_t* frames_dev, int n, int h, int w, const float* weights_host, const float* w2_dev, const struct ta_tensor& out);

struct ta_dw_launch {
  const float* in;
  const float* w;      // [9][C]
  const float* bias;   // [C]
  float* out;
  int N, Ho, Wo, C, stride, relu;
  int in_img, in_row, in_pix, in_off0, in_fmt;
  int out_img, out_row, out_pix, out_off0, out_fmt;
};
int ta_launch_dwconv(ta_ctx* ctx, const ta_dw_launch& p);
int ta_launch_maxpool(ta_ctx* ctx, const ta_tensor& in, const ta_tensor& out, int n);   // n images (<= tensor capacity)
int ta_launch_copych(ta_ctx* ctx, const ta_tensor& in, int in_ch, const ta_tensor& out, int out_ch, int ch, int n);

// pre-processing: uint8 frames -> float NHWC(4) with halo
enum { TA_PRE_RETINAFACE = 1, TA_PRE_OPENPOSE = 2, TA_PRE_ARCFACE_CROPS = 3 };
int ta_launch_preprocess(ta_ctx* ctx, int mode, const uint8_t* src_dev, int n, int h, int w, const ta_tensor& dst);

// ---------------------------------------------------------------------------------------------
// Model
// ---------------------------------------------------------------------------------------------
// One activation plan = everything that depends on the input shape (N, H, W).
struct ta_plan {
  int n = 0, h = 0, w = 0;
  std::vector<ta_tensor> tensors;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  int32_t* ktab_dev = nullptr;
  std::vector<size_t> ktab_off;   // per op, element offset into ktab_dev
  float* splitk_ws = nullptr;     // workspace for K-split partial sums (inside the arena)
  uint64_t last_use = 0;
};

struct ta_model {
  ta_ctx* ctx = nullptr;
  int kind = 0;
  ta_blob_header hdr;
  std::vector<ta_tensor_desc> tdesc;
  std::vector<ta_op_desc> ops;
  char* weights_dev = nullptr;
  bool has_half_ops = false;                    // any conv / dw+pw op with prec 3 / 4 / 5: every store is range-checked (ta_conv_launch::range_check)
  bool amax_on = false;                         // tools (ta_model_debug_amax): ops report the largest |x| they store into the context's slots
  std::vector<char> tensor_read;                // per tensor: some op of the program reads it (results no op reads are not range-checked)
  std::vector<std::vector<float>> unscale_host; // per tensor: host copy of its un-scale vector (empty: none)
  std::vector<char> weights_host_small;         // host copy of the few weights that travel as kernel arguments (TA_OP_RFSTEM)
  // small LRU of plans (lists of differently-sized images alternate between a few shapes); `tensors`,
  // `ktab_dev`, `ktab_off` mirror the active plan
  std::vector<ta_plan*> plans;
  ta_plan* active = nullptr;
  uint64_t use_counter = 0;
  int plan_n = 0, plan_h = 0, plan_w = 0;      // plan_n = CAPACITY of the active plan (>= run_n)
  int run_n = 0;                                // images / crops of the current call: every launch covers run_n, not plan_n
  const uint8_t* input_u8 = nullptr;            // the frames of the current forward_frames call (TA_OP_RFSTEM reads them)
  std::vector<ta_tensor> tensors;
  int32_t* ktab_dev = nullptr;
  std::vector<size_t> ktab_off;
  float* splitk_ws = nullptr;
};

int ta_model_plan(ta_model* m, int n, int h, int w);
int ta_model_run_ops(ta_model* m);

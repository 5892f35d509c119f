/* terran_amd.h -- C ABI of the MI355X-native (gfx950) per-frame human-perception path.
 *
 * This is the drop-in boundary for Terran's three model wrappers: the functions below
 * are exactly what a binding for the reference's plugin classes
 *
 *     terran/face/detection/retinaface/wrapper.py:92-238   class RetinaFace  (.call)
 *     terran/face/recognition/arcface/wrapper.py:102-184   class ArcFace     (.call)
 *     terran/pose/openpose/wrapper.py:166-485              class OpenPose    (.call)
 *
 * (selected through the checkpoint registry, terran/checkpoint.py:29-103,213-245) calls
 * instead of running torch.nn graphs + torch/numpy post-processing.  `terran_amd/` holds the
 * ctypes binding and the Python mirror of those classes; INTEGRATION.md shows the registry
 * entries a Terran maintainer would add.
 *
 * Conventions
 *   - plain C: opaque handles, pointers and sizes only; no exceptions cross the ABI.
 *   - every function returns TA_OK (0) or a negative TA_E_* code; ta_last_error(ctx) holds the text.
 *   - all pointers are HOST memory unless the name ends in _dev.
 *   - variable-length results use caller-allocated arrays with a `capacity` (in objects) and a
 *     `required` out-value; TA_E_CAPACITY is returned (nothing truncated silently) when too small.
 *   - one ta_ctx per GPU; a ctx and everything created from it belong to ONE host thread.
 *   - there is NO CPU fallback: every entry point fails with TA_E_DEVICE when no gfx950 device
 *     is usable.
 */
#ifndef TERRAN_AMD_H
#define TERRAN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TA_OK 0
#define TA_E_INVALID (-1)   /* bad argument / malformed model blob            */
#define TA_E_DEVICE (-2)    /* HIP error, no device, or out of device memory  */
#define TA_E_CAPACITY (-3)  /* caller-provided result arrays are too small    */
#define TA_E_OVERFLOW (-4)  /* an addressing limit was hit (ta_openpose_run: more than 65535 peaks of ONE body part in one image) */
#define TA_E_RANGE (-5)     /* f16x3 / f16x2 / f16 arithmetic modes only: an activation left the half-float range (stored |x| > 65504, inf
                             * or NaN; tensors are stored times a pack-time power of two that puts the expected maximum near 2^10);
                             * no numbers are returned -- run the input on a model packed for f32 (or bf16x3).  Checked on every tensor
                             * some op READS; final float32 results no op reads (embeddings, detector heads) are not range-checked */

#define TA_MODEL_RETINAFACE 1
#define TA_MODEL_ARCFACE 2
#define TA_MODEL_OPENPOSE 3

typedef struct ta_ctx ta_ctx;       /* one per device: stream, scratch, error text           */
typedef struct ta_model ta_model;   /* packed weights + op program + per-shape activation plan */
typedef struct ta_frames ta_frames; /* a batch of uint8 RGB frames resident in HBM (N,H,W,3)   */

/* ---- context -------------------------------------------------------------------------- */
const char* ta_version(void);
int ta_device_count(void);
/* PCI address ("0000:c1:00.0") of a device: what the host side needs to find the NUMA node / local CPUs of a GPU in sysfs
 * (terran_amd/affinity.py binds a rank's or a lane's host threads and pinned staging to them).  capacity >= 16. */
int ta_device_pci_bus_id(int device_id, char* out, int capacity);
int ta_ctx_create(int device_id, ta_ctx** out);
void ta_ctx_destroy(ta_ctx* ctx);
const char* ta_last_error(const ta_ctx* ctx);
int ta_ctx_sync(ta_ctx* ctx);
/* Per-kernel-class HIP-event timing (bench.py `roofline`): enable, run, then read.
 * klass: 0 = implicit-GEMM conv, 1 = depthwise/pool/elementwise, 2 = pre-processing,
 * 3 = post-processing.  ms = summed event time, launches = #kernels, work = algorithmic
 * FLOP (klass 0) or bytes (others) summed over those launches. */
int ta_profile_enable(ta_ctx* ctx, int on);
int ta_profile_reset(ta_ctx* ctx);
int ta_profile_read(ta_ctx* ctx, int klass, double* ms, int64_t* launches, double* work);
/* Event pair on the ctx stream (whole-step timing). */
int ta_timer_start(ta_ctx* ctx);
int ta_timer_stop(ta_ctx* ctx, double* ms);

/* Pinned (page-locked) host memory for staging frame batches: H2D copies from it run at full PCIe rate and
 * overlap with kernels of other contexts (terran/io/video/reader.py:88-117 hands over pageable numpy batches). */
int ta_host_alloc(ta_ctx* ctx, size_t bytes, void** out);
void ta_host_free(ta_ctx* ctx, void* ptr);

/* ---- frames (replaces `torch.as_tensor(images, device=...)`,
 *      retinaface/wrapper.py:144, openpose/wrapper.py:121, arcface/wrapper.py:170) -------- */
int ta_frames_upload(ta_ctx* ctx, const uint8_t* nhwc_rgb, int n, int h, int w, ta_frames** out);
int ta_frames_alloc(ta_ctx* ctx, int n, int h, int w, ta_frames** out);
int ta_frames_shape(const ta_frames* f, int* n, int* h, int* w);
int ta_frames_download(const ta_frames* f, uint8_t* nhwc_rgb);
/* Releases the handle.  The device buffer is parked in its context (a few recent sizes, bounded in bytes) and handed
 * to the next batch of the same size: the steady state of a video loop performs no hipMalloc / hipFree (a hipFree
 * waits for every stream of the process).  Parked buffers are freed with the context. */
void ta_frames_free(ta_frames* f);
/* cv2.resize(..., INTER_LINEAR) semantics on the device
 * (face/detection/__init__.py:33-38, pose/openpose/wrapper.py:106-111). */
int ta_frames_resize(ta_ctx* ctx, const ta_frames* src, int dst_h, int dst_w, ta_frames** out);
/* Pillow Image.resize(size, resample=BICUBIC) semantics (antialiased separable convolution with
 * 22-bit fixed-point coefficients; arcface/wrapper.py:83-85, the no-landmark crop path). */
int ta_frames_resize_bicubic(ta_ctx* ctx, const ta_frames* src, int dst_h, int dst_w, ta_frames** out);
/* Zero-pad `src` image `src_index` into image `dst_index` of `dst` at (top, left)
 * (merge_in, face/detection/__init__.py:96-139 / pose/__init__.py:48-88). */
int ta_frames_paste(ta_ctx* ctx, const ta_frames* src, int src_index, ta_frames* dst, int dst_index,
                    int top, int left);

/* ---- models ---------------------------------------------------------------------------- */
/* `blob` is the packed model produced by terran_amd/pack.py from a Terran state_dict
 * (replaces load_model(): retinaface/wrapper.py:16-22, arcface/wrapper.py:13-19,
 * openpose/wrapper.py:27-36).  The blob is copied; the caller may free it. */
int ta_model_load(ta_ctx* ctx, int kind, const void* blob, size_t bytes, ta_model** out);
void ta_model_free(ta_model* m);
int ta_model_kind(const ta_model* m);

/* Debug taps for parity tests: run only the network on `frames` (RetinaFace / OpenPose) or on
 * uint8 BGR CHW crops (ArcFace), then read any tensor of the op program back as float32 NCHW.
 * `tensor` is a tensor id of the packed program; `ch_off/ch` select a channel slice. */
int ta_model_forward_frames(ta_model* m, const ta_frames* frames);
int ta_model_forward_crops(ta_model* m, const uint8_t* crops_nchw_bgr, int n);
int ta_model_tensor_shape(ta_model* m, int tensor, int* n, int* c, int* h, int* w);
int ta_model_read_tensor(ta_model* m, int tensor, int ch_off, int ch, float* dst_nchw);

/* f16x3 / f16 programs store every CHANNEL of every tensor times a power of two chosen at pack time (terran_amd/pack.py:
 * tensor_scales); ta_model_read_tensor divides it out.  ta_model_tensor_unscale copies the per-channel factors 2^-a[c]
 * (ones where nothing is scaled; capacity >= the tensor's channels).  ta_model_debug_amax (tools / tests: does the packer's
 * expectation hold?): enable != 0 starts (or restarts, zeroed) the collection of the largest |x| every conv / dw+pw op STORES
 * (in stored, i.e. scaled, units: the figure that must stay below 65504); with out != NULL the maxima collected so far are
 * copied out first: out[2 i] = op i's output, out[2 i + 1] = the depthwise intermediate of a dw+pw op; capacity >= 2 x the
 * number of ops (TA_E_CAPACITY otherwise).  enable == 2 only reads (the collection goes on), enable == 0 ends it. */
int ta_model_tensor_unscale(const ta_model* m, int tensor, float* out, int capacity);
int ta_model_debug_amax(ta_model* m, int enable, float* out, int capacity);
/* tools (tools/graph_probe.py): replays the op program of the last forward `reps` times as stream launches and as a hipGraph
 * captured from them.  out_ms[5]: GPU ms per replay (streams, graph), host enqueue ms per replay (streams, graph), capture +
 * instantiate ms.  Measurement aid for DESIGN.md section 4 ("why the programs are not hipGraphs"); no product path calls it. */
int ta_model_graph_probe(ta_model* m, int reps, double* out_ms);

/* ---- RetinaFace.call (retinaface/wrapper.py:133-238) ------------------------------------ */
/* frames: (N,H,W,3) uint8 RGB at network resolution.  Per image: threshold (>=), sort by
 * descending score (ties: ascending anchor index), greedy NMS (IoU > nms_thr suppressed).
 * Results are concatenated over images in order; counts[i] = detections of image i.
 * boxes (x1,y1,x2,y2), landmarks 5x(x,y), network-input pixels, float32. */
int ta_retinaface_run(ta_model* m, const ta_frames* frames, float score_thr, float nms_thr,
                      int capacity, int32_t* counts, float* boxes, float* landmarks, float* scores,
                      int32_t* required);
/* Post-processing only, on host head tensors in the reference layout (nine NCHW float32 arrays,
 * order stride 32,16,8 x cls_prob(4ch), bbox(8ch), landmark(20ch); model.py:304-316). */
int ta_retinaface_postprocess(ta_ctx* ctx, const float* const heads[9], int n, int h, int w,
                              float score_thr, float nms_thr, int capacity, int32_t* counts,
                              float* boxes, float* landmarks, float* scores, int32_t* required);

/* ---- ArcFace.call (arcface/wrapper.py:109-184) ------------------------------------------ */
/* Embed n pre-cropped faces, uint8 (n,3,112,112) BGR.  normalize != 0 applies the row-wise L2
 * normalisation of wrapper.py:176.  out: (n,512) float32. */
int ta_arcface_embed_crops(ta_model* m, const uint8_t* crops_nchw_bgr, int n, int normalize, float* out);
/* Align + embed on the device: face k is warped from frames[frame_index[k]] with the inverse
 * similarity inv_affine[k] (6 doubles, PIL Image.transform(AFFINE) convention, wrapper.py:61-69),
 * bilinear, fill 0, to 112x112 BGR.  crops_out (optional, may be NULL): (n,3,112,112) uint8. */
int ta_arcface_embed_faces(ta_model* m, const ta_frames* frames, const int32_t* frame_index,
                           const double* inv_affine, int n, int normalize, float* out,
                           uint8_t* crops_out);
/* The same for faces cut from SEVERAL resident frame batches in one launch (a video loop's embedder sees the faces of every
 * batch in flight: the network fills the chip at ~256 crops, one 32-frame batch brings ~64): face k comes from
 * frames[source_index[k]] (source_index NULL: all from frames[0]), image frame_index[k] of that batch.  All batches must live
 * on the model's device (any context).  A face's embedding does not depend on the launch it rides in. */
int ta_arcface_embed_faces_multi(ta_model* m, const ta_frames* const* frames, int n_sources, const int32_t* source_index,
                                 const int32_t* frame_index, const double* inv_affine, int n, int normalize,
                                 float* out, uint8_t* crops_out);
/* Cosine distance matrix 1 - a.b/(|a||b|) (examples/match.py:38): a (na,dim), b (nb,dim) -> (na,nb). */
int ta_cosine_distance(ta_ctx* ctx, const float* a, int na, const float* b, int nb, int dim, float* out);

/* ---- OpenPose.call (openpose/wrapper.py:182-485) ---------------------------------------- */
/* frames: uint8 RGB ALREADY at network resolution (the wrapper's resize is ta_frames_resize);
 * scale = short_side / min(H_orig, W_orig) maps keypoints back ((coord/scale) truncated).
 * keypoints: (M,18,3) int32 (x, y, present); scores: (M,) float64; counts[i] humans of image i.
 * No caps, like the reference (wrapper.py:235-262,335-366): the grouping kernels' fast path keeps 1024 peaks per part,
 * 8192 candidate pairs per limb and 192 people under assembly per image in LDS (~50x what a real frame produces); an
 * image that outgrows them -- a saturated heat-map plateau does -- is recomputed alone with lists in global memory sized
 * from its own counts, and network-resolution maps too large for the LDS staging (beyond ~90 x 160 cells) take the same
 * global-memory kernels for the whole batch.  Only > 65535 peaks of one part in one image fail (TA_E_OVERFLOW). */
int ta_openpose_run(ta_model* m, const ta_frames* frames, double scale, int capacity,
                    int32_t* counts, int32_t* keypoints, double* scores, int32_t* required);
/* Grouping only (x8 bicubic, peaks, PAF scoring, greedy matching, assembly) on host maps at
 * network-output resolution: pafs (N,38,h,w), heatmaps (N,19,h,w) float32 NCHW. */
int ta_openpose_group(ta_ctx* ctx, const float* pafs, const float* heatmaps, int n, int h, int w,
                      double scale, int capacity, int32_t* counts, int32_t* keypoints,
                      double* scores, int32_t* required);
/* Workload statistics of the last ta_openpose_run / ta_openpose_group on this context: heat-map peaks found over
 * all images and parts (wrapper.py:235-262) and limb connections accepted by the greedy matching (wrapper.py:335-366). */
int ta_openpose_last_stats(const ta_ctx* ctx, int64_t* peaks, int64_t* connections);

/* Debug taps of the last ta_openpose_run / ta_openpose_group on this context (valid until the next call on it), for
 * parity tests of the seams between the grouping stages: per image and part the peaks found (wrapper.py:235-262, rows
 * in the reference's row-major order), per image and limb the connections kept by the greedy matching
 * (wrapper.py:335-366, in acceptance order).  n must equal the batch of that call.
 *   peak_counts (n,18) int32; peaks_yx (n,18,cap_peaks,2) int32 (y,x in the x8 map); peak_scores (n,18,cap_peaks) f32
 *   conn_counts (n,19) int32, -1 = limb skipped because one of its parts has no peak (wrapper.py:293-296);
 *   conn_ij (n,19,cap_conn,2) int32 (index into the source / destination part's peak list); conn_scores (n,19,cap_conn) f32
 * Rows beyond cap_* are dropped (the counts still report the true numbers). */
int ta_openpose_debug_read(ta_ctx* ctx, int n, int cap_peaks, int32_t* peak_counts, int32_t* peaks_yx,
                           float* peak_scores, int cap_conn, int32_t* conn_counts, int32_t* conn_ij,
                           float* conn_scores);

/* x8 bicubic upsample alone (wrapper.py:214-223): maps (N,C,h,w) -> (N,C,8h,8w). */
int ta_bicubic_x8(ta_ctx* ctx, const float* maps, int n, int c, int h, int w, float* out);

/* ---- conv kernel selection (parity tests and tools) ------------------------------------------ */
/* Every dense conv / FC of the three networks runs on one of these implicit-GEMM kernels (terran_amd/csrc/conv_igemm.hip);
 * the library picks per layer (TA_CONV_AUTO).  ta_debug_conv_variant makes every following conv launch on this context
 * that the variant CAN run use it (layers it cannot run stay automatic); a packed model may also pin single convs to a
 * variant (ta_op_desc.variant, terran_amd/pack.py `variant=`), which fails with TA_E_INVALID when that kernel cannot run
 * the layer.  ta_debug_conv_counts reports (and optionally clears) the launches per variant since the last reset,
 * counts16[TA_CONV_*], so a parity test knows which kernels produced the numbers it compared (counts16[15]: how many
 * of those launches ran the split-role kernels' compile-time specialised epilogue rather than the generic one).
 * Under TA_CONV_AUTO a frame's result does not depend on the batch it arrives in (every kernel the automatic choice
 * can give one layer sums in the same order); a forced preference keeps the parity tolerances but not that bit-level
 * batch invariance (the symmetric-wave kernels pair K differently inside a slab and take no K-split). */
#define TA_CONV_AUTO 0
#define TA_CONV_GENERIC 1      /* K-offset table, any Cin, float32 activations                       */
#define TA_CONV_PIPE64 2       /* 64 cout x 128 px tiles, symmetric waves, 3-stage LDS ring          */
#define TA_CONV_PIPE128 3      /* 128 x 128 tiles, symmetric waves (float32 activations only)        */
#define TA_CONV_SPLIT_2x2 4    /* producer/consumer waves, 128 cout x 128 px                         */
#define TA_CONV_SPLIT_2x2_P8 5 /* ... with 8 producer waves                                          */
#define TA_CONV_SPLIT_2x4 6    /* producer/consumer waves, 128 cout x 256 px (8 consumer waves)      */
#define TA_CONV_SPLIT_1x4 7    /* producer/consumer waves, 64 cout x 256 px                          */
#define TA_CONV_WIN_2x2 8      /* ... 128 x 128 with the pixel operand of a channel block resident in LDS (stride-1 convs with >= 4 taps on
                                * pre-split half-float tensors; the K-split and the fused pool stay with the streaming kernels)  */
#define TA_CONV_WIN_2x4 9      /* ... 128 x 256                                                       */
#define TA_CONV_WIN_1x4 10     /* ... 64 cout x 256 px                                                */
#define TA_CONV_SPLIT_1x4_W2 11 /* producer/consumer waves, 64 x 256, 2-stage ring: TWO workgroups per CU (short-K layers)  */
#define TA_CONV_SPLIT_2x2_W2 12 /* ... 128 x 128                                                      */
int ta_debug_conv_variant(ta_ctx* ctx, int variant);
/* f16x3 mode: after ta_model_forward_* (the debug taps; the task entry points do this themselves), wait for the stream
 * and report whether an epilogue met |x| > 65504: TA_OK or TA_E_RANGE.  Clears the condition. */
int ta_debug_range_check(ta_ctx* ctx);
int ta_debug_conv_counts(ta_ctx* ctx, int64_t* counts16, int reset);
/* Algorithmic FLOPs per dense-conv kernel INSTANCE since the last reset, as text: one "name;launches;flops;ms" line per
 * template instance (ms: HIP-event time of the launches made while ta_ctx_profile was on) (the names rocprofv3 prints, without spaces: conv_igemm_split<2,4,4,3,3>).  Lets a per-kernel time
 * table from a profiler be turned into TFLOP/s per instance (profiles/summarize_round.py).  TA_E_CAPACITY if too small. */
int ta_debug_kernel_work(ta_ctx* ctx, char* csv, size_t capacity, int reset);

#ifdef __cplusplus
}
#endif
#endif /* TERRAN_AMD_H */

// Packed-model loader, per-shape activation planner and op-program executor.
//
// The op program (built by terran_amd/pack.py from a Terran state_dict) is a flat list of
// conv / depthwise / max-pool / channel-copy ops over halo-padded NHWC tensors.  Planning for a
// given (N,H,W) infers every tensor's spatial size, carves ONE zero-filled HBM arena (no buffer
// reuse: 288 GB of HBM makes liveness packing pointless and the zero halos must stay intact),
// and builds the per-conv K-offset tables.  Running is then a straight sequence of launches on
// the context's stream.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>

#include "act_format.h"
#include "ta_internal.h"

static int conv_out(int in, int k, int stride, int pad) { return (in + 2 * pad - k) / stride + 1; }

// A conv may be K-split when it runs on the split-role kernel (uniform K walk; f32 operands in f32 mode, pre-split
// operands in the bf16 modes) and its epilogue is plain (no residual, no second output).
static bool conv_ksplit_eligible(const ta_op_desc& op, int in_fmt) {
  const int kblk = in_fmt == TA_FMT_F16 ? 64 : 32;
  const bool uniform = (op.cin % kblk == 0) && (op.n_slabs == op.kh * op.kw * (op.cin / kblk));
  const bool kernel_ok = in_fmt == ta_split_fmt_of(op.prec);
  return uniform && kernel_ok && op.groups <= 1 && !op.pool;
}

#define TA_MAX_PLANS 4
#define TA_MAX_PLAN_BYTES ((size_t)96 << 30)

static void destroy_plan(ta_plan* pl) {
  if (!pl) return;
  if (pl->arena) (void)hipFree(pl->arena);
  if (pl->ktab_dev) (void)hipFree(pl->ktab_dev);
  delete pl;
}

static void activate(ta_model* m, ta_plan* pl) {
  m->active = pl;
  pl->last_use = ++m->use_counter;
  m->plan_n = pl->n;
  m->plan_h = pl->h;
  m->plan_w = pl->w;
  m->tensors = pl->tensors;
  m->ktab_dev = pl->ktab_dev;
  m->ktab_off = pl->ktab_off;
  m->splitk_ws = pl->splitk_ws;
}

static void free_plans(ta_model* m) {
  for (ta_plan* pl : m->plans) destroy_plan(pl);
  m->plans.clear();
  m->active = nullptr;
  m->plan_n = m->plan_h = m->plan_w = 0;
  m->tensors.clear();
  m->ktab_dev = nullptr;
}

// Batch capacity a plan is carved for.  Frame batches (RetinaFace / OpenPose) come in a few fixed sizes; the ArcFace
// batch is the number of faces in a frame batch and changes on almost every call of a video loop, so its plans are
// carved for a bucketed capacity (8, 32, then multiples of 64) and reused for every smaller count: no stream sync,
// hipMalloc / memset of a multi-GB arena or eviction hipFree per distinct face count.  Launches always cover exactly
// the n crops of the call (ta_model::run_n); the unused tail of the arena is never touched.
static int plan_capacity(int kind, int n) {
  if (kind != TA_MODEL_ARCFACE) return n;
  if (n <= 8) return 8;
  if (n <= 32) return 32;
  return (n + 63) / 64 * 64;
}

int ta_model_plan(ta_model* m, int n_run, int h, int w) {
  ta_ctx* ctx = m->ctx;
  if (n_run <= 0 || h <= 0 || w <= 0) return ta_fail(ctx, TA_E_INVALID, "plan: bad input shape %dx%dx%d", n_run, h, w);
  const int n = plan_capacity(m->kind, n_run);
  m->run_n = n_run;
  if (m->active && m->plan_n == n && m->plan_h == h && m->plan_w == w) {
    m->active->last_use = ++m->use_counter;
    return TA_OK;
  }
  for (ta_plan* pl : m->plans)
    if (pl->n == n && pl->h == h && pl->w == w) {
      activate(m, pl);
      return TA_OK;
    }
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // evict least-recently-used plans beyond the count / byte budget
  for (;;) {
    size_t bytes = 0;
    for (ta_plan* pl : m->plans) bytes += pl->arena_bytes;
    if (m->plans.size() < TA_MAX_PLANS && bytes <= TA_MAX_PLAN_BYTES) break;
    if (m->plans.empty()) break;
    size_t lru = 0;
    for (size_t i = 1; i < m->plans.size(); ++i)
      if (m->plans[i]->last_use < m->plans[lru]->last_use) lru = i;
    if (m->plans[lru] == m->active) m->active = nullptr;
    destroy_plan(m->plans[lru]);
    m->plans.erase(m->plans.begin() + lru);
  }
  ta_plan* np = new ta_plan();
  np->n = n;
  np->h = h;
  np->w = w;
  struct guard_t {
    ta_plan*& p;
    ~guard_t() { if (p) destroy_plan(p); }
  } guard{np};

  const int T = m->hdr.n_tensors;
  std::vector<ta_tensor> ts(T);
  std::vector<bool> set(T, false);
  for (int i = 0; i < T; ++i) {
    ts[i].c = m->tdesc[i].channels;
    ts[i].halo = m->tdesc[i].halo;
    ts[i].fmt = m->tdesc[i].fmt;
    if (m->tdesc[i].unscale_off >= 0) {
      ts[i].unscale_dev = (const float*)(m->weights_dev + m->tdesc[i].unscale_off);
      ts[i].unscale_host = m->unscale_host[i].data();
    }
    ts[i].n = n;
  }
  const int in_id = m->hdr.input_tensor;
  ts[in_id].h = h;
  ts[in_id].w = w;
  set[in_id] = true;

  auto resolve_alias = [&](int id) -> int {
    const int src = m->tdesc[id].alias_of;
    if (src == -2) ts[id].owns = false;            // shape only
    if (src < 0 || set[id]) return TA_OK;
    if (!set[src]) return ta_fail(ctx, TA_E_INVALID, "plan: alias tensor %d used before its source %d", id, src);
    if (ts[src].halo != 0 || (size_t)ts[src].h * ts[src].w * ts[src].c != (size_t)ts[id].c)
      return ta_fail(ctx, TA_E_INVALID, "plan: alias tensor %d does not match source %d", id, src);
    ts[id].h = ts[id].w = 1;
    ts[id].owns = false;
    set[id] = true;
    return TA_OK;
  };

  auto set_out = [&](int id, int oh, int ow) -> int {
    if (oh <= 0 || ow <= 0) return ta_fail(ctx, TA_E_INVALID, "plan: input %dx%d too small for the network", h, w);
    if (set[id]) {
      if (ts[id].h != oh || ts[id].w != ow)
        return ta_fail(ctx, TA_E_INVALID, "plan: tensor %d written with %dx%d and %dx%d", id, ts[id].h, ts[id].w, oh, ow);
      return TA_OK;
    }
    ts[id].h = oh;
    ts[id].w = ow;
    set[id] = true;
    return TA_OK;
  };

  for (size_t oi = 0; oi < m->ops.size(); ++oi) {
    const ta_op_desc& op = m->ops[oi];
    TA_TRY(resolve_alias(op.in));
    if (!set[op.in]) return ta_fail(ctx, TA_E_INVALID, "plan: op %zu reads unset tensor %d", oi, op.in);
    const ta_tensor& ti = ts[op.in];
    if (op.type != TA_OP_CONV && (ti.fmt == TA_FMT_F16 || ts[op.out].fmt == TA_FMT_F16))
      return ta_fail(ctx, TA_E_INVALID, "plan: op %zu: only convs read and write half-float tensors", oi);
    switch (op.type) {
      case TA_OP_CONV:
        if (op.groups > 1) {   // runs on the split-role kernel only: uniform K walk, 128-channel tiles inside one group
          const bool ok = op.cin % 32 == 0 && op.n_slabs == op.kh * op.kw * (op.cin / 32) && op.n_slabs >= 2 &&
                          op.cout % op.groups == 0 && (op.cout / op.groups) % 128 == 0 && op.cout == op.coutp &&
                          op.in_ch_off % 32 == 0 && op.in_ch_off + op.groups * op.cin <= ti.c &&
                          ti.fmt == ta_split_fmt_of(op.prec);
          if (!ok) return ta_fail(ctx, TA_E_INVALID, "plan: op %zu: unsupported grouped convolution", oi);
        }
        [[fallthrough]];
      case TA_OP_DWCONV:
        if (ti.halo < op.pad) return ta_fail(ctx, TA_E_INVALID, "plan: op %zu needs halo %d, tensor has %d", oi, op.pad, ti.halo);
        if (op.type == TA_OP_CONV && op.pool) {      // 2x2 / 2 max-pool (floor) in the conv's epilogue
          if (op.out2 >= 0 || op.res >= 0 || op.groups > 1 || op.cin % 32 || op.coutp % 64)
            return ta_fail(ctx, TA_E_INVALID, "plan: op %zu: unsupported conv + max-pool fusion", oi);
          TA_TRY(set_out(op.out, conv_out(ti.h, op.kh, op.stride, op.pad) / 2, conv_out(ti.w, op.kw, op.stride, op.pad) / 2));
          break;
        }
        TA_TRY(set_out(op.out, conv_out(ti.h, op.kh, op.stride, op.pad), conv_out(ti.w, op.kw, op.stride, op.pad)));
        if (op.out2 >= 0) TA_TRY(set_out(op.out2, ts[op.out].h, ts[op.out].w));
        break;
      case TA_OP_MAXPOOL:
        TA_TRY(set_out(op.out, ti.h / 2, ti.w / 2));
        break;
      case TA_OP_RFSTEM:
        if (oi != 0 || op.in != in_id || m->tdesc[op.in].alias_of != -2 || op.w_off < 0)
          return ta_fail(ctx, TA_E_INVALID, "plan: the RetinaFace front op must be op 0 on a shape-only input tensor");
        if (op.cout == 32)                           // fused with the next block (dw3x3 s2 -> 1x1 16 -> 32): quarter resolution
          TA_TRY(set_out(op.out, ((ti.h + 1) / 2 + 1) / 2, ((ti.w + 1) / 2 + 1) / 2));
        else
          TA_TRY(set_out(op.out, (ti.h + 1) / 2, (ti.w + 1) / 2));
        break;
      case TA_OP_DWPW:
        if (ti.halo < 1) return ta_fail(ctx, TA_E_INVALID, "plan: op %zu (dw+pw) needs an input halo", oi);
        if (ti.fmt != TA_FMT_F32 || op.in_ch_off || op.scale2_off < 0 || op.shift2_off < 0 || op.cin > ti.c)
          return ta_fail(ctx, TA_E_INVALID, "plan: op %zu: unsupported dw+pw block", oi);
        TA_TRY(set_out(op.out, conv_out(ti.h, 3, op.stride, 1), conv_out(ti.w, 3, op.stride, 1)));
        break;

      case TA_OP_COPYCH:
        TA_TRY(set_out(op.out, ti.h, ti.w));
        break;
      default:
        return ta_fail(ctx, TA_E_INVALID, "plan: unknown op type %d", op.type);
    }
  }
  for (int i = 0; i < T; ++i) TA_TRY(resolve_alias(i));
  for (int i = 0; i < T; ++i)
    if ((ts[i].fmt != TA_FMT_F32 && ts[i].c % 32) || (ts[i].fmt == TA_FMT_F16 && ts[i].c % 64) || ts[i].fmt < 0 || ts[i].fmt > TA_FMT_F16)
      return ta_fail(ctx, TA_E_INVALID, "plan: tensor %d: format %d with %d channels", i, ts[i].fmt, ts[i].c);

  // carve the arena
  size_t total = 0;
  std::vector<size_t> offs(T, 0);
  for (int i = 0; i < T; ++i) {
    if (!set[i] || !ts[i].owns) continue;
    const size_t bytes = ts[i].elems() * sizeof(float);
    if (bytes >= ((size_t)1 << 32)) return ta_fail(ctx, TA_E_INVALID, "plan: tensor %d exceeds 4 GiB; split the batch", i);
    offs[i] = total;
    total += (bytes + 255) & ~(size_t)255;
  }
  // workspace of the K-split convs (see ta_conv_ksplit): the largest partial[k][pixel][coutp] any op needs
  size_t ws_bytes = 0;
  for (size_t oi = 0; oi < m->ops.size(); ++oi) {
    const ta_op_desc& op = m->ops[oi];
    if (op.type != TA_OP_CONV) continue;
    const int M = ts[op.out].n * ts[op.out].h * ts[op.out].w;
    const int ks = ta_conv_ksplit(op.coutp, op.n_slabs, conv_ksplit_eligible(op, ts[op.in].fmt), (op.variant >> 8) & 255);
    if (ks > 1) ws_bytes = std::max(ws_bytes, (size_t)ks * M * op.coutp * sizeof(float));
  }
  const size_t ws_off = total;
  total += (ws_bytes + 255) & ~(size_t)255;
  hipError_t e = hipMalloc((void**)&np->arena, total ? total : 256);
  if (e != hipSuccess) return ta_fail(ctx, TA_E_DEVICE, "plan: hipMalloc(%zu) failed: %s", total, hipGetErrorString(e));
  np->arena_bytes = total;
  TA_HIP(ctx, hipMemsetAsync(np->arena, 0, total ? total : 256, ctx->stream));
  for (int i = 0; i < T; ++i)
    if (set[i] && ts[i].owns) ts[i].dev = (float*)(np->arena + offs[i]);
  for (int i = 0; i < T; ++i)
    if (set[i] && !ts[i].owns) ts[i].dev = m->tdesc[i].alias_of >= 0 ? ts[m->tdesc[i].alias_of].dev : nullptr;
  np->splitk_ws = ws_bytes ? (float*)(np->arena + ws_off) : nullptr;

  // K-offset tables
  std::vector<int32_t> ktab;
  np->ktab_off.assign(m->ops.size(), 0);
  for (size_t oi = 0; oi < m->ops.size(); ++oi) {
    const ta_op_desc& op = m->ops[oi];
    if (op.type != TA_OP_CONV) continue;
    const ta_tensor& ti = ts[op.in];
    np->ktab_off[oi] = ktab.size();
    if (ti.fmt == TA_FMT_F16) continue;          // split-role kernels only: uniform K walk, no table
    const int cpt = op.cin / 4;
    const int nq = op.kh * op.kw * cpt;
    if (nq > op.n_slabs * 8) return ta_fail(ctx, TA_E_INVALID, "plan: op %zu has too few K slabs", oi);
    for (int q = 0; q < op.n_slabs * 8; ++q) {
      int32_t off = 0;
      if (q < nq) {
        const int tap = q / cpt, ch = (q % cpt) * 4;
        const int ky = tap / op.kw, kx = tap % op.kw;
        off = (int32_t)((((size_t)ky * ti.wp() + kx) * ti.c + op.in_ch_off + ch) * sizeof(float));
      }
      ktab.push_back(off);
    }
  }
  if (!ktab.empty()) {
    TA_HIP(ctx, hipMalloc((void**)&np->ktab_dev, ktab.size() * sizeof(int32_t)));
    TA_HIP(ctx, hipMemcpy(np->ktab_dev, ktab.data(), ktab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  np->tensors.swap(ts);
  ta_plan* done = np;
  np = nullptr;                 // disarm the guard
  m->plans.push_back(done);
  activate(m, done);
  return TA_OK;
}

static const float* wptr(const ta_model* m, int64_t off) { return off < 0 ? nullptr : (const float*)(m->weights_dev + off); }

// tools only (TA_PROFILE_OPS=1): a HIP event pair around every op, then one table per forward on stderr
static void print_op_profile(ta_model* m, std::vector<hipEvent_t>& ev) {
  (void)hipStreamSynchronize(m->ctx->stream);
  double tot_ms = 0, tot_fl = 0;
  for (size_t oi = 0; oi < m->ops.size(); ++oi) {
    const ta_op_desc& op = m->ops[oi];
    const ta_tensor& to = m->tensors[op.out];
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ev[2 * oi], ev[2 * oi + 1]);
    const ta_tensor& ti = m->tensors[op.in];
    // a conv with a fused 2x2 max-pool computes every pixel of the UNPOOLED map: count those, not the pooled output
    const double M = op.type == TA_OP_CONV && op.pool
                         ? (double)m->run_n * conv_out(ti.h, op.kh, op.stride, op.pad) * conv_out(ti.w, op.kw, op.stride, op.pad)
                         : (double)m->run_n * to.h * to.w;
    const double fl = (op.type == TA_OP_CONV || op.type == TA_OP_DWPW) ? 2.0 * op.macs_per_pixel * M : 0.0;
    tot_ms += ms;
    tot_fl += fl;
    fprintf(stderr, "op %3zu type %d k%dx%d s%d g%d cin %4d cout %4d  %3dx%-3d M %8.0f slabs %4d  %8.1f us %7.1f TF\n", oi, op.type, op.kh,
            op.kw, op.stride, op.groups, op.cin, op.cout, to.h, to.w, M, op.n_slabs, ms * 1e3, ms > 0 ? fl / (ms * 1e-3) / 1e12 : 0.0);
  }
  fprintf(stderr, "model kind %d n %d: %.3f ms in ops, %.1f GFLOP, %.1f TF\n", m->kind, m->run_n, tot_ms, tot_fl / 1e9, tot_fl / (tot_ms * 1e-3) / 1e12);
  for (auto e : ev) (void)hipEventDestroy(e);
}

int ta_model_run_ops(ta_model* m) {
  ta_ctx* ctx = m->ctx;
  static const bool prof_ops = getenv("TA_PROFILE_OPS") != nullptr;
  std::vector<hipEvent_t> ev;
  if (prof_ops) {
    ev.resize(2 * m->ops.size());
    for (auto& e : ev) (void)hipEventCreate(&e);
  }
  struct fin_t {
    ta_model* m;
    std::vector<hipEvent_t>& ev;
    bool on;
    ~fin_t() { if (on) print_op_profile(m, ev); }
  } fin{m, ev, prof_ops};
  // lanes: an op with lane L != 0 goes to side stream L - 1, which first waits for everything queued on the main stream
  // so far (the packer places a branch right behind the op that produces its input); the main stream waits for the
  // branches at the end of the program.  The loader checked that nothing outside a branch touches what it writes.
  static const bool no_lanes = getenv("TA_NO_LANES") != nullptr;          // A/B switch
  const bool lanes_on = !prof_ops && !ctx->profiling && !no_lanes;        // profiles time a serial program
  struct lanes_t {
    ta_ctx* ctx;
    hipStream_t main;
    bool started[2] = {false, false};
    ~lanes_t() {                                                           // also on an error return
      ctx->stream = main;
      for (int i = 0; i < 2; ++i)
        if (started[i]) {
          (void)hipEventRecord(ctx->side_join[i], ctx->side_stream[i]);
          (void)hipStreamWaitEvent(main, ctx->side_join[i], 0);
        }
    }
  } lanes{ctx, ctx->stream};
  for (size_t oi = 0; oi < m->ops.size(); ++oi) {
    const ta_op_desc& op = m->ops[oi];
    const ta_tensor& ti = m->tensors[op.in];
    const ta_tensor& to = m->tensors[op.out];
    const int lane = lanes_on ? (op.variant >> 17) & 3 : 0;
    ctx->stream = lanes.main;
    if (lane) {
      const int li = lane - 1;
      if (!ctx->side_stream[li]) {
        TA_HIP(ctx, hipStreamCreateWithFlags(&ctx->side_stream[li], hipStreamNonBlocking));
        TA_HIP(ctx, hipEventCreateWithFlags(&ctx->side_fork[li], hipEventDisableTiming));
        TA_HIP(ctx, hipEventCreateWithFlags(&ctx->side_join[li], hipEventDisableTiming));
      }
      if (!lanes.started[li]) {
        TA_HIP(ctx, hipEventRecord(ctx->side_fork[li], lanes.main));
        TA_HIP(ctx, hipStreamWaitEvent(ctx->side_stream[li], ctx->side_fork[li], 0));
        lanes.started[li] = true;
      }
      ctx->stream = ctx->side_stream[li];
    }
    struct evp_t {
      hipEvent_t e;
      hipStream_t s;
      bool on;
      ~evp_t() { if (on) (void)hipEventRecord(e, s); }
    } evp{prof_ops ? ev[2 * oi + 1] : nullptr, ctx->stream, prof_ops};
    if (prof_ops) (void)hipEventRecord(ev[2 * oi], ctx->stream);
    switch (op.type) {
      case TA_OP_CONV: {
        ta_conv_launch p;
        memset(&p, 0, sizeof(p));
        p.in = ti.dev;
        p.w = wptr(m, op.w_off);
        p.ktab = m->ktab_dev + m->ktab_off[oi];
        p.bias = wptr(m, op.bias_off);
        p.prelu = wptr(m, op.prelu_off);
        p.out = to.dev;
        p.M = m->run_n * to.h * to.w;
        p.Ho = to.h;
        p.Wo = to.w;
        p.n_slabs = op.n_slabs;
        p.coutp = op.coutp;
        p.cout = op.cout;
        p.act = op.act;
        p.stride = op.stride;
        p.prec = op.prec;
        // channels per K slab: 32 (float32 and the hi | lo formats: 128 bytes per pixel and slab), 64 in TA_FMT_F16
        const int kblk = ti.fmt == TA_FMT_F16 ? 64 : 32;
        p.uniform_k = (op.cin % kblk == 0) && (op.n_slabs == op.kh * op.kw * (op.cin / kblk));
        p.k_cblocks = op.cin / kblk;
        if (ti.fmt == TA_FMT_F16 && (!p.uniform_k || op.in_ch_off || op.groups > 1 || op.prec != 4))
          return ta_fail(ctx, TA_E_INVALID, "op %zu: a half-float tensor feeds whole-tensor convs of the f16 mode only", oi);
        p.k_w = op.kw;
        p.k_h = op.kh;
        p.in_ch_off = op.in_ch_off;
        p.in_img = (int)((size_t)ti.hp() * ti.wp() * ti.cf());
        p.in_row = ti.wp() * ti.cf();
        p.in_pix = ti.cf();
        p.win_wp = ti.wp();
        p.win_img = ti.hp() * ti.wp();
        p.in_off0 = (int)(((size_t)(ti.halo - op.pad) * ti.wp() + (ti.halo - op.pad)) * ti.cf());
        p.out_img = (int)((size_t)to.hp() * to.wp() * to.cf());
        p.out_row = to.wp() * to.cf();
        p.out_pix = to.cf();
        p.out_off0 = (int)to.off(0, 0, 0);
        p.out_ch = op.out_ch_off;
        p.out_fmt = to.fmt;
        p.in_fmt = ti.fmt;
        if (op.res >= 0) {
          const ta_tensor& tr = m->tensors[op.res];
          p.res = tr.dev;
          p.res_img = (int)((size_t)tr.hp() * tr.wp() * tr.cf());
          p.res_row = tr.wp() * tr.cf();
          p.res_pix = tr.cf();
          p.res_off0 = (int)tr.off(0, 0, 0);
          p.res_ch = op.res_ch_off;
          p.res_fmt = tr.fmt;
          p.res_up2 = op.res_up2;
        }
        if (op.out2 >= 0) {
          const ta_tensor& t2 = m->tensors[op.out2];
          p.out2 = t2.dev;
          p.scale2 = wptr(m, op.scale2_off);
          p.shift2 = wptr(m, op.shift2_off);
          p.o2_img = (int)((size_t)t2.hp() * t2.wp() * t2.cf());
          p.o2_row = t2.wp() * t2.cf();
          p.o2_pix = t2.cf();
          p.o2_off0 = (int)t2.off(0, 0, 0);
          p.o2_ch = op.out2_ch_off;
          p.o2_fmt = t2.fmt;
        }
        if (op.groups > 1) {
          p.group_cout = op.cout / op.groups;
          p.group_cin = op.cin;
        }
        p.variant = op.variant & 255;
        if ((op.variant >> 16) & 1) p.bias9 = wptr(m, op.scale2_off);
        // a tensor no op reads is a float32 RESULT (embeddings, detector heads): nothing splits it into half floats, whatever it holds
        p.range_check = (m->has_half_ops && (m->tensor_read[op.out] || (op.out2 >= 0 && m->tensor_read[op.out2]))) ? 1 : 0;
        p.amax_index = (m->amax_on && oi < TA_AMAX_OPS) ? (int)oi : -1;
        double flops = 2.0 * op.macs_per_pixel * (double)p.M;
        if (op.pool) {
          p.pool = 1;
          p.M = m->run_n * to.h * to.w * 4;            // the four pixels of every 2x2 window
          flops = 2.0 * op.macs_per_pixel * (double)m->run_n * conv_out(ti.h, op.kh, op.stride, op.pad) *
                  conv_out(ti.w, op.kw, op.stride, op.pad);   // algorithmic: the whole conv output, odd edge included
        }
        p.k_split = m->splitk_ws ? ta_conv_ksplit(p.coutp, p.n_slabs, conv_ksplit_eligible(op, ti.fmt), (op.variant >> 8) & 255) : 1;
        p.partial = m->splitk_ws;
        if (lane) p.k_split = 1;                       // the K-split workspace belongs to the main stream
        TA_TRY(ta_launch_conv(ctx, p, flops));
        break;
      }
      case TA_OP_RFSTEM:
        TA_TRY(ta_launch_rfstem(ctx, m->input_u8, m->run_n, ti.h, ti.w, (const float*)m->weights_host_small.data(),
                                op.cout == 32 ? wptr(m, op.w_off) + 448 : nullptr, to));
        break;
      case TA_OP_DWPW: {
        ta_conv_launch p;
        memset(&p, 0, sizeof(p));
        p.in = ti.dev;
        p.w = wptr(m, op.w_off);
        p.bias = wptr(m, op.bias_off);
        p.dw_w = wptr(m, op.scale2_off);          // [9][cin]   (the op reuses the second-output fields for the depthwise part)
        p.dw_bias = wptr(m, op.shift2_off);       // [cin]
        p.dw_c = op.cin;
        p.dw_stride = op.stride;
        p.out = to.dev;
        p.M = m->run_n * to.h * to.w;
        p.Ho = to.h;
        p.Wo = to.w;
        p.n_slabs = op.n_slabs;
        p.coutp = op.coutp;
        p.cout = op.cout;
        p.act = op.act;
        p.stride = 1;
        p.prec = op.prec;
        p.range_check = (m->has_half_ops && m->tensor_read[op.out]) ? 1 : 0;
        p.amax_index = (m->amax_on && oi < TA_AMAX_OPS) ? (int)oi : -1;
        p.in_img = (int)((size_t)ti.hp() * ti.wp() * ti.c);
        p.in_row = ti.wp() * ti.c;
        p.in_pix = ti.c;
        p.in_off0 = (int)(((size_t)(ti.halo - 1) * ti.wp() + (ti.halo - 1)) * ti.c);
        p.in_fmt = ti.fmt;
        p.out_img = (int)((size_t)to.hp() * to.wp() * to.c);
        p.out_row = to.wp() * to.c;
        p.out_pix = to.c;
        p.out_off0 = (int)to.off(0, 0, 0);
        p.out_ch = op.out_ch_off;
        p.out_fmt = to.fmt;
        TA_TRY(ta_launch_dwpw(ctx, p, 2.0 * op.macs_per_pixel * (double)p.M));
        break;
      }
      case TA_OP_DWCONV: {
        ta_dw_launch p;
        memset(&p, 0, sizeof(p));
        p.in = ti.dev;
        p.w = wptr(m, op.w_off);
        p.bias = wptr(m, op.bias_off);
        p.out = to.dev;
        p.N = m->run_n;
        p.Ho = to.h;
        p.Wo = to.w;
        p.C = op.cin;
        p.stride = op.stride;
        p.relu = op.act == TA_ACT_RELU;
        p.in_img = (int)((size_t)ti.hp() * ti.wp() * ti.c);
        p.in_row = ti.wp() * ti.c;
        p.in_pix = ti.c;
        if (op.in_ch_off || op.out_ch_off) return ta_fail(ctx, TA_E_INVALID, "depthwise conv on a channel slice is not supported");
        p.in_off0 = (int)(((size_t)(ti.halo - op.pad) * ti.wp() + (ti.halo - op.pad)) * ti.c);
        p.in_fmt = ti.fmt;
        p.out_img = (int)((size_t)to.hp() * to.wp() * to.c);
        p.out_row = to.wp() * to.c;
        p.out_pix = to.c;
        p.out_off0 = (int)to.off(0, 0, 0);
        p.out_fmt = to.fmt;
        TA_TRY(ta_launch_dwconv(ctx, p));
        break;
      }
      case TA_OP_MAXPOOL:
        TA_TRY(ta_launch_maxpool(ctx, ti, to, m->run_n));
        break;
      case TA_OP_COPYCH:
        TA_TRY(ta_launch_copych(ctx, ti, op.in_ch_off, to, op.out_ch_off, op.cin, m->run_n));
        break;
    }
  }
  return TA_OK;
}

extern "C" {

int ta_model_load(ta_ctx* ctx, int kind, const void* blob, size_t bytes, ta_model** out) {
  ta_enter(ctx);
  if (!ctx || !blob || !out) return TA_E_INVALID;
  *out = nullptr;
  if (bytes < sizeof(ta_blob_header)) return ta_fail(ctx, TA_E_INVALID, "model blob too small");
  ta_blob_header h;
  memcpy(&h, blob, sizeof(h));
  if (h.magic != TA_BLOB_MAGIC || h.version != 9) return ta_fail(ctx, TA_E_INVALID, "model blob: bad magic/version (this library reads version 9)");
  if (h.kind != kind) return ta_fail(ctx, TA_E_INVALID, "model blob is kind %d, expected %d", h.kind, kind);
  if (h.n_tensors <= 0 || h.n_ops <= 0 || h.n_outputs < 0 || h.n_outputs > 16 || h.input_tensor < 0 ||
      h.input_tensor >= h.n_tensors)
    return ta_fail(ctx, TA_E_INVALID, "model blob: bad counts");
  const size_t t_end = (size_t)h.tensors_off + (size_t)h.n_tensors * sizeof(ta_tensor_desc);
  const size_t o_end = (size_t)h.ops_off + (size_t)h.n_ops * sizeof(ta_op_desc);
  const size_t w_end = (size_t)h.weights_off + (size_t)h.weights_bytes;
  if (t_end > bytes || o_end > bytes || w_end > bytes) return ta_fail(ctx, TA_E_INVALID, "model blob: truncated");
  ta_model* m = new ta_model();
  m->ctx = ctx;
  m->kind = kind;
  m->hdr = h;
  m->tdesc.resize(h.n_tensors);
  m->ops.resize(h.n_ops);
  memcpy(m->tdesc.data(), (const char*)blob + h.tensors_off, h.n_tensors * sizeof(ta_tensor_desc));
  memcpy(m->ops.data(), (const char*)blob + h.ops_off, h.n_ops * sizeof(ta_op_desc));
  for (auto& op : m->ops) {
    auto bad_t = [&](int t) { return t < 0 || t >= h.n_tensors; };
    auto bad_w = [&](int64_t off, size_t need) { return off >= 0 && (size_t)off + need > (size_t)h.weights_bytes; };
    bool bad = bad_t(op.in) || bad_t(op.out) || (op.res >= 0 && bad_t(op.res)) || (op.out2 >= 0 && bad_t(op.out2));
    if (op.type == TA_OP_CONV) {
      bad = bad || op.w_off < 0 || op.bias_off < 0 || op.cin % 4 || op.cout % 4 || op.coutp % 32 || op.n_slabs <= 0 ||
            op.stride <= 0 || op.wus_off != op.bias_off + 4 * (int64_t)op.coutp || bad_w(op.wus_off, (size_t)op.coutp * 4) || op.prec < 0 || op.prec > 5 || bad_w(op.w_off, (size_t)op.n_slabs * op.coutp * 128) ||
            bad_w(op.bias_off, (size_t)op.coutp * 4) || bad_w(op.prelu_off, (size_t)op.coutp * 4) ||
            bad_w(op.scale2_off, (size_t)op.coutp * 4) || bad_w(op.shift2_off, (size_t)op.coutp * 4) ||
            (op.act == TA_ACT_PRELU && op.prelu_off < 0) || (op.out2 >= 0 && (op.scale2_off < 0 || op.shift2_off < 0)) ||
            (((op.variant >> 16) & 1) && (op.out2 >= 0 || op.scale2_off < 0 || op.kh != 3 || op.kw != 3 || op.stride != 1 || op.pad != 1 ||
                                          op.pool || bad_w(op.scale2_off, (size_t)16 * op.coutp * 4)));
    } else if (op.type == TA_OP_RFSTEM) {
      bad = bad || op.w_off < 0 || (op.cout != 16 && op.cout != 32) || bad_w(op.w_off, (op.cout == 32 ? 448 + 704 : 448) * 4);
    } else if (op.type == TA_OP_DWPW) {
      bad = bad || op.w_off < 0 || op.bias_off < 0 || op.scale2_off < 0 || op.shift2_off < 0 || op.cin % 4 || op.cout % 4 ||
            (op.prec != 0 && op.prec != 3) || op.wus_off != op.bias_off + 4 * (int64_t)op.coutp || bad_w(op.wus_off, (size_t)op.coutp * 4) ||
            op.coutp % 32 || op.n_slabs <= 0 || op.n_slabs * 32 < op.cin || (op.stride != 1 && op.stride != 2) ||
            bad_w(op.w_off, (size_t)op.n_slabs * op.coutp * 128) || bad_w(op.bias_off, (size_t)op.coutp * 4) ||
            bad_w(op.scale2_off, (size_t)op.cin * 36) || bad_w(op.shift2_off, (size_t)op.cin * 4);
    } else if (op.type == TA_OP_DWCONV) {
      bad = bad || op.w_off < 0 || op.bias_off < 0 || op.cin % 4 || op.kh != 3 || op.kw != 3 ||
            bad_w(op.w_off, (size_t)op.cin * 36) || bad_w(op.bias_off, (size_t)op.cin * 4);
    }
    if (bad) {
      delete m;
      return ta_fail(ctx, TA_E_INVALID, "model blob: malformed op");
    }
  }
  {  // activation scales: per tensor an optional vector of per-channel powers of two in the weights region; the input is read by
     // the pre-processing kernels as it is
    bool bad = m->tdesc[h.input_tensor].unscale_off >= 0;
    m->unscale_host.resize(h.n_tensors);
    for (int t = 0; t < h.n_tensors && !bad; ++t) {
      const int64_t off = m->tdesc[t].unscale_off;
      if (off < 0) continue;
      const size_t need = (size_t)m->tdesc[t].channels * sizeof(float);
      if ((off & 3) || (size_t)off + need > (size_t)h.weights_bytes) {
        bad = true;
        break;
      }
      m->unscale_host[t].resize(m->tdesc[t].channels);
      memcpy(m->unscale_host[t].data(), (const char*)blob + h.weights_off + off, need);
    }
    for (int i = 0; i < h.n_outputs; ++i)
      if (h.outputs[i] < 0 || h.outputs[i] >= h.n_tensors) bad = true;
    m->tensor_read.assign(h.n_tensors, 0);
    for (auto& op : m->ops) {
      m->tensor_read[op.in] = 1;
      if (op.res >= 0) m->tensor_read[op.res] = 1;
    }
    for (int t = 0; t < h.n_tensors; ++t)            // a view is read when its source is, and the other way round
      if (m->tdesc[t].alias_of >= 0 && m->tdesc[t].alias_of < h.n_tensors && (m->tensor_read[t] || m->tensor_read[m->tdesc[t].alias_of]))
        m->tensor_read[t] = m->tensor_read[m->tdesc[t].alias_of] = 1;
    for (auto& op : m->ops) {
      if (op.type == TA_OP_DWCONV && (m->tdesc[op.in].unscale_off >= 0 || m->tdesc[op.out].unscale_off >= 0)) bad = true;
      if ((op.type == TA_OP_CONV || op.type == TA_OP_DWPW) && (op.prec == 3 || op.prec == 4 || op.prec == 5)) m->has_half_ops = true;
    }
    if (bad) {
      delete m;
      return ta_fail(ctx, TA_E_INVALID, "model blob: inconsistent activation scales");
    }
  }
  {  // lanes (variant bits 17..18): a branch may read what earlier main-stream ops wrote and its own tensors; nothing outside
     // the branch may touch what it writes, nothing later may write what it reads, and it takes plain convs only
    std::vector<int> writer_lane(h.n_tensors, 0), reader_lanes(h.n_tensors, 0);
    bool bad = false;
    for (auto& op : m->ops) {
      const int lane = (op.variant >> 17) & 3;
      if (lane == 3 || (lane && (op.type != TA_OP_CONV || ((op.variant >> 8) & 255) > 1 || op.pool))) bad = true;
      const int reads[3] = {op.in, op.res, -1};
      for (int t : reads) {
        if (t < 0) continue;
        if (writer_lane[t] && writer_lane[t] != lane) bad = true;            // a branch's result read outside the branch
        reader_lanes[t] |= 1 << lane;
      }
      const int writes[2] = {op.out, op.out2};
      for (int t : writes) {
        if (t < 0) continue;
        if (reader_lanes[t] & ~(1 << lane)) bad = true;                      // written while another lane may still read it
        if (writer_lane[t] && writer_lane[t] != lane) bad = true;
        if (lane) writer_lane[t] = lane;
        else if (reader_lanes[t] >> 1) bad = true;
      }
    }
    if (bad) {
      delete m;
      return ta_fail(ctx, TA_E_INVALID, "model blob: an op lane (side stream) shares tensors with ops outside it");
    }
  }
  hipError_t e = hipMalloc((void**)&m->weights_dev, h.weights_bytes ? h.weights_bytes : 16);
  if (e != hipSuccess) {
    delete m;
    return ta_fail(ctx, TA_E_DEVICE, "hipMalloc(weights %lld) failed: %s", (long long)h.weights_bytes, hipGetErrorString(e));
  }
  e = hipMemcpy(m->weights_dev, (const char*)blob + h.weights_off, h.weights_bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(m->weights_dev);
    delete m;
    return ta_fail(ctx, TA_E_DEVICE, "weights upload failed: %s", hipGetErrorString(e));
  }
  if (m->ops[0].type == TA_OP_RFSTEM)
    m->weights_host_small.assign((const char*)blob + h.weights_off + m->ops[0].w_off, (const char*)blob + h.weights_off + m->ops[0].w_off + 448 * 4);
  *out = m;
  return TA_OK;
}

// tools (tools/graph_probe.py): the op program of the LAST forward (same plan, same input) replayed `reps` times as plain stream
// launches and as a hipGraph captured from them (lanes = fork / join through events: part of the capture).
// out_ms[0] / [1]: GPU time per replay, streams / graph (HIP events on the main stream); out_ms[2] / [3]: host time the
// enqueue of one replay takes, streams / graph; out_ms[4]: capture + instantiate, once.
int ta_model_graph_probe(ta_model* m, int reps, double* out_ms) {
  ta_enter(m ? m->ctx : nullptr);
  if (!m || !out_ms || reps < 1) return TA_E_INVALID;
  ta_ctx* ctx = m->ctx;
  if (m->run_n <= 0) return ta_fail(ctx, TA_E_INVALID, "graph_probe: run a forward first (it replays that plan)");
  struct res_t {                                      // released on every return path
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    ~res_t() {
      if (exec) (void)hipGraphExecDestroy(exec);
      if (graph) (void)hipGraphDestroy(graph);
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } r;
  hipEvent_t& e0 = r.e0;
  hipEvent_t& e1 = r.e1;
  hipGraph_t& graph = r.graph;
  hipGraphExec_t& exec = r.exec;
  TA_HIP(ctx, hipEventCreate(&e0));
  TA_HIP(ctx, hipEventCreate(&e1));
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int i = 0; i < 2; ++i) TA_TRY(ta_model_run_ops(m));          // lazy state (function attributes, side streams) exists
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  float ms = 0.f;
  double t0 = now();
  TA_HIP(ctx, hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < reps; ++i) TA_TRY(ta_model_run_ops(m));
  TA_HIP(ctx, hipEventRecord(e1, ctx->stream));
  out_ms[2] = (now() - t0) / reps;
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  TA_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  out_ms[0] = ms / reps;
  t0 = now();
  TA_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
  const int rc = ta_model_run_ops(m);
  const hipError_t ce = hipStreamEndCapture(ctx->stream, &graph);
  if (rc != TA_OK) return rc;
  TA_HIP(ctx, ce);
  TA_HIP(ctx, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  out_ms[4] = now() - t0;
  for (int i = 0; i < 2; ++i) TA_HIP(ctx, hipGraphLaunch(exec, ctx->stream));
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  t0 = now();
  TA_HIP(ctx, hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < reps; ++i) TA_HIP(ctx, hipGraphLaunch(exec, ctx->stream));
  TA_HIP(ctx, hipEventRecord(e1, ctx->stream));
  out_ms[3] = (now() - t0) / reps;
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  TA_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  out_ms[1] = ms / reps;
  return TA_OK;
}

int ta_model_debug_amax(ta_model* m, int enable, float* out, int capacity) {
  ta_enter(m ? m->ctx : nullptr);
  if (!m) return TA_E_INVALID;
  ta_ctx* ctx = m->ctx;
  const size_t n = 2 * std::min(m->ops.size(), (size_t)TA_AMAX_OPS);
  unsigned* slots = (unsigned*)ctx->range_flag + TA_AMAX_SLOT0;      // one set per context: one model collects at a time
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (out) {
    if (!m->amax_on) return ta_fail(ctx, TA_E_INVALID, "debug_amax: not enabled");
    if ((size_t)capacity < 2 * m->ops.size()) return ta_fail(ctx, TA_E_CAPACITY, "debug_amax: %zu floats needed", 2 * m->ops.size());
    memset(out, 0, 2 * m->ops.size() * sizeof(float));
    TA_HIP(ctx, hipMemcpy(out, slots, n * sizeof(float), hipMemcpyDeviceToHost));   // bit patterns of |x| ARE the floats
  }
  if (enable == 2) return TA_OK;                 // read only: the collection goes on
  if (enable) {
    // the slots belong to the context, the switch to the model: a second model of the same context must not zero or overwrite
    // what the first one is collecting
    if (ctx->amax_owner && ctx->amax_owner != m)
      return ta_fail(ctx, TA_E_INVALID, "debug_amax: another model of this context is collecting (disable it first)");
    TA_HIP(ctx, hipMemset(slots, 0, 2 * TA_AMAX_OPS * sizeof(unsigned)));
    ctx->amax_owner = m;
  } else if (ctx->amax_owner == m) {
    ctx->amax_owner = nullptr;
  }
  m->amax_on = enable != 0;
  return TA_OK;
}

int ta_model_tensor_unscale(const ta_model* m, int tensor, float* out, int capacity) {
  if (!m || !out || tensor < 0 || tensor >= (int)m->tdesc.size()) return TA_E_INVALID;
  const int c = m->tdesc[tensor].channels;
  if (capacity < c) return TA_E_CAPACITY;
  for (int i = 0; i < c; ++i) out[i] = m->unscale_host[tensor].empty() ? 1.0f : m->unscale_host[tensor][i];
  return TA_OK;
}

void ta_model_free(ta_model* m) {
  ta_enter(m ? m->ctx : nullptr);
  if (!m) return;
  (void)hipStreamSynchronize(m->ctx->stream);
  if (m->ctx->amax_owner == m) m->ctx->amax_owner = nullptr;
  free_plans(m);
  if (m->weights_dev) (void)hipFree(m->weights_dev);
  delete m;
}

int ta_model_kind(const ta_model* m) { return m ? m->kind : TA_E_INVALID; }

int ta_model_forward_frames(ta_model* m, const ta_frames* f) {
  ta_enter(m ? m->ctx : nullptr);
  if (!m || !f) return TA_E_INVALID;
  ta_ctx* ctx = m->ctx;
  if (m->kind != TA_MODEL_RETINAFACE && m->kind != TA_MODEL_OPENPOSE)
    return ta_fail(ctx, TA_E_INVALID, "forward_frames: model kind %d takes crops", m->kind);
  if (f->n == 0) return TA_OK;
  TA_TRY(ta_model_plan(m, f->n, f->h, f->w));
  m->input_u8 = f->dev;
  if (m->ops[0].type != TA_OP_RFSTEM)              // that op reads the frames itself
    TA_TRY(ta_launch_preprocess(ctx, m->kind == TA_MODEL_RETINAFACE ? TA_PRE_RETINAFACE : TA_PRE_OPENPOSE, f->dev, f->n,
                                f->h, f->w, m->tensors[m->hdr.input_tensor]));
  return ta_model_run_ops(m);
}

int ta_model_forward_crops(ta_model* m, const uint8_t* crops, int n) {
  ta_enter(m ? m->ctx : nullptr);
  if (!m || (!crops && n > 0)) return TA_E_INVALID;
  ta_ctx* ctx = m->ctx;
  if (m->kind != TA_MODEL_ARCFACE) return ta_fail(ctx, TA_E_INVALID, "forward_crops: not an ArcFace model");
  if (n == 0) return TA_OK;
  TA_TRY(ta_model_plan(m, n, 112, 112));
  void* scr = nullptr;
  const size_t bytes = (size_t)n * 3 * 112 * 112;
  TA_TRY(ta_scratch(ctx, bytes, &scr));
  TA_HIP(ctx, hipMemcpyAsync(scr, crops, bytes, hipMemcpyHostToDevice, ctx->stream));
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  TA_TRY(ta_launch_preprocess(ctx, TA_PRE_ARCFACE_CROPS, (const uint8_t*)scr, n, 112, 112,
                              m->tensors[m->hdr.input_tensor]));
  return ta_model_run_ops(m);
}

int ta_model_tensor_shape(ta_model* m, int tensor, int* n, int* c, int* h, int* w) {
  if (!m || tensor < 0 || tensor >= (int)m->tensors.size()) return TA_E_INVALID;
  const ta_tensor& t = m->tensors[tensor];
  if (n) *n = m->run_n;
  if (c) *c = t.c;
  if (h) *h = t.h;
  if (w) *w = t.w;
  return TA_OK;
}

int ta_model_read_tensor(ta_model* m, int tensor, int ch_off, int ch, float* dst) {
  ta_enter(m ? m->ctx : nullptr);
  if (!m || !dst || tensor < 0 || tensor >= (int)m->tensors.size()) return TA_E_INVALID;
  ta_ctx* ctx = m->ctx;
  const ta_tensor& t = m->tensors[tensor];
  if (!t.dev || ch_off < 0 || ch <= 0 || ch_off + ch > t.c) return ta_fail(ctx, TA_E_INVALID, "read_tensor: bad slice");
  std::vector<float> host((size_t)m->run_n * t.hp() * t.wp() * t.c);
  TA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  TA_HIP(ctx, hipMemcpy(host.data(), t.dev, host.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (int i = 0; i < m->run_n; ++i)
    for (int c = 0; c < ch; ++c)
      for (int y = 0; y < t.h; ++y)
        for (int x = 0; x < t.w; ++x)
        {
          const int cc = ch_off + c;
          float v;
          if (t.fmt == TA_FMT_F16) {
            _Float16 hh;
            memcpy(&hh, (const char*)&host[t.off(i, y, x)] + 2 * cc, 2);
            v = (float)hh;
          } else if (t.fmt != TA_FMT_F32) {
            const char* b = (const char*)&host[t.off(i, y, x)] + ((cc >> 5) << 7) + ((cc & 31) << 1);
            uint16_t h16, l16;
            memcpy(&h16, b, 2);
            memcpy(&l16, b + 64, 2);
            if (t.fmt == TA_FMT_SPLIT16) {
              _Float16 hh, ll;
              memcpy(&hh, &h16, 2);
              memcpy(&ll, &l16, 2);
              v = (float)hh + (float)ll;
            } else {
              const uint32_t hb = (uint32_t)h16 << 16, lb = (uint32_t)l16 << 16;
              float hf, lf;
              memcpy(&hf, &hb, 4);
              memcpy(&lf, &lb, 4);
              v = hf + lf;
            }
          } else {
            v = host[t.off(i, y, x) + cc];
          }
          dst[(((size_t)i * ch + c) * t.h + y) * t.w + x] = t.unscale_host ? v * t.unscale_host[cc] : v;   // channel cc is stored times 2^a[cc]
        }
  return TA_OK;
}

}  // extern "C"

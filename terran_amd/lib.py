"""ctypes binding of `libterran_amd.so` (the C ABI in include/terran_amd.h).

There is no fallback: if the library is missing or no gfx950 device is usable, the
product path raises `TerranAmdError`.  Nothing here imports `oracle/`.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libterran_amd.so')

OK, E_INVALID, E_DEVICE, E_CAPACITY, E_OVERFLOW, E_RANGE = 0, -1, -2, -3, -4, -5
_CODES = {E_INVALID: 'TA_E_INVALID', E_DEVICE: 'TA_E_DEVICE', E_CAPACITY: 'TA_E_CAPACITY', E_OVERFLOW: 'TA_E_OVERFLOW',
          E_RANGE: 'TA_E_RANGE'}


class TerranAmdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('%s: %s' % (_CODES.get(code, code), msg))
        self.code = code


c_int, c_float, c_double, c_void_p, c_size_t = C.c_int, C.c_float, C.c_double, C.c_void_p, C.c_size_t
P = C.POINTER
u8p, i32p, f32p, f64p = P(C.c_uint8), P(C.c_int32), P(C.c_float), P(C.c_double)

# name -> (restype, argtypes); mirrors include/terran_amd.h one to one
SIGNATURES = {
    'ta_version': (C.c_char_p, []),
    'ta_device_count': (c_int, []),
    'ta_device_pci_bus_id': (c_int, [c_int, C.c_char_p, c_int]),
    'ta_ctx_create': (c_int, [c_int, P(c_void_p)]),
    'ta_ctx_destroy': (None, [c_void_p]),
    'ta_last_error': (C.c_char_p, [c_void_p]),
    'ta_ctx_sync': (c_int, [c_void_p]),
    'ta_profile_enable': (c_int, [c_void_p, c_int]),
    'ta_profile_reset': (c_int, [c_void_p]),
    'ta_profile_read': (c_int, [c_void_p, c_int, f64p, P(C.c_int64), f64p]),
    'ta_timer_start': (c_int, [c_void_p]),
    'ta_timer_stop': (c_int, [c_void_p, f64p]),
    'ta_host_alloc': (c_int, [c_void_p, c_size_t, P(c_void_p)]),
    'ta_host_free': (None, [c_void_p, c_void_p]),
    'ta_frames_upload': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, P(c_void_p)]),
    'ta_frames_alloc': (c_int, [c_void_p, c_int, c_int, c_int, P(c_void_p)]),
    'ta_frames_shape': (c_int, [c_void_p, P(c_int), P(c_int), P(c_int)]),
    'ta_frames_download': (c_int, [c_void_p, c_void_p]),
    'ta_frames_free': (None, [c_void_p]),
    'ta_frames_resize': (c_int, [c_void_p, c_void_p, c_int, c_int, P(c_void_p)]),
    'ta_frames_resize_bicubic': (c_int, [c_void_p, c_void_p, c_int, c_int, P(c_void_p)]),
    'ta_frames_paste': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int]),
    'ta_model_load': (c_int, [c_void_p, c_int, c_void_p, c_size_t, P(c_void_p)]),
    'ta_model_free': (None, [c_void_p]),
    'ta_model_kind': (c_int, [c_void_p]),
    'ta_model_forward_frames': (c_int, [c_void_p, c_void_p]),
    'ta_model_forward_crops': (c_int, [c_void_p, c_void_p, c_int]),
    'ta_model_tensor_shape': (c_int, [c_void_p, c_int, P(c_int), P(c_int), P(c_int), P(c_int)]),
    'ta_model_tensor_unscale': (c_int, [c_void_p, c_int, c_void_p, c_int]),
    'ta_model_debug_amax': (c_int, [c_void_p, c_int, c_void_p, c_int]),
    'ta_model_graph_probe': (c_int, [c_void_p, c_int, c_void_p]),
    'ta_model_read_tensor': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    'ta_retinaface_run': (c_int, [c_void_p, c_void_p, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p, P(C.c_int32)]),
    'ta_retinaface_postprocess': (c_int, [c_void_p, P(c_void_p), c_int, c_int, c_int, c_float, c_float, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p, P(C.c_int32)]),
    'ta_arcface_embed_crops': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'ta_arcface_embed_faces': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'ta_arcface_embed_faces_multi': (c_int, [c_void_p, P(c_void_p), c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                             c_void_p]),
    'ta_cosine_distance': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'ta_openpose_run': (c_int, [c_void_p, c_void_p, c_double, c_int, c_void_p, c_void_p, c_void_p, P(C.c_int32)]),
    'ta_openpose_group': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_int, c_void_p,
                                  c_void_p, c_void_p, P(C.c_int32)]),
    'ta_bicubic_x8': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ta_openpose_last_stats': (c_int, [c_void_p, c_void_p, c_void_p]),
    'ta_openpose_debug_read': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                       c_void_p]),
    'ta_debug_conv_variant': (c_int, [c_void_p, c_int]),
    'ta_debug_range_check': (c_int, [c_void_p]),
    'ta_debug_conv_counts': (c_int, [c_void_p, c_void_p, c_int]),
    'ta_debug_kernel_work': (c_int, [c_void_p, C.c_char_p, c_size_t, c_int]),
}

# conv kernel variants (include/terran_amd.h TA_CONV_*)
CONV_VARIANTS = {'auto': 0, 'generic': 1, 'pipe64': 2, 'pipe128': 3, 'split_2x2': 4, 'split_2x2_p8': 5, 'split_2x4': 6,
                 'split_1x4': 7, 'win_2x2': 8, 'win_2x4': 9, 'win_1x4': 10, 'split_1x4_w2': 11, 'split_2x2_w2': 12}

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TerranAmdError(E_DEVICE, 'libterran_amd.so not built (run `python -m terran_amd.build`); '
                                           'there is no CPU fallback')
        # the .so travels prebuilt to the GPU box: it must be the build of the sources that travel beside it
        from . import build as _build
        try:
            with open(_build.STAMP) as fh:
                stamp = fh.read().strip()
        except OSError:
            stamp = None
        try:
            current = _build.source_hash()
        except OSError:                             # relocated package without its sources: nothing to compare with
            current = stamp
        if (stamp is None or stamp != current) and not os.environ.get('TA_ALLOW_STALE_LIB'):
            raise TerranAmdError(E_DEVICE, 'libterran_amd.so was not built from the sources in terran_amd/csrc '
                                           '(run `python -m terran_amd.build`)')
        # A lane of the StreamPipeline is four HIP streams and a GPU runs several lanes; ROCm maps all streams of a process
        # onto GPU_MAX_HW_QUEUES hardware queues (default 4), and kernels of different streams that share a queue wait for
        # each other.  12 queues measured +3 % on the 1080p pipeline (16 streams).  Only a default: the user's setting wins,
        # and it has no effect when the HIP runtime was initialised before this library was loaded.
        os.environ.setdefault('GPU_MAX_HW_QUEUES', '12')
        # kernel-argument blocks in device memory instead of host-coherent memory: a conv workgroup's first instruction is the
        # fetch of its 400-byte launch record (~2 000 cycles from host memory; ArcFace at 64 crops 4.84 -> 4.67 ms)
        os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
        lib = C.CDLL(LIB_PATH)
        missing = []
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        if missing and not os.environ.get('TA_BRINGUP_ALLOW_MISSING'):
            raise TerranAmdError(E_INVALID, 'libterran_amd.so lacks ABI symbols: %s' % ', '.join(missing))
        _lib = lib
    return _lib


def ptr(a):
    return a.ctypes.data_as(c_void_p) if a is not None else None


class Context:
    """One per GPU (and per host thread)."""

    def __init__(self, device_id=0):
        self.lib = load()
        h = c_void_p()
        rc = self.lib.ta_ctx_create(int(device_id), C.byref(h))
        if rc != OK:
            raise TerranAmdError(rc, 'ta_ctx_create(%d) failed: no usable gfx950 device '
                                     '(ta_device_count=%d)' % (device_id, self.lib.ta_device_count()))
        self.h = h
        self.device_id = device_id

    def check(self, rc):
        if rc != OK:
            raise TerranAmdError(rc, self.lib.ta_last_error(self.h).decode(errors='replace'))

    def last_error(self):
        return self.lib.ta_last_error(self.h).decode(errors='replace')

    def sync(self):
        self.check(self.lib.ta_ctx_sync(self.h))

    def close(self):
        if getattr(self, 'h', None):
            self.lib.ta_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # profiling / timing
    def profile(self, on):
        self.check(self.lib.ta_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self.check(self.lib.ta_profile_reset(self.h))

    def profile_read(self, klass):
        ms, n, w = c_double(), C.c_int64(), c_double()
        self.check(self.lib.ta_profile_read(self.h, klass, C.byref(ms), C.byref(n), C.byref(w)))
        return ms.value, n.value, w.value

    def timer_start(self):
        self.check(self.lib.ta_timer_start(self.h))

    def timer_stop(self):
        ms = c_double()
        self.check(self.lib.ta_timer_stop(self.h, C.byref(ms)))
        return ms.value

    # frames
    def upload(self, images):
        images = np.ascontiguousarray(images, dtype=np.uint8)
        assert images.ndim == 4 and images.shape[3] == 3
        return Frames(self, images)

    def pinned_array(self, shape):
        """uint8 numpy array backed by page-locked host memory (freed with `free_pinned`)."""
        n = int(np.prod(shape))
        p = c_void_p()
        self.check(self.lib.ta_host_alloc(self.h, n, C.byref(p)))
        arr = np.ctypeslib.as_array((C.c_uint8 * n).from_address(p.value)).reshape(shape)
        arr_ptr = p.value
        return arr, arr_ptr

    def free_pinned(self, ptr):
        if getattr(self, 'h', None):
            self.lib.ta_host_free(self.h, c_void_p(ptr))

    def cosine_distance(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        out = np.empty((a.shape[0], b.shape[0]), np.float32)
        self.check(self.lib.ta_cosine_distance(self.h, ptr(a), a.shape[0], ptr(b), b.shape[0], a.shape[1], ptr(out)))
        return out

    def conv_variant(self, name):
        """Debug: force every following conv on this context onto one kernel variant ('auto' to release)."""
        self.check(self.lib.ta_debug_conv_variant(self.h, CONV_VARIANTS[name]))

    def conv_counts(self, reset=False):
        """Debug: {variant name: conv launches since the last reset}."""
        c = np.zeros(16, np.int64)
        self.check(self.lib.ta_debug_conv_counts(self.h, ptr(c), int(reset)))
        out = {k: int(c[v]) for k, v in CONV_VARIANTS.items() if v and c[v]}
        if c[15]:
            out['lean_epilogue'] = int(c[15])          # split-role launches that ran the specialised drain
        return out

    def kernel_work(self, reset=False):
        """{kernel instance name: (launches, algorithmic FLOPs, HIP-event ms)} of the dense-conv kernels since the last reset; the
        time covers the launches made while `profile(True)` was on (0.0 otherwise)."""
        buf = C.create_string_buffer(1 << 16)
        self.check(self.lib.ta_debug_kernel_work(self.h, buf, len(buf), int(reset)))
        out = {}
        for line in buf.value.decode().splitlines():
            f = line.split(';')
            out[f[0]] = (int(f[1]), float(f[2]), float(f[3]) if len(f) > 3 else 0.0)
        return out

    def pose_debug(self, n, cap_peaks=1024, cap_conn=1024):
        """Debug taps of the last OpenPose run / grouping on this context -> (peaks, connections):
        peaks[i][part] = (yx (k,2) int32, scores (k,) f32); connections[i][limb] = None (limb skipped) or
        (ij (k,2) int32, scores (k,) f32)."""
        pc = np.zeros((n, 18), np.int32)
        pyx = np.zeros((n, 18, cap_peaks, 2), np.int32)
        psc = np.zeros((n, 18, cap_peaks), np.float32)
        cc = np.zeros((n, 19), np.int32)
        cij = np.zeros((n, 19, cap_conn, 2), np.int32)
        csc = np.zeros((n, 19, cap_conn), np.float32)
        self.check(self.lib.ta_openpose_debug_read(self.h, n, cap_peaks, ptr(pc), ptr(pyx), ptr(psc), cap_conn,
                                                   ptr(cc), ptr(cij), ptr(csc)))
        assert pc.max(initial=0) <= cap_peaks and cc.max(initial=0) <= cap_conn
        peaks = [[(pyx[i, p, :pc[i, p]].copy(), psc[i, p, :pc[i, p]].copy()) for p in range(18)] for i in range(n)]
        conns = [[None if cc[i, l] < 0 else (cij[i, l, :cc[i, l]].copy(), csc[i, l, :cc[i, l]].copy())
                  for l in range(19)] for i in range(n)]
        return peaks, conns

    def pose_stats(self):
        """(peaks, limb connections) of the last OpenPose grouping run on this context."""
        a, b = C.c_int64(), C.c_int64()
        self.check(self.lib.ta_openpose_last_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def bicubic_x8(self, maps):
        maps = np.ascontiguousarray(maps, dtype=np.float32)
        n, c, h, w = maps.shape
        out = np.empty((n, c, 8 * h, 8 * w), np.float32)
        self.check(self.lib.ta_bicubic_x8(self.h, ptr(maps), n, c, h, w, ptr(out)))
        return out


class Frames:
    """uint8 RGB (N,H,W,3) batch resident in HBM."""

    def __init__(self, ctx, images=None, handle=None):
        self.ctx = ctx
        if handle is not None:
            self.h = handle
        else:
            h = c_void_p()
            n, hh, ww = images.shape[:3]
            ctx.check(ctx.lib.ta_frames_upload(ctx.h, ptr(images), n, hh, ww, C.byref(h)))
            self.h = h
        n, hh, ww = c_int(), c_int(), c_int()
        ctx.lib.ta_frames_shape(self.h, C.byref(n), C.byref(hh), C.byref(ww))
        self.shape = (n.value, hh.value, ww.value, 3)

    def __len__(self):
        return self.shape[0]

    @classmethod
    def zeros(cls, ctx, n, h, w):
        hd = c_void_p()
        ctx.check(ctx.lib.ta_frames_alloc(ctx.h, n, h, w, C.byref(hd)))
        return cls(ctx, handle=hd)

    def resize(self, h, w, ctx=None):
        """`ctx`: the context (stream, scratch) the resize runs on and the result belongs to -- the CALLER's, so that a
        batch handed over by another thread's context (video.RawVideoReader) is only ever read here, never driven."""
        ctx = ctx or self.ctx
        hd = c_void_p()
        ctx.check(ctx.lib.ta_frames_resize(ctx.h, self.h, int(h), int(w), C.byref(hd)))
        return Frames(ctx, handle=hd)

    def resize_bicubic(self, h, w, ctx=None):
        ctx = ctx or self.ctx
        hd = c_void_p()
        ctx.check(ctx.lib.ta_frames_resize_bicubic(ctx.h, self.h, int(h), int(w), C.byref(hd)))
        return Frames(ctx, handle=hd)

    def paste(self, src, src_index, dst_index, top, left):
        self.ctx.check(self.ctx.lib.ta_frames_paste(self.ctx.h, src.h, src_index, self.h, dst_index, top, left))

    def download(self):
        out = np.empty(self.shape, np.uint8)
        self.ctx.check(self.ctx.lib.ta_frames_download(self.h, ptr(out)))
        return out

    def free(self):
        if getattr(self, 'h', None) and getattr(self.ctx, 'h', None):
            self.ctx.lib.ta_frames_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Model:
    """A packed model on one context."""

    def __init__(self, ctx, program):
        self.ctx = ctx
        self.kind = program.kind
        self.names = dict(program.names)
        blob = program.blob()
        buf = np.frombuffer(blob, dtype=np.uint8)
        h = c_void_p()
        ctx.check(ctx.lib.ta_model_load(ctx.h, program.kind, ptr(buf), len(blob), C.byref(h)))
        self.h = h

    def forward_frames(self, frames):
        self.ctx.check(self.ctx.lib.ta_model_forward_frames(self.h, frames.h))

    def forward_crops(self, crops):
        crops = np.ascontiguousarray(crops, dtype=np.uint8)
        self.ctx.check(self.ctx.lib.ta_model_forward_crops(self.h, ptr(crops), crops.shape[0]))

    def read(self, name):
        """Debug tap: named tensor slice as float32 NCHW."""
        tid, off, ch = self.names[name]
        n, c, h, w = c_int(), c_int(), c_int(), c_int()
        self.ctx.check(self.ctx.lib.ta_model_tensor_shape(self.h, tid, C.byref(n), C.byref(c), C.byref(h), C.byref(w)))
        out = np.empty((n.value, ch, h.value, w.value), np.float32)
        self.ctx.check(self.ctx.lib.ta_model_read_tensor(self.h, tid, off, ch, ptr(out)))
        return out

    def read_tensor(self, tid, ch_off=0, ch=None):
        """Debug tap by tensor id (true values: the activation scale is divided out)."""
        n, c, h, w = c_int(), c_int(), c_int(), c_int()
        self.ctx.check(self.ctx.lib.ta_model_tensor_shape(self.h, tid, C.byref(n), C.byref(c), C.byref(h), C.byref(w)))
        ch = c.value - ch_off if ch is None else ch
        out = np.empty((n.value, ch, h.value, w.value), np.float32)
        self.ctx.check(self.ctx.lib.ta_model_read_tensor(self.h, tid, ch_off, ch, ptr(out)))
        return out

    def tensor_unscale(self, tid, channels):
        """Per-channel factors 2^-a[c] the stored values of tensor `tid` are multiplied with to get the true ones."""
        out = np.empty(channels, np.float32)
        self.ctx.check(self.ctx.lib.ta_model_tensor_unscale(self.h, tid, ptr(out), channels))
        return out

    def amax_collect(self, on=True):
        """Start (zeroed) / stop collecting the largest |x| every conv / dw+pw op stores (ta_model_debug_amax)."""
        self.ctx.check(self.ctx.lib.ta_model_debug_amax(self.h, int(bool(on)), None, 0))

    def graph_probe(self, reps=20):
        """Tools: (gpu_ms_streams, gpu_ms_graph, host_ms_streams, host_ms_graph, capture_ms) of the last forward's op program."""
        out = np.zeros(5, np.float64)
        self.ctx.check(self.ctx.lib.ta_model_graph_probe(self.h, int(reps), ptr(out)))
        return tuple(float(x) for x in out)

    def amax_read(self, n_ops):
        """-> (n_ops, 2) float32 in STORED units: [:, 0] op outputs, [:, 1] depthwise intermediates of dw+pw ops; the
        collection goes on (not zeroed)."""
        out = np.zeros(2 * n_ops, np.float32)
        n = c_int()
        # read without restarting: enable = 1 would zero the slots, so read first through a second call
        self.ctx.check(self.ctx.lib.ta_model_debug_amax(self.h, 2, ptr(out), out.size))
        return out.reshape(n_ops, 2)

    def free(self):
        if getattr(self, 'h', None) and getattr(self.ctx, 'h', None):
            self.ctx.lib.ta_model_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

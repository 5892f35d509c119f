"""`OpenPose` -- drop-in for terran/pose/openpose/wrapper.py:166-485 on MI355X."""
import ctypes as C

import numpy as np

from . import lib, pack, runtime


class PoseOverflow(RuntimeError):
    """Kept for callers that catch it: the library no longer caps the grouping lists (an image that outgrows the LDS
    fast path is recomputed with lists in global memory, include/terran_amd.h: ta_openpose_run), so a per-image count
    of -1 -- what this exception reported, `results` holding `None` at the positions in `images` -- cannot come back."""

    def __init__(self, message, results, images):
        super().__init__(message)
        self.results = results
        self.images = images


def _unpack(counts, kp, sc):
    out, o = [], 0
    for c in counts:
        if c < 0:                                   # over a cap: no rows for this image
            out.append(None)
            continue
        out.append([{'keypoints': kp[i].copy(), 'score': np.float64(sc[i])} for i in range(o, o + int(c))])
        o += int(c)
    return out


def _run(ctx, fn, n):
    counts = np.zeros(max(n, 1), np.int32)
    cap = max(64, 16 * n)
    while True:
        kp = np.empty((cap, 18, 3), np.int32)
        sc = np.empty(cap, np.float64)
        req = C.c_int32(0)
        rc = fn(cap, lib.ptr(counts), lib.ptr(kp), lib.ptr(sc), C.byref(req))
        if rc == lib.E_CAPACITY:
            cap = int(req.value)
            continue
        ctx.check(rc)
        out = _unpack(counts[:n], kp, sc)
        over = [i for i, r in enumerate(out) if r is None]
        if over:
            raise PoseOverflow(ctx.last_error() + '; images %s' % over, out, over)
        return out


class OpenPose(runtime.RangeFallback):

    def __init__(self, device=None, short_side=184, state=None, ctx=None, precision=None):
        self.device = device
        self.precision = runtime.resolve_precision(precision)
        self.short_side = short_side
        self.ctx = ctx if ctx is not None else runtime.get_context(device)     # ctx: an extra stream on the same GPU
        self.model = lib.Model(self.ctx, runtime.packed_program('openpose', state, self.precision))
        self._init_fallback('openpose', state)

    def call_frames(self, frames):
        """frames: lib.Frames at ORIGINAL resolution; resized on the device to `short_side`."""
        n, H, W = frames.shape[:3]
        if n == 0:
            return []
        scale = self.short_side / min(H, W)
        nw, nh = int(W * scale), int(H * scale)                  # openpose/wrapper.py:95-99
        resized = frames if (nh, nw) == (H, W) else frames.resize(nh, nw, ctx=self.ctx)
        try:
            return self._with_fallback(lambda model: _run(self.ctx, lambda cap, *a: self.ctx.lib.ta_openpose_run(
                model.h, resized.h, float(scale), cap, *a), n))
        finally:
            if resized is not frames:
                resized.free()

    def call(self, images):
        """images: (N,H,W,3) uint8 RGB -> list[N] of list[{'keypoints': int32 (18,3), 'score': float64}],
        keypoints (x, y, present) in the coordinates of `images`."""
        frames = self.ctx.upload(np.asarray(images))
        try:
            return self.call_frames(frames)
        finally:
            frames.free()


def group(ctx, pafs, heatmaps, scale=1.0):
    """Debug/parity entry: grouping only, on network-resolution maps (N,38,h,w), (N,19,h,w)."""
    pafs = np.ascontiguousarray(pafs, dtype=np.float32)
    heatmaps = np.ascontiguousarray(heatmaps, dtype=np.float32)
    n, _, h, w = pafs.shape
    return _run(ctx, lambda cap, *a: ctx.lib.ta_openpose_group(ctx.h, lib.ptr(pafs), lib.ptr(heatmaps), n, h, w,
                                                               float(scale), cap, *a), n)

"""`RetinaFace` -- drop-in for terran/face/detection/retinaface/wrapper.py:92-238 on MI355X."""
import ctypes as C

import numpy as np

from . import lib, pack, results, runtime


class RetinaFace(runtime.RangeFallback):

    def __init__(self, device=None, nms_threshold=0.4, state=None, ctx=None, precision=None, lazy_results=False):
        self.device = device
        self.lazy_results = lazy_results               # opt-in: per-image results.LazyFaces instead of plain lists (results.py)
        self.precision = runtime.resolve_precision(precision)
        self.nms_threshold = nms_threshold
        self.ctx = ctx if ctx is not None else runtime.get_context(device)     # ctx: an extra stream on the same GPU
        self.model = lib.Model(self.ctx, runtime.packed_program('retinaface', state, self.precision))
        self._init_fallback('retinaface', state)       # f16x3: the refiner's half-float range (TA_E_RANGE -> exact-f32 twin)

    def detect_arrays(self, frames, threshold=0.5):
        """-> (counts (N,) int32, boxes (T,4), landmarks (T,5,2), scores (T,)) float32, images concatenated."""
        ctx = self.ctx
        n = frames.shape[0]
        counts = np.zeros(n, np.int32)
        if n == 0:
            return counts, np.empty((0, 4), np.float32), np.empty((0, 5, 2), np.float32), np.empty(0, np.float32)
        return self._with_fallback(lambda model: self._detect_arrays(model, frames, threshold, counts, n))

    def _detect_arrays(self, model, frames, threshold, counts, n):
        ctx = self.ctx
        cap = getattr(self, '_cap', max(256, 64 * n))
        while True:
            boxes = np.empty((cap, 4), np.float32)
            lmks = np.empty((cap, 5, 2), np.float32)
            scores = np.empty(cap, np.float32)
            req = C.c_int32(0)
            rc = ctx.lib.ta_retinaface_run(model.h, frames.h, float(threshold), float(self.nms_threshold), cap,
                                           lib.ptr(counts), lib.ptr(boxes), lib.ptr(lmks), lib.ptr(scores),
                                           C.byref(req))
            if rc == lib.E_CAPACITY:
                cap = self._cap = int(req.value) * 5 // 4 + 64
                continue
            ctx.check(rc)
            break
        total = int(counts.sum())
        return counts, boxes[:total], lmks[:total], scores[:total]

    def call_frames(self, frames, threshold=0.5):
        """frames: lib.Frames (N,H,W,3) at network resolution, resident in HBM."""
        return results.detections(*self.detect_arrays(frames, threshold), lazy=self.lazy_results or None)

    def call(self, images, threshold=0.5):
        """images: (N,H,W,3) uint8 RGB ndarray -> list[N] of list[{'bbox','landmarks','score'}] in
        descending score order, network-input pixel coordinates."""
        frames = self.ctx.upload(np.asarray(images))
        try:
            return self.call_frames(frames, threshold)
        finally:
            frames.free()


def postprocess(ctx, outputs, H, W, threshold=0.5, nms_threshold=0.4):
    """Debug/parity entry: the wrapper's decode + NMS tail on nine reference-layout head arrays."""
    heads = [np.ascontiguousarray(o, dtype=np.float32) for o in outputs]
    n = heads[0].shape[0]
    arr = (C.c_void_p * 9)(*[h.ctypes.data for h in heads])
    counts = np.zeros(max(n, 1), np.int32)
    cap = 1024
    while True:
        boxes = np.empty((cap, 4), np.float32)
        lmks = np.empty((cap, 5, 2), np.float32)
        scores = np.empty(cap, np.float32)
        req = C.c_int32(0)
        rc = ctx.lib.ta_retinaface_postprocess(ctx.h, arr, n, int(H), int(W), float(threshold), float(nms_threshold),
                                               cap, lib.ptr(counts), lib.ptr(boxes), lib.ptr(lmks), lib.ptr(scores),
                                               C.byref(req))
        if rc == lib.E_CAPACITY:
            cap = int(req.value)
            continue
        ctx.check(rc)
        break
    out, o = [], 0
    for c in counts[:n]:
        out.append([{'bbox': boxes[i].copy(), 'landmarks': lmks[i].copy(), 'score': scores[i]}
                    for i in range(o, o + int(c))])
        o += int(c)
    return out

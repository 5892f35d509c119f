"""Task facades with the reference's call signatures and return values:

  Detection   : terran/face/detection/__init__.py:185-287  (resize -> pad-merge -> model -> un-pad -> un-scale)
  Recognition : terran/face/recognition/__init__.py:7-90
  Estimation  : terran/pose/__init__.py:131-223

Generic pre-processing runs on the device (`ta_frames_resize` = cv2 INTER_LINEAR semantics,
`ta_frames_paste` = zero pad-merge), so a batch of frames is uploaded once.
"""
import math
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import lib, results, runtime
from .checkpoint import get_class_for_checkpoint
from .shard import shard_bounds


def _is_single(images):
    return not isinstance(images, (list, tuple)) and len(images.shape) == 3


def _pads(shapes, canvas=None):
    """Per image ((top,bottom),(left,right)) to the max size; the odd pixel goes top/left.
    `canvas` = (mh, mw) of the WHOLE list when this call only sees a shard of it (`_Fanout`)."""
    mh = max(s[0] for s in shapes) if canvas is None else canvas[0]
    mw = max(s[1] for s in shapes) if canvas is None else canvas[1]
    pads = []
    for h, w in shapes:
        dh, dw = max(0, (mh - h) / 2), max(0, (mw - w) / 2)
        pads.append(((int(math.ceil(dh)), int(math.floor(dh))), (int(math.ceil(dw)), int(math.floor(dw)))))
    return mh, mw, pads


def _merge(ctx, frame_list, canvas=None):
    """List of single-image lib.Frames -> one padded lib.Frames + pads."""
    mh, mw, pads = _pads([f.shape[1:3] for f in frame_list], canvas)
    canvas = lib.Frames.zeros(ctx, len(frame_list), mh, mw)
    for i, (f, p) in enumerate(zip(frame_list, pads)):
        canvas.paste(f, 0, i, p[0][0], p[1][0])
    return canvas, pads


class ShardedFrames:
    """A host frame batch uploaded ONCE across the devices of a fan-out: one resident `lib.Frames` per replica (its
    contiguous sub-batch).  Returned by `Detection / Recognition / Estimation(device=[...]).upload(images)` and accepted
    by all three facades built over the same device list, so detect, embed and pose share one PCIe upload per device."""

    def __init__(self, parts, bounds, n):
        self.parts, self.bounds, self.n = parts, bounds, n
        first = next((p for p in parts if p is not None), None)        # replicas beyond the batch size hold no part
        self.shape = (n,) + tuple(first.shape[1:]) if first is not None else (n, 0, 0, 3)

    def __len__(self):
        return self.n

    def free(self):
        for p in self.parts:
            if p is not None:
                p.free()
        self.parts = []


class _Fanout:
    """SURVEY.md 8(e) inside ONE process: `device=[0, 1, ..., 7]` makes a facade a fan-out over one replica of itself
    per listed device -- its own context (HIP stream, scratch, pinned staging) and weights, driven by its own host
    thread.  A call cuts the frame batch (or list) into contiguous sub-batches, one per device (`shard.shard_bounds`:
    video order is kept, sizes differ by at most one), every thread uploads and processes its sub-batch, and the
    variable-length per-frame results are concatenated in device order on the calling thread.  Frames are independent
    through the whole path: there is no collective, and the result equals the one-device result exactly.  The same
    device may be listed more than once (several streams on one GPU)."""

    def __init__(self, devices, make_replica):
        self.devices = list(devices)
        if not self.devices:
            raise ValueError('`device` list is empty')
        # every replica gets a context of its own (stream, scratch, staging): replicas of different facades run at the
        # same time on their own host threads, and a context belongs to one thread at a time
        self.replicas = [make_replica(d, runtime.new_context(runtime.device_index(d))) for d in self.devices]
        self.pool = ThreadPoolExecutor(max_workers=len(self.devices), thread_name_prefix='terran_amd-device')

    def upload(self, images):
        """Scatter: every replica's thread uploads its contiguous sub-batch to its device -> ShardedFrames."""
        images = np.asarray(images)
        n, k = len(images), len(self.replicas)
        bounds = [shard_bounds(n, k, r) for r in range(k)]
        futs = [self.pool.submit(rep._ctx().upload, images[lo:hi]) if hi > lo else None
                for rep, (lo, hi) in zip(self.replicas, bounds)]
        return ShardedFrames([f.result() if f is not None else None for f in futs], bounds, n)

    def __call__(self, items, *per_item, **kw):
        """items: ndarray batch, list or ShardedFrames; per_item: sequences aligned with it (sharded the same way)."""
        if isinstance(items, ShardedFrames):
            if len(items.parts) != len(self.replicas):
                raise ValueError('these frames were scattered over %d replicas, this facade has %d'
                                 % (len(items.parts), len(self.replicas)))
            futs = [self.pool.submit(rep, part, *[p[lo:hi] for p in per_item], **kw)
                    for rep, part, (lo, hi) in zip(self.replicas, items.parts, items.bounds) if part is not None]
            out = []
            for f in futs:
                out.extend(f.result())
            return out
        if isinstance(items, lib.Frames):
            raise ValueError('a batch that is already resident on one device cannot be fanned out: pass host frames')
        n = len(items)
        k = len(self.replicas)
        futs = []
        for r, rep in enumerate(self.replicas):
            lo, hi = shard_bounds(n, k, r)
            if hi > lo:
                futs.append(self.pool.submit(rep, items[lo:hi], *[p[lo:hi] for p in per_item], **kw))
        out = []
        for f in futs:                     # device order == frame order
            out.extend(f.result())
        return out


def _is_device_list(device):
    return isinstance(device, (list, tuple))


class Detection:

    def __init__(self, checkpoint=None, short_side=416, merge_method='padding', device=None, lazy=False, **model_kw):
        self.device = device
        self._fanout = None
        if _is_device_list(device):
            self._fanout = _Fanout(device, lambda d, ctx: Detection(checkpoint, short_side, merge_method, d, False,
                                                                     ctx=ctx, **model_kw))
            lazy = True
        self.detection_cls = get_class_for_checkpoint('face-detection', checkpoint)
        self.short_side = short_side
        if merge_method == 'crop':
            self._merge_error = NotImplementedError()
        elif merge_method != 'padding':
            self._merge_error = ValueError('Invalid `method` set, options are `padding` or `crop`.')
        else:
            self._merge_error = None
        self._model_kw = model_kw
        self.model = None if lazy else self.detection_cls(device=device, **model_kw)

    def __repr__(self):
        return '<Detection(%s)>' % self.detection_cls.__name__

    def _ctx(self):
        if self.model is None:
            self.model = self.detection_cls(device=self.device, **self._model_kw)
        return self.model.ctx

    def upload(self, images):
        """Frames -> HBM once, for several facade calls: `lib.Frames`, or `ShardedFrames` when built over a device list."""
        return self._fanout.upload(images) if self._fanout is not None else self._ctx().upload(images)

    def _resized(self, ctx, image_batch):
        """-> (lib.Frames at network resolution, scale).  `image_batch` may already be resident."""
        H, W = image_batch.shape[1:3]
        scale = self.short_side / min(H, W)
        if isinstance(image_batch, lib.Frames):           # possibly another thread's batch: read it, run on OUR context
            return image_batch.resize(int(H * scale), int(W * scale), ctx=ctx), scale
        src = ctx.upload(image_batch)
        try:
            return src.resize(int(H * scale), int(W * scale)), scale
        finally:
            src.free()

    def __call__(self, images, _canvas=None):
        expanded = not isinstance(images, (lib.Frames, ShardedFrames)) and _is_single(images)
        if expanded:
            images = np.expand_dims(images, 0)
        if self._fanout is not None:
            kw = {}
            if not isinstance(images, (np.ndarray, ShardedFrames)):     # list: every shard pads to the WHOLE list's canvas
                if self._merge_error is not None:
                    raise self._merge_error
                sizes = [(int(h * (self.short_side / min(h, w))), int(w * (self.short_side / min(h, w))))
                         for h, w in (np.asarray(im).shape[:2] for im in images)]
                kw['_canvas'] = (max(s[0] for s in sizes), max(s[1] for s in sizes))
            out = self._fanout(images, **kw)
            return out[0] if expanded else out
        if self.model is None:
            self.model = self.detection_cls(device=self.device, **self._model_kw)
        ctx = self.model.ctx
        if isinstance(images, (np.ndarray, lib.Frames)):
            frames, scales = self._resized(ctx, images)
            pads = None
        else:
            if self._merge_error is not None:
                raise self._merge_error
            singles, scales = [], []
            for im in images:
                f, s = self._resized(ctx, np.asarray(im)[None])
                singles.append(f)
                scales.append(s)
            frames, pads = _merge(ctx, singles, _canvas)
            for f in singles:
                f.free()
        try:
            counts, boxes, lmks, scores = self.model.detect_arrays(frames)
        finally:
            frames.free()
        if pads is None:
            # one scale for the whole batch (ndarray / resident frames): un-scale and round every detection at once -- the
            # same float32 element-wise arithmetic as the reference's per-face code (face/detection/__init__.py:59-84)
            b = np.around(boxes / scales).astype(np.int32)
            l = np.around(lmks / scales).astype(np.int32)
            out = results.detections(counts, b, l, scores, lazy=self._model_kw.get('lazy_results') or None)
            return out[0] if expanded else out
        # un-pad and un-scale one image at a time (same element-wise arithmetic and dtypes as the
        # reference's per-face code, face/detection/__init__.py:59-84,141-176), then hand out row views
        out, o = [], 0
        for i, c in enumerate(counts):
            c = int(c)
            b, l = boxes[o:o + c], lmks[o:o + c]
            top, left = pads[i][0][0], pads[i][1][0]
            b = b - np.array([left, top, left, top], np.float32)            # float32 - int stays float32
            l = l - np.array([left, top]).reshape(1, 1, 2)                  # int64 array: promotes to float64
            b = np.around(b / scales[i]).astype(np.int32)
            l = np.around(l / scales[i]).astype(np.int32)
            out.append([{'bbox': x, 'landmarks': y, 'score': z} for x, y, z in zip(b, l, scores[o:o + c])])
            o += c
        return out[0] if expanded else out


class Recognition:

    def __init__(self, checkpoint=None, device=None, lazy=False, **model_kw):
        self.device = device
        self._fanout = None
        if _is_device_list(device):
            self._fanout = _Fanout(device, lambda d, ctx: Recognition(checkpoint, d, False, ctx=ctx, **model_kw))
            lazy = True
        self.recognition_cls = get_class_for_checkpoint('face-recognition', checkpoint)
        self._model_kw = model_kw
        self.model = None if lazy else self.recognition_cls(device=device, **model_kw)

    def __repr__(self):
        return '<Recognition(%s)>' % self.recognition_cls.__name__

    def _ctx(self):
        if self.model is None:
            self.model = self.recognition_cls(device=self.device, **self._model_kw)
        return self.model.ctx

    def upload(self, images):
        return self._fanout.upload(images) if self._fanout is not None else self._ctx().upload(images)

    def __call__(self, images, faces_per_image=None):
        expanded = False
        if not isinstance(images, (lib.Frames, ShardedFrames)) and _is_single(images):
            expanded = True
            images = [images]
            faces_per_image = [[faces_per_image]] if isinstance(faces_per_image, dict) else [faces_per_image]
        if faces_per_image is not None and len(faces_per_image) != len(images):
            raise ValueError('`images` and `faces_per_image` must be of the same size, but the former is of size '
                             '%d while the latter of size %d.' % (len(images), len(faces_per_image)))
        if self._fanout is not None and faces_per_image is None and isinstance(images, ShardedFrames):
            raise ValueError('a scattered frame batch needs `faces_per_image` (pre-cropped faces go in as a host list)')
        if self._fanout is not None and faces_per_image is not None:
            out = self._fanout(images if isinstance(images, ShardedFrames) else list(images), list(faces_per_image))
            if any(len(f) for f in faces_per_image):                    # a shard without faces answers float64 (0,512)
                out = [o.astype(np.float32) if o.shape[0] == 0 else o for o in out]   # (wrapper.py:160-164); 1-way: float32
            return out[0] if expanded else out
        if self.model is None:
            self.model = self.recognition_cls(device=self.device if self._fanout is None else self.device[0],
                                              **self._model_kw)
        out = self.model.call(images, faces_per_image)
        # the reference's `isinstance(faces_per_image, dict)` test (line 85) can never be true after the
        # re-binding above, so a single image + single dict yields (1,512); reproduced here.
        return out[0] if expanded else out


class Estimation:

    def __init__(self, checkpoint=None, short_side=184, merge_method='padding', device=None, lazy=False, **model_kw):
        self.device = device
        self._fanout = None
        if _is_device_list(device):
            self._fanout = _Fanout(device, lambda d, ctx: Estimation(checkpoint, short_side, merge_method, d, False,
                                                                      ctx=ctx, **model_kw))
            lazy = True
        self.estimation_cls = get_class_for_checkpoint('pose-estimation', checkpoint)
        self.short_side = short_side
        if merge_method == 'crop':
            self._merge_error = NotImplementedError()
        elif merge_method != 'padding':
            self._merge_error = ValueError('Invalid `method` set, options are `padding` or `crop`.')
        else:
            self._merge_error = None
        self._model_kw = model_kw
        self.model = None if lazy else self.estimation_cls(device=device, short_side=short_side, **model_kw)

    def __repr__(self):
        return '<Estimation(%s)>' % self.estimation_cls.__name__

    def _ctx(self):
        if self.model is None:
            self.model = self.estimation_cls(device=self.device, short_side=self.short_side, **self._model_kw)
        return self.model.ctx

    def upload(self, images):
        return self._fanout.upload(images) if self._fanout is not None else self._ctx().upload(images)

    def __call__(self, images, _canvas=None):
        expanded = not isinstance(images, (lib.Frames, ShardedFrames)) and _is_single(images)
        if expanded:
            images = np.expand_dims(images, 0)
        if self._fanout is not None:
            kw = {}
            if not isinstance(images, (np.ndarray, ShardedFrames)):
                if self._merge_error is not None:
                    raise self._merge_error
                sizes = [np.asarray(im).shape[:2] for im in images]
                kw['_canvas'] = (max(s[0] for s in sizes), max(s[1] for s in sizes))
            out = self._fanout(images, **kw)
            return out[0] if expanded else out
        if self.model is None:
            self.model = self.estimation_cls(device=self.device, short_side=self.short_side, **self._model_kw)
        ctx = self.model.ctx
        pads = None
        resident = isinstance(images, lib.Frames)
        if resident:
            frames = images
        elif isinstance(images, np.ndarray):
            frames = ctx.upload(images)
        else:
            if self._merge_error is not None:
                raise self._merge_error
            singles = [ctx.upload(np.asarray(im)[None]) for im in images]
            frames, pads = _merge(ctx, singles, _canvas)
            for f in singles:
                f.free()
        try:
            out = self.model.call_frames(frames)
        finally:
            if not resident:
                frames.free()
        if pads is not None:
            new = []
            for poses, p in zip(out, pads):
                row = []
                for pose in poses:
                    kp = pose['keypoints'] - np.array([p[1][0], p[0][0], 0]).reshape(1, -1)
                    kp[kp[..., 2] == 0] = 0
                    row.append({'keypoints': kp, 'score': pose['score']})
                new.append(row)
            out = new
        return out[0] if expanded else out

"""Task facades with the reference's call signatures and return values:

  Detection   : terran/face/detection/__init__.py:185-287  (resize -> pad-merge -> model -> un-pad -> un-scale)
  Recognition : terran/face/recognition/__init__.py:7-90
  Estimation  : terran/pose/__init__.py:131-223

Generic pre-processing runs on the device (`ta_frames_resize` = cv2 INTER_LINEAR semantics,
`ta_frames_paste` = zero pad-merge), so a batch of frames is uploaded once.
"""
import math

import numpy as np

from . import lib
from .checkpoint import get_class_for_checkpoint


def _is_single(images):
    return not isinstance(images, (list, tuple)) and len(images.shape) == 3


def _pads(shapes):
    """Per image ((top,bottom),(left,right)) to the max size; the odd pixel goes top/left."""
    mh = max(s[0] for s in shapes)
    mw = max(s[1] for s in shapes)
    pads = []
    for h, w in shapes:
        dh, dw = max(0, (mh - h) / 2), max(0, (mw - w) / 2)
        pads.append(((int(math.ceil(dh)), int(math.floor(dh))), (int(math.ceil(dw)), int(math.floor(dw)))))
    return mh, mw, pads


def _merge(ctx, frame_list):
    """List of single-image lib.Frames -> one padded lib.Frames + pads."""
    mh, mw, pads = _pads([f.shape[1:3] for f in frame_list])
    canvas = lib.Frames.zeros(ctx, len(frame_list), mh, mw)
    for i, (f, p) in enumerate(zip(frame_list, pads)):
        canvas.paste(f, 0, i, p[0][0], p[1][0])
    return canvas, pads


class Detection:

    def __init__(self, checkpoint=None, short_side=416, merge_method='padding', device=None, lazy=False, **model_kw):
        self.device = device
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

    def __call__(self, images):
        expanded = not isinstance(images, lib.Frames) and _is_single(images)
        if expanded:
            images = np.expand_dims(images, 0)
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
            frames, pads = _merge(ctx, singles)
            for f in singles:
                f.free()
        try:
            counts, boxes, lmks, scores = self.model.detect_arrays(frames)
        finally:
            frames.free()
        if not isinstance(scales, list):
            scales = [scales] * len(counts)
        # un-pad and un-scale one image at a time (same element-wise arithmetic and dtypes as the
        # reference's per-face code, face/detection/__init__.py:59-84,141-176), then hand out row views
        out, o = [], 0
        for i, c in enumerate(counts):
            c = int(c)
            b, l = boxes[o:o + c], lmks[o:o + c]
            if pads is not None:
                top, left = pads[i][0][0], pads[i][1][0]
                b = b - np.array([left, top, left, top], np.float32)            # float32 - int stays float32
                l = l - np.array([left, top]).reshape(1, 1, 2)                  # int64 array: promotes to float64
            b = np.around(b / scales[i]).astype(np.int32)
            l = np.around(l / scales[i]).astype(np.int32)
            out.append([{'bbox': b[k], 'landmarks': l[k], 'score': scores[o + k]} for k in range(c)])
            o += c
        return out[0] if expanded else out


class Recognition:

    def __init__(self, checkpoint=None, device=None, lazy=False, **model_kw):
        self.device = device
        self.recognition_cls = get_class_for_checkpoint('face-recognition', checkpoint)
        self._model_kw = model_kw
        self.model = None if lazy else self.recognition_cls(device=device, **model_kw)

    def __repr__(self):
        return '<Recognition(%s)>' % self.recognition_cls.__name__

    def __call__(self, images, faces_per_image=None):
        expanded = False
        if _is_single(images):
            expanded = True
            images = [images]
            faces_per_image = [[faces_per_image]] if isinstance(faces_per_image, dict) else [faces_per_image]
        if faces_per_image is not None and len(faces_per_image) != len(images):
            raise ValueError('`images` and `faces_per_image` must be of the same size, but the former is of size '
                             '%d while the latter of size %d.' % (len(images), len(faces_per_image)))
        if self.model is None:
            self.model = self.recognition_cls(device=self.device, **self._model_kw)
        out = self.model.call(images, faces_per_image)
        # the reference's `isinstance(faces_per_image, dict)` test (line 85) can never be true after the
        # re-binding above, so a single image + single dict yields (1,512); reproduced here.
        return out[0] if expanded else out


class Estimation:

    def __init__(self, checkpoint=None, short_side=184, merge_method='padding', device=None, lazy=False, **model_kw):
        self.device = device
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

    def __call__(self, images):
        expanded = not isinstance(images, lib.Frames) and _is_single(images)
        if expanded:
            images = np.expand_dims(images, 0)
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
            frames, pads = _merge(ctx, singles)
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

"""`ArcFace` -- drop-in for terran/face/recognition/arcface/wrapper.py:102-184 on MI355X."""
import os
import threading
import warnings

import numpy as np

from . import lib, pack, runtime

# Target landmark positions on the 112x112 crop: the 96x112 template with x shifted by 8,
# built in float32 exactly like the reference does (arcface/wrapper.py:39-48).
_TEMPLATE = np.array([[30.2946, 51.6963], [65.5318, 51.5014], [48.0252, 71.7366],
                      [33.5493, 92.3655], [62.7299, 92.2041]], dtype=np.float32)
_TEMPLATE[:, 0] += 8.0
_IDENTITY = np.array([1.0, 0.0, 0.0, 0.0, 1.0, 0.0])


def align_matrices(landmarks):
    """(n,5,2) landmark sets -> (n,6) inverse 2x3 similarities (float64, PIL AFFINE convention) taking crop pixels to
    image coordinates: least-squares similarity landmarks -> template (Umeyama; closed form in 2-D, proper rotations
    only), inverted in closed form.  arcface/wrapper.py:50-61.  One vectorised pass for all faces of a frame batch
    (element-wise float64 arithmetic in a fixed association order, so a face's matrix does not depend on its batch)."""
    p = np.asarray(landmarks).astype(np.float32).astype(np.float64).reshape(-1, 5, 2)
    q = _TEMPLATE.astype(np.float64)
    px, py = p[:, :, 0], p[:, :, 1]
    pmx = (((px[:, 0] + px[:, 1]) + px[:, 2]) + px[:, 3] + px[:, 4]) / 5.0
    pmy = (((py[:, 0] + py[:, 1]) + py[:, 2]) + py[:, 3] + py[:, 4]) / 5.0
    qm = q.mean(0)
    qd = q - qm
    dx, dy = px - pmx[:, None], py - pmy[:, None]
    var = np.zeros(len(p))
    a = np.zeros(len(p))
    b = np.zeros(len(p))
    for k in range(5):                                             # fixed order over the five points
        var = var + (dx[:, k] * dx[:, k] + dy[:, k] * dy[:, k])
        a = a + (dx[:, k] * qd[k, 0] + dy[:, k] * qd[k, 1])        # s*cos (x n var)
        b = b + (dx[:, k] * qd[k, 1] - dy[:, k] * qd[k, 0])        # s*sin (x n var)
    a, b = a / var, b / var                                        # the 1/n factors cancel
    tx = qm[0] - (a * pmx - b * pmy)
    ty = qm[1] - (b * pmx + a * pmy)
    # inverse of [[a,-b,tx],[b,a,ty],[0,0,1]]
    d = a * a + b * b
    ia, ib = a / d, b / d
    return np.stack([ia, ib, -(ia * tx + ib * ty), -ib, ia, ib * tx - ia * ty], axis=1)


def align_matrix(landmarks):
    """One face: 6 float64 (see align_matrices)."""
    return align_matrices(np.asarray(landmarks)[None])[0]


# -- the load-time guard of the two-product mode ------------------------------------------------------------------------
# 'f16x2' feeds every activation into the contractions as its hi half (11 bits).  How far that moves the unit embedding is a
# property of the WEIGHTS (1.8e-4 on the seeded ones, 8.2e-4 on the wild-statistics ones, never measured on a trained
# checkpoint), and the reference's contract is fp32 (arcface/wrapper.py:166-176).  So the mode is not taken on trust: when
# an 'f16x2' embedder is loaded, GUARD_CROPS fixed calibration crops are embedded in 'f16x2' and in 'f16x3' (float32-grade)
# and the two-product program is kept only if no unit-embedding component moves by more than GUARD_TOL -- half of
# north_star's 1e-3 bar.  Otherwise the model IS the 'f16x3' one, with one warning.
GUARD_TOL = 5e-4
GUARD_CROPS = 32
_guard_memo = {}                 # id(state dict) -> (state dict, decision): dict states (tests, bench, pipeline lanes)
_guard_lock = threading.Lock()


def calibration_crops(n=GUARD_CROPS):
    """(n,3,112,112) uint8 BGR crops, fixed (seeds 4242 / 4243): half smooth colour fields + sensor-like noise
    (terran_amd.synth.frames: the statistics of an aligned face crop as far as the first convolutions care), half uniform
    byte noise (the worst input for an 11-bit activation path: every frequency at full amplitude)."""
    from . import synth
    smooth = synth.frames(4242, n - n // 2, 112, 112).transpose(0, 3, 1, 2)
    noise = np.random.default_rng(4243).integers(0, 256, (n // 2, 3, 112, 112), dtype=np.uint8)
    return np.ascontiguousarray(np.concatenate([smooth, noise]))


def guard_f16x2(ctx, state, tol=None):
    """-> {'selected': 'f16x2' | 'f16x3', 'max_abs_diff': float, 'tol': float, 'crops': int} for these weights.
    Cached: in the f16x2 repack cache of a checkpoint file (Program.extra, rewritten once), per state-dict object otherwise."""
    tol = GUARD_TOL if tol is None else tol
    prog2 = runtime.packed_program('arcface', state, 'f16x2')
    hit = prog2.extra.get('f16x2_guard')
    if hit and hit.get('tol') == tol and hit.get('crops') == GUARD_CROPS:
        return hit
    sd = state if isinstance(state, dict) else None
    if sd is not None:
        with _guard_lock:
            m = _guard_memo.get((id(sd), tol))
            if m is not None and m[0] is sd:
                return m[1]
    crops = calibration_crops()
    outs = {}
    for mode, prog in (('f16x2', prog2), ('f16x3', runtime.packed_program('arcface', state, 'f16x3'))):
        model = lib.Model(ctx, prog)
        try:
            out = np.empty((len(crops), 512), np.float32)
            try:
                ctx.check(ctx.lib.ta_arcface_embed_crops(model.h, lib.ptr(crops), len(crops), 1, lib.ptr(out)))
            except lib.TerranAmdError as e:
                if e.code != lib.E_RANGE:
                    raise
                out = None                      # the calibration crops leave the half-float range: not a mode to opt into
            outs[mode] = out
        finally:
            model.free()
    if outs['f16x2'] is None or outs['f16x3'] is None:
        diff = float('inf')
    else:
        diff = float(np.abs(outs['f16x2'] - outs['f16x3']).max())
    res = {'selected': 'f16x2' if diff <= tol else 'f16x3', 'max_abs_diff': diff if np.isfinite(diff) else None, 'tol': tol,
           'crops': GUARD_CROPS}
    prog2.extra['f16x2_guard'] = res
    cache = getattr(prog2, '_cache_path', None)
    if cache:
        try:
            prog2.save_cache(cache)
        except OSError:
            pass                                # read-only checkpoint dir: calibrate again next time
    if sd is not None:
        with _guard_lock:
            _guard_memo[(id(sd), tol)] = (sd, res)
            while len(_guard_memo) > 8:
                _guard_memo.pop(next(iter(_guard_memo)))
    return res


class ArcFace(runtime.RangeFallback):

    def __init__(self, device=None, image_side=112, state=None, ctx=None, precision=None, guard=None):
        """guard: run the load-time calibration of precision='f16x2' (guard_f16x2; None = yes unless $TERRAN_AMD_F16X2_UNGUARDED)."""
        if image_side != 112:
            raise ValueError('the ArcFace-R100 head is a 25088->512 linear layer: image_side must be 112')
        self.device = device
        self.precision = runtime.resolve_precision(precision)
        self.image_side = image_side
        self.ctx = ctx if ctx is not None else runtime.get_context(device)     # ctx: an extra stream on the same GPU
        self.guard = None
        if guard is None:
            guard = not os.environ.get('TERRAN_AMD_F16X2_UNGUARDED')
        if self.precision == 'f16x2' and guard:
            self.guard = guard_f16x2(self.ctx, state)
            if self.guard['selected'] != 'f16x2':
                warnings.warn('terran_amd: precision="f16x2" moves the unit embedding of these ArcFace weights by %s on the %d '
                              'calibration crops (limit %g): the embedder runs in "f16x3" (float32-grade) instead'
                              % ('more than the half-float range allows' if self.guard['max_abs_diff'] is None
                                 else '%.2e' % self.guard['max_abs_diff'], self.guard['crops'], self.guard['tol']),
                              RuntimeWarning, stacklevel=2)
                self.precision = 'f16x3'
        self.model = lib.Model(self.ctx, runtime.packed_program('arcface', state, self.precision))
        self._init_fallback('arcface', state)

    # -- device entry points ---------------------------------------------------------------
    def embed_crops(self, crops, normalize=True):
        crops = np.ascontiguousarray(crops, dtype=np.uint8)
        out = np.empty((crops.shape[0], 512), np.float32)
        self._with_fallback(lambda model: self.ctx.check(self.ctx.lib.ta_arcface_embed_crops(
            model.h, lib.ptr(crops), crops.shape[0], int(normalize), lib.ptr(out))))
        return out

    def embed_faces(self, frames, frame_index, matrices, normalize=True, return_crops=False):
        """frames: lib.Frames; face k is cut from frames[frame_index[k]] with matrices[k] (6 doubles)."""
        idx = np.ascontiguousarray(frame_index, dtype=np.int32)
        mats = np.ascontiguousarray(matrices, dtype=np.float64).reshape(-1, 6)
        n = idx.shape[0]
        out = np.empty((n, 512), np.float32)
        crops = np.empty((n, 3, 112, 112), np.uint8) if return_crops else None
        self._with_fallback(lambda model: self.ctx.check(self.ctx.lib.ta_arcface_embed_faces(
            model.h, frames.h, lib.ptr(idx), lib.ptr(mats), n, int(normalize), lib.ptr(out), lib.ptr(crops))))
        return (out, crops) if return_crops else out

    def embed_faces_multi(self, frames_list, source_index, frame_index, matrices, normalize=True):
        """Faces cut from SEVERAL resident batches in one launch: face k comes from frames_list[source_index[k]], image
        frame_index[k] of it (ta_arcface_embed_faces_multi)."""
        import ctypes
        src = np.ascontiguousarray(source_index, dtype=np.int32)
        idx = np.ascontiguousarray(frame_index, dtype=np.int32)
        mats = np.ascontiguousarray(matrices, dtype=np.float64).reshape(-1, 6)
        n = idx.shape[0]
        out = np.empty((n, 512), np.float32)
        handles = (ctypes.c_void_p * len(frames_list))(*[f.h for f in frames_list])
        self._with_fallback(lambda model: self.ctx.check(self.ctx.lib.ta_arcface_embed_faces_multi(
            model.h, handles, len(frames_list), lib.ptr(src), lib.ptr(idx), lib.ptr(mats), n, int(normalize), lib.ptr(out), None)))
        return out

    def call_multi(self, items):
        """items: [(lib.Frames, faces_per_image), ...] -> [what `call(frames, faces_per_image)` returns, ...], all faces of
        all items embedded in ONE launch (terran_amd.pipeline's embed worker)."""
        counts = [[len(f) for f in faces] for _, faces in items]
        if sum(sum(c) for c in counts) == 0:
            return [[np.empty((0, 512)) for _ in c] for c in counts]
        lms = np.array([face['landmarks'] for _, faces in items for f in faces for face in f])
        src = np.concatenate([np.full(sum(c), s, np.int32) for s, c in enumerate(counts)])
        idx = np.concatenate([np.repeat(np.arange(len(c)), c) for c in counts])
        feats = self.embed_faces_multi([fr for fr, _ in items], src, idx, align_matrices(lms))
        out, o = [], 0
        for c in counts:
            n = sum(c)
            if n == 0:
                out.append([np.empty((0, 512)) for _ in c])                 # float64, as wrapper.py:160-164
            else:
                out.append(np.split(feats[o:o + n], np.cumsum(c)[:-1], axis=0))
            o += n
        return out

    # -- the reference call ------------------------------------------------------------------
    def call(self, images, faces_per_image=None):
        """images: list of (H_i,W_i,3) uint8 RGB (or an (N,H,W,3) array); faces_per_image: list of
        lists of dicts with 'landmarks'.  Returns a list with one float32 (N_i,512) L2-normalised
        array per image, or a single array when `faces_per_image is None`."""
        if isinstance(images, lib.Frames):          # resident batch: all faces in one device call
            counts = [len(f) for f in faces_per_image]
            if sum(counts) == 0:
                return [np.empty((0, 512)) for _ in counts]
            idx = np.repeat(np.arange(len(counts)), counts)
            mats = align_matrices(np.array([face['landmarks'] for f in faces_per_image for face in f]))
            return np.split(self.embed_faces(images, idx, mats), np.cumsum(counts)[:-1], axis=0)
        n_images = len(images)
        if faces_per_image is not None:
            counts = [len(f) for f in faces_per_image]
            total = sum(counts)
            if total == 0:
                return [np.empty((0, 512)) for _ in range(n_images)]      # float64, as wrapper.py:160-164
            feats = np.empty((total, 512), np.float32)
            # one device batch per distinct image size (video batches: exactly one)
            groups = {}
            k = 0
            for i, (image, faces) in enumerate(zip(images, faces_per_image)):
                for face in faces:
                    groups.setdefault(np.asarray(image).shape[:2], []).append((i, k, face))
                    k += 1
            for shape, items in groups.items():
                img_ids = sorted({i for i, _, _ in items})
                slot = {i: s for s, i in enumerate(img_ids)}
                frames = self.ctx.upload(np.stack([np.asarray(images[i]) for i in img_ids]))
                try:
                    mats = align_matrices(np.array([face['landmarks'] for _, _, face in items]))
                    out = self.embed_faces(frames, [slot[i] for i, _, _ in items], mats)
                finally:
                    frames.free()
                feats[[kk for _, kk, _ in items]] = out
            return np.split(feats, np.cumsum(counts)[:-1], axis=0)
        if n_images == 0:
            return []
        # no landmarks: aspect-preserving Pillow-bicubic resize + centre pad (wrapper.py:75-99), on device
        canvas = lib.Frames.zeros(self.ctx, n_images, 112, 112)
        try:
            for i, image in enumerate(images):
                image = np.asarray(image)
                h, w = image.shape[:2]
                scale = 112 / max(w, h)
                nw, nh = int(w * scale), int(h * scale)
                src = self.ctx.upload(image[None])
                try:
                    small = src.resize_bicubic(nh, nw)
                    try:
                        canvas.paste(small, 0, i, int((112 - nh) / 2), int((112 - nw) / 2))
                    finally:
                        small.free()
                finally:
                    src.free()
            return self.embed_faces(canvas, np.arange(n_images), np.tile(_IDENTITY, (n_images, 1)))
        finally:
            canvas.free()

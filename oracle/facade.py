"""ORACLE (test infrastructure only): numpy restatement of the task facades'
generic pre/post-processing.  Never imported by the product path.

Follows:
  cv2_resize_linear     : cv2.resize(..., INTER_LINEAR) for uint8 (opencv-python-headless,
                          unpinned in setup.py:19, NOT vendored; call sites
                          terran/face/detection/__init__.py:33-38,49-53 and
                          terran/pose/openpose/wrapper.py:106-111).  Restates OpenCV's classic
                          fixed-point bilinear (11-bit coefficients, two-pass).  PARITY UNPINNED:
                          cv2 is not importable in the build container and the reference holds no
                          resize vectors.
  det_resize_in/out     : terran/face/detection/__init__.py:13-86
  merge_in              : terran/face/detection/__init__.py:96-139 == terran/pose/__init__.py:48-88
  det_merge_out         : terran/face/detection/__init__.py:141-176
  pose_merge_out        : terran/pose/__init__.py:94-122
  pose_resize           : terran/pose/openpose/wrapper.py:93-113
"""
import math

import numpy as np

_COEF_BITS = 11
_COEF_SCALE = 1 << _COEF_BITS


def _axis_coeffs(src, dst):
    """OpenCV resize index/coefficient tables for one axis (no edge clamping of
    the fractional part: used for rows)."""
    scale = 1.0 / (dst / src)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _to_short(c):
    # saturate_cast<short>(float): round half to even, saturate
    return np.clip(np.rint(c.astype(np.float32) * np.float32(_COEF_SCALE)), -32768, 32767).astype(np.int64)


def cv2_resize_linear(src, dsize):
    """src (H,W,C) uint8, dsize=(dst_w,dst_h) -> (dst_h,dst_w,C) uint8."""
    src = np.asarray(src)
    H, W = src.shape[:2]
    dw, dh = int(dsize[0]), int(dsize[1])
    sx, fx = _axis_coeffs(W, dw)
    lo = sx < 0
    fx = np.where(lo, np.float32(0), fx)
    sx = np.where(lo, 0, sx)
    hi = sx >= W - 1
    fx = np.where(hi, np.float32(0), fx)
    sx = np.where(hi, W - 1, sx)
    a0 = _to_short(np.float32(1) - fx)
    a1 = _to_short(fx)
    sx1 = np.minimum(sx + 1, W - 1)

    sy, fy = _axis_coeffs(H, dh)
    b0 = _to_short(np.float32(1) - fy)
    b1 = _to_short(fy)
    y0 = np.clip(sy, 0, H - 1)
    y1 = np.clip(sy + 1, 0, H - 1)

    s = src.astype(np.int64)
    # horizontal pass on every source row (int32 intermediates, scale 2^11)
    hbuf = s[:, sx, :] * a0[None, :, None] + s[:, sx1, :] * a1[None, :, None]
    r0 = hbuf[y0]
    r1 = hbuf[y1]
    out = ((((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2)
    return np.clip(out, 0, 255).astype(np.uint8)


# ---- detection facade ----------------------------------------------------------
def det_resize_in(images, short_side=416):
    """ndarray (N,H,W,3) -> (resized ndarray, scale); list -> (list, list of scales)."""
    if isinstance(images, np.ndarray):
        H, W = images.shape[1:3]
        scale = short_side / min(H, W)
        new_size = (int(W * scale), int(H * scale))
        out = np.empty((images.shape[0], new_size[1], new_size[0], images.shape[3]), images.dtype)
        for i, im in enumerate(images):
            out[i] = cv2_resize_linear(im, new_size)
        return out, scale
    resized, scales = [], []
    for im in images:
        H, W = im.shape[:2]
        scale = short_side / min(H, W)
        resized.append(cv2_resize_linear(im, (int(W * scale), int(H * scale))))
        scales.append(scale)
    return resized, scales


def det_resize_out(faces_per_image, scales):
    if not isinstance(scales, list):
        scales = [scales] * len(faces_per_image)
    out = []
    for faces, scale in zip(faces_per_image, scales):
        out.append([{
            'bbox': np.around(f['bbox'] / scale).astype(np.int32),
            'landmarks': np.around(f['landmarks'] / scale).astype(np.int32),
            'score': f['score'],
        } for f in faces])
    return out


def merge_in(images):
    """List of (H_i,W_i,3) -> zero-padded batch; the odd pixel goes top/left."""
    if isinstance(images, np.ndarray):
        return images, {'merged': False}
    mh = max(a.shape[0] for a in images)
    mw = max(a.shape[1] for a in images)
    padded = np.zeros((len(images), mh, mw, 3), np.uint8)
    pads = []
    for i, im in enumerate(images):
        dh = max(0, (mh - im.shape[0]) / 2)
        dw = max(0, (mw - im.shape[1]) / 2)
        top, left = int(math.ceil(dh)), int(math.ceil(dw))
        padded[i, top:top + im.shape[0], left:left + im.shape[1]] = im
        pads.append([(top, int(math.floor(dh))), (left, int(math.floor(dw))), (0, 0)])
    return padded, {'merged': True, 'pads_per_image': pads}


def det_merge_out(faces_per_image, params):
    if not params['merged']:
        return faces_per_image
    out = []
    for faces, pads in zip(faces_per_image, params['pads_per_image']):
        top, left = pads[0][0], pads[1][0]
        out.append([{
            'bbox': np.array([f['bbox'][0] - left, f['bbox'][1] - top,
                              f['bbox'][2] - left, f['bbox'][3] - top]),
            'landmarks': f['landmarks'] - np.array([left, top]).reshape(1, -1),
            'score': f['score'],
        } for f in faces])
    return out


# ---- pose facade ---------------------------------------------------------------
def pose_resize(images, short_side=184):
    H, W = images.shape[1:3]
    scale = short_side / min(H, W)
    new_size = (int(W * scale), int(H * scale))
    out = np.empty((images.shape[0], new_size[1], new_size[0], images.shape[3]), images.dtype)
    for i, im in enumerate(images):
        out[i] = cv2_resize_linear(im, new_size)
    return out, scale


def pose_merge_out(poses_per_image, params):
    if not params['merged']:
        return poses_per_image
    out = []
    for poses, pads in zip(poses_per_image, params['pads_per_image']):
        new = []
        for p in poses:
            kp = p['keypoints'] - np.array([pads[1][0], pads[0][0], 0]).reshape(1, -1)
            kp[kp[..., 2] == 0] = 0
            new.append({'keypoints': kp, 'score': p['score']})
        out.append(new)
    return out

"""Seeded synthetic inputs shared by tests, fixtures and bench.py.

No datasets or checkpoints are reachable offline, so workloads are synthetic:
uint8 frames (low-frequency noise + blobs) and, for the pose-grouping stage,
network-resolution heat-maps / part-affinity fields drawn from a stick-figure
generator (random-weight OpenPose outputs are structureless).
"""
import numpy as np

from .arch import MAP_IDX, LIMBSEQ

# canonical skeleton in a unit box: (x, y) per OpenPose-18 keypoint index
# (terran/pose/__init__.py:13-36 gives the index meaning).
_SKELETON = np.array([
    [0.50, 0.08],  # nose
    [0.50, 0.22],  # neck
    [0.36, 0.22], [0.30, 0.42], [0.27, 0.60],   # r shoulder/elbow/hand
    [0.64, 0.22], [0.70, 0.42], [0.73, 0.60],   # l shoulder/elbow/hand
    [0.42, 0.56], [0.40, 0.76], [0.39, 0.96],   # r hip/knee/foot
    [0.58, 0.56], [0.60, 0.76], [0.61, 0.96],   # l hip/knee/foot
    [0.46, 0.05], [0.54, 0.05],                 # eyes
    [0.41, 0.08], [0.59, 0.08],                 # ears
], dtype=np.float64)


def frames(seed, n, h, w):
    """(n,h,w,3) uint8: smooth low-frequency colour field + a few bright blobs."""
    rng = np.random.default_rng(seed)
    gh, gw = max(2, h // 32), max(2, w // 32)
    out = np.empty((n, h, w, 3), np.uint8)
    ys = np.linspace(0, gh - 1, h)
    xs = np.linspace(0, gw - 1, w)
    y0 = np.floor(ys).astype(int).clip(0, gh - 2)
    x0 = np.floor(xs).astype(int).clip(0, gw - 2)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    for i in range(n):
        g = rng.uniform(0, 255, (gh, gw, 3))
        a = g[y0][:, x0] * (1 - fx) + g[y0][:, x0 + 1] * fx
        b = g[y0 + 1][:, x0] * (1 - fx) + g[y0 + 1][:, x0 + 1] * fx
        img = a * (1 - fy) + b * fy
        img += rng.normal(0, 6.0, img.shape)
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def people(seed, n_people, h, w, drop_prob=0.1, jitter=0.02):
    """Keypoints for `n_people` stick figures in an (h,w) map: (P,18,2) float (x,y)
    and (P,18) bool visibility."""
    rng = np.random.default_rng(seed)
    kps = np.empty((n_people, 18, 2))
    vis = np.ones((n_people, 18), bool)
    for p in range(n_people):
        size = rng.uniform(0.45, 0.9) * h
        cx = rng.uniform(0.15, 0.85) * w
        cy = rng.uniform(0.05, 0.95 - 0.9 * size / h) * h if size < h else 0.0
        sk = _SKELETON + rng.normal(0, jitter, _SKELETON.shape)
        kps[p, :, 0] = cx + (sk[:, 0] - 0.5) * size * 0.55
        kps[p, :, 1] = cy + sk[:, 1] * size
        vis[p] = rng.uniform(size=18) >= drop_prob
        vis[p] &= (kps[p, :, 0] > 1) & (kps[p, :, 0] < w - 2) & (kps[p, :, 1] > 1) & (kps[p, :, 1] < h - 2)
    return kps, vis


def pose_maps(seed, n_people, h, w, sigma=0.9, paf_width=0.8, noise=0.01, **kw):
    """Network-resolution maps for one image: heatmaps (19,h,w), pafs (38,h,w)
    float32.  Heat-map = max of Gaussians; PAF = unit limb vector within
    `paf_width` cells of the segment (averaged where limbs overlap)."""
    rng = np.random.default_rng(seed + 7919)
    kps, vis = people(seed, n_people, h, w, **kw)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    hm = np.zeros((19, h, w))
    for p in range(n_people):
        for k in range(18):
            if vis[p, k]:
                g = np.exp(-((xx - kps[p, k, 0]) ** 2 + (yy - kps[p, k, 1]) ** 2) / (2 * sigma ** 2))
                hm[k] = np.maximum(hm[k], g)
    hm[18] = 1.0 - hm[:18].max(0)
    paf = np.zeros((38, h, w))
    cnt = np.zeros((19, h, w))
    for l, (a, b) in enumerate(LIMBSEQ):
        cxi, cyi = MAP_IDX[l][0] - 19, MAP_IDX[l][1] - 19
        for p in range(n_people):
            if not (vis[p, a - 1] and vis[p, b - 1]):
                continue
            p0, p1 = kps[p, a - 1], kps[p, b - 1]
            v = p1 - p0
            L = np.hypot(*v)
            if L < 1e-6:
                continue
            u = v / L
            rx, ry = xx - p0[0], yy - p0[1]
            along = rx * u[0] + ry * u[1]
            perp = np.abs(rx * u[1] - ry * u[0])
            m = (along >= -0.5) & (along <= L + 0.5) & (perp <= paf_width)
            paf[cxi][m] += u[0]
            paf[cyi][m] += u[1]
            cnt[l][m] += 1
        nz = cnt[l] > 0
        paf[cxi][nz] /= cnt[l][nz]
        paf[cyi][nz] /= cnt[l][nz]
    hm += rng.normal(0, noise, hm.shape)
    paf += rng.normal(0, noise, paf.shape)
    return hm.astype(np.float32), paf.astype(np.float32)


def pose_maps_batch(seed, n_images, n_people, h, w, **kw):
    hms, pafs = zip(*[pose_maps(seed + 31 * i, n_people, h, w, **kw) for i in range(n_images)])
    return np.stack(hms), np.stack(pafs)


def landmarks(seed, n, h, w):
    """n plausible 5-point landmark sets (n,5,2) float32 inside an (h,w) image."""
    rng = np.random.default_rng(seed)
    tmpl = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366],
                     [41.5493, 92.3655], [70.7299, 92.2041]]) - 56.0
    out = np.empty((n, 5, 2), np.float32)
    for i in range(n):
        s = rng.uniform(0.4, 1.6)
        th = rng.uniform(-0.4, 0.4)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        c = np.array([rng.uniform(0.2, 0.8) * w, rng.uniform(0.2, 0.8) * h])
        out[i] = tmpl @ R.T * s + c + rng.normal(0, 0.7, (5, 2))
    return out


# ---- frames that carry pose maps (for weights.make_openpose_decoder_state) -------------------------------
def _position_code(n):
    """Pixel value of the G / B code plane along an axis of n pixels: 32 c + 16 with
    c = 4 (p & 1) + 2 (p >> 1 & 1) + (p >> 2 & 1): the finest position bit is the most significant."""
    p = np.arange(n)
    return (32 * (4 * (p & 1) + 2 * ((p >> 1) & 1) + ((p >> 2) & 1)) + 16).astype(np.uint8)


def encode_pose_maps(hm, paf, H, W):
    """Network-resolution maps (19,h,w), (38,h,w) -> (H,W,3) uint8 frame at network INPUT resolution (h = H // 8,
    w = W // 8) whose 8 x 8 block of every map cell holds that cell's 57 values in the R plane (PAF channel k at
    in-block position q = k, heat-map channel j at q = 38 + j, q = 8 dy + dx; PAF stored as (p + 1) / 2) and the
    row / column position codes in G / B.  Also returns the maps the decoder network will reproduce from it
    (8-bit quantised): (hm_q (19,h,w), paf_q (38,h,w)) float64."""
    h, w = H // 8, W // 8
    assert hm.shape == (19, h, w) and paf.shape == (38, h, w)
    data = np.concatenate([(np.clip(paf, -1, 1) + 1.0) / 2.0, np.clip(hm, 0, 1)])        # (57,h,w) in [0,1]
    pix = np.rint(data * 255.0).astype(np.uint8)
    frame = np.zeros((H, W, 3), np.uint8)
    blocks = np.zeros((h, 8, w, 8), np.uint8)
    for q in range(57):
        blocks[:, q // 8, :, q % 8] = pix[q]
    frame[:8 * h, :8 * w, 0] = blocks.reshape(8 * h, 8 * w)
    frame[:, :, 1] = _position_code(H)[:, None]
    frame[:, :, 2] = _position_code(W)[None, :]
    dq = pix.astype(np.float64) / 255.0
    return frame, dq[38:], 2.0 * dq[:38] - 1.0


def pose_code_frames(seed, n, H, W, n_people, **kw):
    """(n,H,W,3) uint8 frames at network input resolution carrying `pose_maps(seed + 31 i, n_people, H // 8, W // 8)`."""
    out = np.empty((n, H, W, 3), np.uint8)
    for i in range(n):
        hm, paf = pose_maps(seed + 31 * i, n_people, H // 8, W // 8, **kw)
        out[i] = encode_pose_maps(hm, paf, H, W)[0]
    return out


def _resize_footprint_owner(src, dst):
    """For cv2-style bilinear resize src -> dst pixels along one axis: owner[s] = the destination index whose two
    source taps {floor((d + .5) src/dst - .5), +1} lie closest to source pixel s (taps of different destinations do not
    overlap when src / dst >= 2)."""
    scale = 1.0 / (dst / src)
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    centre = np.floor(f).astype(np.float64) + 0.5
    s = np.arange(src, dtype=np.float64)
    return np.abs(s[:, None] - centre[None, :]).argmin(1)


def upscale_for_resize(frames, H_big, W_big):
    """(n,H,W,3) -> (n,H_big,W_big,3) such that the wrappers' bilinear short-side resize back to (H,W) returns `frames`
    EXACTLY (every destination pixel's 2 x 2 source taps hold its value); needs H_big >= 2 H and W_big >= 2 W."""
    n, H, W = frames.shape[:3]
    assert H_big >= 2 * H and W_big >= 2 * W
    ry, rx = _resize_footprint_owner(H_big, H), _resize_footprint_owner(W_big, W)
    return np.ascontiguousarray(frames[:, ry][:, :, rx])


# ---- adversarial pose maps (noise-free: exact plateaus, exact score ties, zero-length limbs) ------------------
def pose_maps_adversarial(kind, seed, h, w):
    """Noise-free network-resolution maps (heatmaps (19,h,w), pafs (38,h,w)) built to hit the corners of the grouping
    stage that noisy maps never reach (openpose/wrapper.py:235-262, 299-300, 335-366):
      'plateau'    : a block of equal heat-map cells -> flat tops in the x8 map, where `>=` yields several peaks
      'twins'      : one figure pasted twice at an integer cell offset -> bit-identical patches, exact score ties in
                     the candidate sort of every limb
      'coincident' : nose / neck and eye pairs drawn at the same cell -> peaks of different parts at one pixel,
                     zero-length limb vectors (0/0 = NaN scores, rejected by the `> 0` criterion)
    """
    rng = np.random.default_rng(seed)
    if kind == 'twins':
        hw = w // 2
        hm1, paf1 = pose_maps(seed, 1, h, hw, noise=0.0, drop_prob=0.0)
        hm = np.concatenate([hm1, hm1], axis=2)
        paf = np.concatenate([paf1, paf1], axis=2)
        if hm.shape[2] < w:
            hm = np.pad(hm, ((0, 0), (0, 0), (0, w - hm.shape[2])))
            paf = np.pad(paf, ((0, 0), (0, 0), (0, w - paf.shape[2])))
        hm[18] = 1.0 - hm[:18].max(0)
        return hm.astype(np.float32), paf.astype(np.float32)
    hm, paf = pose_maps(seed, 2, h, w, noise=0.0, drop_prob=0.0)
    if kind == 'plateau':
        for part in (0, 5, 9):
            y0, x0 = int(rng.integers(1, h - 4)), int(rng.integers(1, w - 5))
            hm[part, y0:y0 + 2, x0:x0 + 3] = np.float32(0.75)                  # equal cells: flat top after x8 bicubic
        hm[3, 2:4, 2:4] = np.float32(0.1)                                      # a plateau exactly at the 0.1 threshold
        return hm, paf
    if kind == 'coincident':
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        cy, cx = h // 2, w // 2
        g = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 0.9 ** 2)).astype(np.float32)
        for part in (0, 1, 14, 15):                                            # nose, neck, both eyes at ONE cell
            hm[part] = np.maximum(hm[part], g)
        return hm, paf
    raise ValueError(kind)

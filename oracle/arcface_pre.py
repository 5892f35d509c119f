"""ORACLE (test infrastructure only): numpy restatement of ArcFace pre/post
processing.  Never imported by the product path (`terran_amd/`).

Follows:
  umeyama            : skimage.transform.SimilarityTransform.estimate (scikit-image, unpinned in
                       setup.py, NOT vendored in /root/reference; call site
                       terran/face/recognition/arcface/wrapper.py:52-53).  Restates Umeyama (1991)
                       least-squares similarity in float64.  PARITY UNPINNED (skimage absent here).
  face_template / align_matrix / preprocess_face
                     : terran/face/recognition/arcface/wrapper.py:22-72
  pil_affine_bilinear: Pillow Image.transform(AFFINE, BILINEAR, fillcolor=0) (call site
                       wrapper.py:63-69).  Pillow IS importable in the build container, so this
                       restatement is pinned bit-exactly against real PIL by
                       tests/golden/make_golden.py -> tests/golden/arcface_pre.npz.
  preprocess_face_no_landmarks / pil_resize_bicubic
                     : wrapper.py:75-99 (PIL `Image.resize` default filter = BICUBIC), pinned
                       against real PIL the same way.
  l2_normalize       : sklearn.preprocessing.normalize(axis=1) (wrapper.py:176)
  cosine_distance    : scipy.spatial.distance.cosine (examples/match.py:38)
"""
import numpy as np

# Target landmark positions on the 112x112 crop (wrapper.py:39-48: the 96x112
# template with x shifted by +8 for a 112-wide crop).
FACE_TEMPLATE_112 = np.array([
    [30.2946, 51.6963],
    [65.5318, 51.5014],
    [48.0252, 71.7366],
    [33.5493, 92.3655],
    [62.7299, 92.2041],
], dtype=np.float32)
FACE_TEMPLATE_112[:, 0] += 8.0        # float32 add, as the reference does it (wrapper.py:47-48)


def umeyama(src, dst):
    """3x3 float64 similarity T (rotation, uniform scale, translation; no
    reflection) minimising sum |T(src_i) - dst_i|^2."""
    src = np.asarray(src, dtype=np.float64)
    dst = np.asarray(dst, dtype=np.float64)
    n, dim = src.shape
    sm, dm = src.mean(0), dst.mean(0)
    sd, dd = src - sm, dst - dm
    A = dd.T @ sd / n
    d = np.ones(dim)
    if np.linalg.det(A) < 0:
        d[dim - 1] = -1
    T = np.eye(dim + 1)
    U, S, V = np.linalg.svd(A)
    rank = np.linalg.matrix_rank(A)
    if rank == 0:
        return np.nan * T
    if rank == dim - 1:
        if np.linalg.det(U) * np.linalg.det(V) > 0:
            T[:dim, :dim] = U @ V
        else:
            s = d[dim - 1]
            d[dim - 1] = -1
            T[:dim, :dim] = U @ np.diag(d) @ V
            d[dim - 1] = s
    else:
        T[:dim, :dim] = U @ np.diag(d) @ V
    scale = 1.0 / sd.var(axis=0).sum() * (S @ d)
    T[:dim, dim] = dm - scale * (T[:dim, :dim] @ sm.T)
    T[:dim, :dim] *= scale
    return T


def align_matrix(landmarks):
    """(5,2) detected landmarks -> flattened inverse 2x3 (6,) float64 mapping
    crop pixel centres to source-image coordinates (wrapper.py:50-61)."""
    dst = np.asarray(landmarks).astype(np.float32)
    T = umeyama(dst, FACE_TEMPLATE_112)
    return np.linalg.inv(T)[0:2, :].flatten()


def pil_affine_bilinear(image, a, out_h=112, out_w=112):
    """image (H,W,3) uint8, a = 6 doubles -> (out_h,out_w,3) uint8.

    Pillow Geometry.c: affine_transform (pixel centre +0.5, all double) followed by
    bilinear_filter32RGB: reject if the source point is outside [0,W)x[0,H); shift by
    -0.5, floor, clip the four taps to the image, lerp in double, TRUNCATE to uint8.
    Rejected pixels keep the fill colour 0."""
    img = np.asarray(image)
    H, W = img.shape[:2]
    a = np.asarray(a, dtype=np.float64)
    ys, xs = np.mgrid[0:out_h, 0:out_w]
    xin = xs + 0.5
    yin = ys + 0.5
    sx = a[0] * xin + a[1] * yin + a[2]
    sy = a[3] * xin + a[4] * yin + a[5]
    inside = ~((sx < 0.0) | (sx >= W) | (sy < 0.0) | (sy >= H))
    sx = sx - 0.5
    sy = sy - 0.5
    x = np.floor(sx).astype(np.int64)
    y = np.floor(sy).astype(np.int64)
    dx = (sx - x)[..., None]
    dy = (sy - y)[..., None]
    x0 = np.clip(x, 0, W - 1)
    x1 = np.clip(x + 1, 0, W - 1)
    yc = np.clip(y, 0, H - 1)
    f = img.astype(np.float64)
    p00, p01 = f[yc, x0], f[yc, x1]
    v1 = p00 + (p01 - p00) * dx
    row2 = (y + 1 >= 0) & (y + 1 < H)
    y1 = np.clip(y + 1, 0, H - 1)
    p10, p11 = f[y1, x0], f[y1, x1]
    v2 = np.where(row2[..., None], p10 + (p11 - p10) * dx, v1)
    v = v1 + (v2 - v1) * dy
    out = v.astype(np.uint8)                       # C cast (UINT8)v1: truncation
    out[~inside] = 0
    return out


def preprocess_face(image, landmarks):
    """-> (3,112,112) uint8, BGR, CHW (wrapper.py:22-72)."""
    warped = pil_affine_bilinear(image, align_matrix(landmarks))
    return np.ascontiguousarray(warped.transpose(2, 0, 1)[::-1])


# ---- Pillow Image.resize (separable convolution, Resample.c) -------------------
_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _resample_coeffs(in_size, out_size, support=2.0, filt=_bicubic):
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = support * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds, kk = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(k)
        k = [(v / ww if ww != 0.0 else v) for v in k]
        k += [0.0] * (ksize - len(k))
        ik = [int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
              for v in k]
        bounds.append((xmin, xmax))
        kk.append(ik)
    return bounds, kk


def _resample_axis_h(img, out_size):
    H, W, C = img.shape
    bounds, kk = _resample_coeffs(W, out_size)
    out = np.empty((H, out_size, C), np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        k = np.asarray(kk[xx][:xmax], dtype=np.int64)
        ss = (1 << (_PRECISION_BITS - 1)) + (src[:, xmin:xmin + xmax, :] * k[None, :, None]).sum(1)
        out[:, xx, :] = np.clip(ss >> _PRECISION_BITS, 0, 255)
    return out


def pil_resize_bicubic(image, size):
    """image (H,W,3) uint8, size=(out_w,out_h) -> uint8; horizontal pass first,
    then vertical (Pillow ImagingResample)."""
    img = np.asarray(image)
    ow, oh = size
    H, W = img.shape[:2]
    tmp = _resample_axis_h(img, ow) if ow != W else img
    if oh != H:
        tmp = _resample_axis_h(tmp.transpose(1, 0, 2), oh).transpose(1, 0, 2)
    return np.ascontiguousarray(tmp)


def preprocess_face_no_landmarks(image, image_side=112):
    """Aspect-preserving resize + centre pad, BGR CHW uint8 (wrapper.py:75-99)."""
    h, w = image.shape[:2]
    scale = image_side / max(w, h)
    nw, nh = int(w * scale), int(h * scale)
    face = pil_resize_bicubic(image, (nw, nh))
    x_min = int((image_side - nw) / 2)
    y_min = int((image_side - nh) / 2)
    out = np.zeros((3, image_side, image_side), np.uint8)
    out[:, y_min:y_min + nh, x_min:x_min + nw] = face.transpose(2, 0, 1)[::-1]
    return out


def l2_normalize(x):
    """Row-wise x / ||x||_2, float32 in -> float32 out; zero rows stay zero
    (sklearn.preprocessing.normalize semantics)."""
    x = np.asarray(x, dtype=np.float32)
    n = np.sqrt(np.einsum('ij,ij->i', x, x)).astype(np.float32)
    n[n == 0.0] = 1.0
    return x / n[:, None]


def cosine_distance(u, v):
    """1 - u.v/(|u||v|) for every pair: u (A,D), v (B,D) -> (A,B) float64."""
    u = np.asarray(u, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    uu = np.sqrt((u * u).sum(1))[:, None]
    vv = np.sqrt((v * v).sum(1))[None, :]
    return 1.0 - (u @ v.T) / (uu * vv)

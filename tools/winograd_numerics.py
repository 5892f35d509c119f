"""Numerical cost of Winograd F(2x2, 3x3) on split-half operands (CPU, numpy; design aid for DESIGN.md section 7).

One 3x3 stride-1 layer (128 -> 128 channels, 28 x 28, wild per-channel gains) against a float64 evaluation:
float32 direct, the f16x3 scheme as the kernels do it (hi*hi + hi*lo + lo*hi, per-channel / per-row powers of two, f32
accumulate), and Winograd with float32 input / output transforms around the same split-half products.

    python tools/winograd_numerics.py
"""
import numpy as np
rng = np.random.default_rng(0)

def split_f16(x):
    """x (float32/64) -> hi, lo float16 with per-row/col scaling assumed done outside; returns float64 of hi+lo"""
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float64)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)

def mm_x3(A, B):
    """A [M,K], B [K,N] float64 inputs already scaled into half range; emulate hi*hi + hi*lo + lo*hi with f32 accumulate"""
    ah, al = split_f16(A); bh, bl = split_f16(B)
    # products exact in f32? (11b x 11b = 22b fits f32 24b) ; accumulate in f32: emulate by float32 matmul of exact products
    return (ah.astype(np.float32) @ bh.astype(np.float32)) + (ah.astype(np.float32) @ bl.astype(np.float32)) + (al.astype(np.float32) @ bh.astype(np.float32))

def scale_rows(X):  # power-of-two per-row scale to put max in [2^9, 2^10)
    m = np.abs(X).max(axis=1, keepdims=True); m[m == 0] = 1
    e = 9 - np.floor(np.log2(m))
    return np.ldexp(1.0, e.astype(int)), X * np.ldexp(1.0, e.astype(int))

cin, cout, H, W = 128, 128, 28, 28
x = np.maximum(rng.normal(0.3, 1.0, (cin, H + 2, W + 2)), 0) * np.exp(rng.uniform(-3, 3, (cin, 1, 1)))   # padded input, wild channel gains
x[:, 0, :] = x[:, -1, :] = 0; x[:, :, 0] = x[:, :, -1] = 0
w = rng.normal(0, 1, (cout, cin, 3, 3)) / np.sqrt(cin * 9) * np.exp(rng.uniform(-2, 2, (cout, 1, 1, 1))) / np.exp(rng.uniform(-3, 3, (1, cin, 1, 1)))
# float64 reference
ref = np.zeros((cout, H, W))
for ky in range(3):
    for kx in range(3):
        ref += np.einsum('oc,chw->ohw', w[:, :, ky, kx], x[:, ky:ky + H, kx:kx + W])
scale = np.abs(ref).max()
# (1) float32 direct (im2col matmul in f32)
col = np.stack([x[:, ky:ky + H, kx:kx + W] for ky in range(3) for kx in range(3)], 1).reshape(cin * 9, H * W)
wm = w.transpose(0, 1, 2, 3).reshape(cout, cin * 9)
d32 = (wm.astype(np.float32) @ col.astype(np.float32)).reshape(cout, H, W)
print('f32 direct       max err / max|ref| = %.3g' % (np.abs(d32 - ref).max() / scale))
# (2) f16x3 direct: per-cout weight scale, per-channel activation scale folded: emulate with row scaling of wm and per-channel of col
sw, wms = scale_rows(wm)
# activation per-channel exponent (stored x*2^a), folded 2^-a into weights columns: emulate by scaling col rows per channel to ~2^9 max
ch_max = np.abs(col.reshape(cin, 9, -1)).max(axis=(1, 2)); e = 9 - np.floor(np.log2(ch_max)); sa = np.ldexp(1.0, e.astype(int))
cols = col * np.repeat(sa, 9)[:, None]
wm2 = wm / np.repeat(sa, 9)[None, :]
sw, wms = scale_rows(wm2)
d3 = (mm_x3(wms, cols) / sw).reshape(cout, H, W)
print('f16x3 direct     max err / max|ref| = %.3g' % (np.abs(d3 - ref).max() / scale))
# (3) Winograd F(2x2,3x3)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
U = np.einsum('ij,ocjk,lk->ocil', G, w, G)                       # [cout, cin, 4, 4] in f64 (pack time)
th, tw = H // 2, W // 2
tiles = np.zeros((cin, th, tw, 4, 4))
for i in range(th):
    for j in range(tw):
        tiles[:, i, j] = x[:, 2 * i:2 * i + 4, 2 * j:2 * j + 4]
def run_wino(f32_transforms, x3):
    t = tiles.astype(np.float32) if f32_transforms else tiles
    Btt = Bt.astype(t.dtype)
    V = np.einsum('ij,cabjk,lk->cabil', Btt, t, Btt)            # input transform (device: f32 VALU)
    M = np.zeros((cout, th, tw, 4, 4))
    for a in range(4):
        for b in range(4):
            Uab = U[:, :, a, b]; Vab = V[:, :, :, a, b].reshape(cin, th * tw).astype(np.float64)
            if x3:
                cm = np.abs(Vab).max(axis=1); cm[cm == 0] = 1; ea = 9 - np.floor(np.log2(cm)); s_a = np.ldexp(1.0, ea.astype(int))
                Vs = Vab * s_a[:, None]; Us = Uab / s_a[None, :]
                s_w, Uss = scale_rows(Us)
                Mab = mm_x3(Uss, Vs) / s_w
            else:
                Mab = (Uab.astype(np.float32) @ Vab.astype(np.float32)).astype(np.float64)
            M[:, :, :, a, b] = Mab.reshape(cout, th, tw)
    Mo = M.astype(np.float32) if f32_transforms else M
    Att = At.astype(Mo.dtype)
    Y = np.einsum('ij,oabjk,lk->oabil', Att, Mo, Att)            # [cout, th, tw, 2, 2]
    return Y.transpose(0, 1, 3, 2, 4).reshape(cout, H, W).astype(np.float64)
for name, f32t, x3 in (('winograd f64 transforms, f32 GEMM', False, False), ('winograd f32 transforms, f32 GEMM', True, False),
                       ('winograd f32 transforms, f16x3 GEMM', True, True)):
    y = run_wino(f32t, x3)
    print('%-38s max err / max|ref| = %.3g' % (name, np.abs(y - ref).max() / scale))

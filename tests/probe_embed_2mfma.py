"""CPU emulation (no GPU; uses the oracle, so it lives under tests/): the ArcFace embedding when ONE side of every conv / FC
contraction carries 11 significant bits (a single half float with an ideal per-channel exponent) and the other side the
22 bits of the split-half pair -- the TWO-MFMA embedder the round-4 VERDICT (item 6) asks to measure:
    x11 : activations rounded to 11 bits in front of every contraction, weights exact     (w_hi + w_lo) . x_hi
    w11 : weights rounded to 11 bits, activations exact                                   w_hi . (x_hi + x_lo)
    both: the single-MFMA `f16` mode (for comparison with its measured 3.3e-4 / 1.8e-3)
The shortcut trunk stays float32 in all three (what a kernel that reads only the hi half of a split-half tensor would see).
    python tests/probe_embed_2mfma.py            ->  profiles/r05_embed_2mfma.txt"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nets, arcface_pre          # noqa: E402
from terran_amd import weights                # noqa: E402
from tests import wild_weights                # noqa: E402

torch.set_num_threads(32)
orig_conv, orig_lin = F.conv2d, F.linear


def r11(t):
    """round to 11 significant bits, no range limits (the device stores every channel with its own exponent)"""
    m, e = torch.frexp(t)
    return torch.ldexp(torch.round(m * 2048.0) / 2048.0, e)


def run(sd, x, rx, rw):
    def conv(x_, w, *a, **k):
        return orig_conv(rx(x_), rw(w), *a, **k)

    def lin(x_, w, *a, **k):
        return orig_lin(rx(x_), rw(w), *a, **k)
    F.conv2d, F.linear = conv, lin
    try:
        return arcface_pre.l2_normalize(nets.arcface_forward(sd, x).numpy())
    finally:
        F.conv2d, F.linear = orig_conv, orig_lin


def crops(seed, n):
    rng = np.random.default_rng(seed)
    c = rng.integers(0, 256, (n, 3, 112, 112)).astype(np.float32)
    c[n // 2:] = wild_weights._calib_frames(77 + seed, n - n // 2, 112, 112)[..., ::-1].transpose(0, 3, 1, 2)
    return torch.from_numpy(c)


ident = lambda t: t                           # noqa: E731
for name, sd in (('benign', weights.make_arcface_state()), ('wild', wild_weights.MAKERS['arcface']())):
    x = crops(5, int(os.environ.get("N_CROPS", "16")))
    ref = run(sd, x, ident, ident)
    for mode, rx, rw in (('x11 (2 MFMAs: w_hi+w_lo times x_hi)', r11, ident), ('w11 (2 MFMAs: w_hi times x_hi+x_lo)', ident, r11),
                         ('both 11 bits (1 MFMA, the f16 mode)', r11, r11)):
        got = run(sd, x, rx, rw)
        d = np.abs(got - ref)
        print('%-7s %-40s max |d component| %.2e  rms %.2e  max cosine distance %.2e' % (name, mode, d.max(), np.sqrt((d ** 2).mean()), (1 - (got * ref).sum(1)).max()), flush=True)

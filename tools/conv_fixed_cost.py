"""Fixed cost per conv launch: 1x1 / 3x3 / 5x5 / 7x7 convs 128 -> 128 on the 230 tiles of the 1080p pose layers
(4 / 36 / 100 / 196 K slabs); a linear fit of duration over slabs gives the per-slab time and the prologue + epilogue
intercept (HIP events around each launch, kernels back to back on one stream)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import lib, pack, synth   # noqa: E402


def run(ctx, k, precision, c=128, n=32, h=23, w=40, reps=30, layers=4):
    rng = np.random.default_rng(0)
    P = pack.Program(pack.MODEL_OPENPOSE, precision)
    t0 = P.tensor(4, 1)
    P.input_tensor = t0
    t1 = P.tensor(c, k // 2)
    P.conv(t0, t1, rng.normal(0, 0.3, (c, 3, 3, 3)).astype(np.float32), np.zeros(c, np.float32), act=pack.ACT_RELU)
    cur = t1
    for _ in range(layers):
        t = P.tensor(c, k // 2)
        P.conv(cur, t, rng.normal(0, 0.05, (c, c, k, k)).astype(np.float32), np.zeros(c, np.float32), act=pack.ACT_RELU)
        cur = t
    P.outputs = [cur]
    m = lib.Model(ctx, P)
    fr = ctx.upload(synth.frames(1, n, h, w))
    m.forward_frames(fr)
    ctx.sync()
    ctx.profile_reset()
    ctx.profile(True)
    for _ in range(reps):
        m.forward_frames(fr)
    ms, launches, _ = ctx.profile_read(0)
    ctx.profile(False)
    m.free()
    fr.free()
    return ms / reps


if __name__ == '__main__':
    ctx = lib.Context(0)
    prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
    stem = run(ctx, 1, prec, layers=0)
    xs, ys = [], []
    for k in (1, 3, 5, 7):
        t = (run(ctx, k, prec) - stem) / 4 * 1e3
        xs.append(k * k * 4)
        ys.append(t)
        print('%dx%d 128->128 on 230 tiles: %3d slabs  %6.1f us per launch' % (k, k, k * k * 4, t))
    a, b = np.polyfit(xs, ys, 1)
    print('fit: %.3f us per slab + %.1f us fixed per launch' % (a, b))

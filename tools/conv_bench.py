"""Single-layer conv micro-benchmark on the HIP path (tuning aid; run on the GPU box).

    python tools/conv_bench.py            # the two shapes that dominate the 1080p workload
Prints TFLOP/s (algorithmic) per shape and precision.  TA_CONV_PREFER=<TA_CONV_* code> prefers one kernel variant
wherever it is eligible; TA_CONV_PROBE=1|2 are timing ablations of the split-role kernel (no pixel-row DMA / no DMA at
all once the LDS ring is full: WRONG results, upper bounds only)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import lib, pack, synth   # noqa: E402

SHAPES = [
    # name, n, h, w, cin, cout, k, groups
    ('pose 7x7 128->128 @32x23x40', 32, 23, 40, 128, 128, 7, 1),
    ('pose 7x7 2x(128->128) grouped @32x23x40', 32, 23, 40, 256, 256, 7, 2),
    ('pose 7x7 2x(128->128) grouped @16x46x82', 16, 46, 82, 256, 256, 7, 2),
    ('arc  3x3 256->256 @64x14x14', 64, 14, 14, 256, 256, 3, 1),
    ('arc  3x3 256->256 @256x14x14', 256, 14, 14, 256, 256, 3, 1),
    ('arc  3x3 128->128 @64x28x28', 64, 28, 28, 128, 128, 3, 1),
    ('arc  3x3 64->64 @64x56x56', 64, 56, 56, 64, 64, 3, 1),
    # the transform-domain products of Winograd F(2x2, 3x3) as 1x1 convs of the same dimensions (16 positions x tiles = 'pixels', K = cin):
    # what the MFMA part of such a layer would cost on the kernels that exist, before any transform
    ('wino-domain of arc 256->256 @64x14x14: 1x1 @64x28x28', 64, 28, 28, 256, 256, 1, 1),
    ('wino-domain of pose 256->256 @32x46x81: 1x1 @32x122x124', 32, 122, 124, 256, 256, 1, 1),
    ('vgg  3x3 64->64 @32x184x327', 32, 184, 327, 64, 64, 3, 1),
    ('pose 3x3 256->256 @32x46x81', 32, 46, 81, 256, 256, 3, 1),
    ('pose 3x3 128->128 @32x92x163', 32, 92, 163, 128, 128, 3, 1),
]


def bench(ctx, name, n, h, w, cin, cout, k, groups, precision, reps=20):
    rng = np.random.default_rng(0)
    P = pack.Program(pack.MODEL_OPENPOSE, precision)
    t0 = P.tensor(4, 1)
    P.input_tensor = t0
    t1 = P.tensor(cin, k // 2)
    P.conv(t0, t1, rng.normal(0, 0.3, (cin, 3, 3, 3)).astype(np.float32), np.zeros(cin, np.float32), act=pack.ACT_RELU)
    cur = t1
    layers = 4
    for i in range(layers):
        t = P.tensor(cout, k // 2)
        wscale = 0.0 if os.environ.get('TA_BENCH_ZERO') else 1.0    # zero data: shows the DVFS/power share of a result
        P.conv(cur, t, (wscale * rng.normal(0, 1 / np.sqrt(cin * k * k), (cout, cin // groups, k, k))).astype(np.float32),
               np.zeros(cout, np.float32), act=pack.ACT_RELU, groups=groups)
        cur = t
        cin = cout
    P.outputs = [cur]
    m = lib.Model(ctx, P)
    fr = ctx.upload(synth.frames(1, n, h, w))
    m.forward_frames(fr)
    ctx.sync()
    ctx.profile_reset()
    ctx.profile(True)
    for _ in range(reps):
        m.forward_frames(fr)
    ms, launches, work = ctx.profile_read(0)
    ctx.profile(False)
    # subtract the small first conv: measure it alone
    flops_layer = 2.0 * n * h * w * cout * (cout // groups) * k * k
    per_layer_ms = ms / reps / (layers + 1) * (layers + 1)      # total per forward
    tf = (layers * flops_layer) / (per_layer_ms * 1e-3) / 1e12
    print('%-58s %-7s %7.3f ms/forward(%d layers)  ~%6.1f TF  %s' % (name, precision, per_layer_ms, layers, tf,
                                                                    ctx.conv_counts(reset=True)))
    m.free()
    fr.free()


if __name__ == '__main__':
    ctx = lib.Context(0)
    precs = sys.argv[1:] or ['f32', 'f16x3', 'bf16x3', 'bf16']
    for s in SHAPES:
        for p in precs:
            bench(ctx, *s, p)

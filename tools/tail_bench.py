"""Tools: the OpenPose stage tails (1x1 grouped 2 x (128 -> 128) + ReLU, then 1x1 256 -> 60) under each conv kernel variant.
    python tools/tail_bench.py"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import lib, pack, synth   # noqa: E402


def bench(ctx, variant, n=32, h=23, w=40, reps=30):
    rng = np.random.default_rng(0)
    P = pack.Program(pack.MODEL_OPENPOSE, 'f16x3')
    t0 = P.tensor(4, 1)
    P.input_tensor = t0
    t1 = P.tensor(256, 0)
    P.conv(t0, t1, rng.normal(0, 0.3, (256, 3, 3, 3)).astype(np.float32), np.zeros(256, np.float32), act=pack.ACT_RELU)
    t2 = P.tensor(256, 0)
    P.conv(t1, t2, rng.normal(0, 0.1, (256, 128, 1, 1)).astype(np.float32), np.zeros(256, np.float32), act=pack.ACT_RELU, groups=2)
    X = P.tensor(192, 3, f32=False)
    outs = []
    for i in range(4):
        P.conv(t2, X, rng.normal(0, 0.1, (60, 256, 1, 1)).astype(np.float32), np.zeros(60, np.float32), out_ch_off=128, cout_p=60,
               variant=lib.CONV_VARIANTS[variant])
    t3 = P.tensor(64, 0, f32=True)
    P.conv(X, t3, rng.normal(0, 0.1, (64, 192, 1, 1)).astype(np.float32), np.zeros(64, np.float32))
    P.outputs = [t3]
    m = lib.Model(ctx, P)
    fr = ctx.upload(synth.frames(1, n, h, w))
    os.environ['TA_PROFILE_OPS'] = '1'
    m.forward_frames(fr)
    m.forward_frames(fr)
    ctx.sync()
    m.free()


if __name__ == '__main__':
    os.environ['TA_PROFILE_OPS'] = '1'
    ctx = lib.Context(0)
    for v in ('auto', 'split_1x4', 'pipe64', 'generic'):
        print('=== variant', v, file=sys.stderr, flush=True)
        bench(ctx, v)

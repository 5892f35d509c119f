"""Cycle stamps inside workgroup 0 of the detector's fused depthwise + 1x1 block (debug build:
TA_EXTRA_FLAGS=-DTA_CONV_TRACE python -m terran_amd.build).  Where do the ~13 us of a 128-pixel tile go?

    python tools/dwpw_trace.py [C cout n h w]        default 32 32 32 160 160 (RetinaFace block 2 at C2)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import lib, pack, synth   # noqa: E402

a = [int(v) for v in sys.argv[1:]]
ch, cout, n, h, w = a if len(a) == 5 else (32, 32, 32, 160, 160)
ctx = lib.Context(0)
rng = np.random.default_rng(0)
P = pack.Program(pack.MODEL_OPENPOSE, 'f16x3')
t0 = P.tensor(4, 1)
P.input_tensor = t0
P.input_stats = (np.array([-0.05] * 3 + [0.0]), np.array([0.08] * 3 + [0.0]))
t1 = P.tensor(ch, 1)
P.conv(t0, t1, rng.normal(0, 0.3, (ch, 3, 3, 3)).astype(np.float32), np.zeros(ch, np.float32), act=pack.ACT_RELU, precision='f32')
t2 = P.tensor(cout, 1, name='out', f32=True)
P.dwpw(t1, t2, rng.normal(0, 0.3, (ch, 1, 3, 3)).astype(np.float32), rng.normal(0, 0.1, ch).astype(np.float32),
       rng.normal(0, 0.1, (cout, ch, 1, 1)).astype(np.float32), rng.normal(0, 0.1, cout).astype(np.float32), precision='f32')
P.outputs = [t2]
m = lib.Model(ctx, P)
fr = ctx.upload(synth.frames(1, n, h, w))
for _ in range(3):
    m.forward_frames(fr)
ctx.sync()
buf = (C.c_longlong * 32)()
ctx.lib.ta_debug_trace_read.argtypes = [C.c_void_p, C.c_int]
assert ctx.lib.ta_debug_trace_read(buf, 32) == 0
t = list(buf)
names = {16: 'entry', 17: 'pixel addresses ready', 18: 'slab 0 produced (taps loaded, rows written)', 19: 'past the first barrier (weights landed)',
         20: 'MFMAs issued', 21: 'finish: ring free', 22: 'tile parked', 23: 'drained (stores issued)'}
for i in range(16, 24):
    print('%-48s +%7d cycles' % (names[i], t[i] - t[16]))
print('dw %d -> pw %d @ %d x %d x %d: %s' % (ch, cout, n, h, w, {k: v for k, v in ctx.conv_counts().items() if v}))

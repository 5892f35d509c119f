"""Cycle stamps inside one workgroup of the split-role conv kernel (debug build: TA_EXTRA_FLAGS=-DTA_CONV_TRACE
python -m terran_amd.build --force).  Prints where the fixed ~10 us of a launch go."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import lib, pack, synth   # noqa: E402

ctx = lib.Context(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cin, cout = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (128, 128)
n, h, w = (int(v) for v in sys.argv[4:7]) if len(sys.argv) > 6 else (32, 23, 40)
prec = sys.argv[7] if len(sys.argv) > 7 else 'f16x3'         # TA_CONV_PROBE=$((block << 8)) picks the stamped workgroup
rng = np.random.default_rng(0)
P = pack.Program(pack.MODEL_OPENPOSE, prec)
t0 = P.tensor(4, 1)
P.input_tensor = t0
t1 = P.tensor(cin, k // 2)
P.conv(t0, t1, rng.normal(0, 0.3, (cin, 3, 3, 3)).astype(np.float32), np.zeros(cin, np.float32), act=pack.ACT_RELU)
t2 = P.tensor(cout, 1)
P.conv(t1, t2, rng.normal(0, 0.05, (cout, cin, k, k)).astype(np.float32), np.zeros(cout, np.float32), act=pack.ACT_RELU)
t3 = P.tensor(64, 0, f32=True)                                      # keeps t2 in the split format (lean epilogue)
P.conv(t2, t3, rng.normal(0, 0.05, (64, cout, 3, 3)).astype(np.float32), np.zeros(64, np.float32))
P.outputs = [t3]
m = lib.Model(ctx, P)
fr = ctx.upload(synth.frames(1, n, h, w))
for _ in range(3):
    m.forward_frames(fr)
ctx.sync()
buf = (C.c_longlong * 16)()
ctx.lib.ta_debug_trace_read.argtypes = [C.c_void_p, C.c_int]
assert ctx.lib.ta_debug_trace_read(buf, 16) == 0
t = list(buf)
base = min(t[0], t[8])
names = {0: 'consumer entry', 1: 'consumer set up (waits B_0)', 2: 'slab 0 landed', 3: 'main loop done', 4: 'epilogue issued',
         5: 'past barrier E0', 6: 'accumulators parked', 7: 'past barrier E1',
         8: 'producer entry', 9: 'producer addresses ready', 10: 'first slabs issued'}
for i in (8, 9, 10, 0, 1, 2, 3, 5, 6, 7, 4):
    print('%-32s +%7d cycles' % (names[i], t[i] - base))
print('layer k%d %d->%d @%dx%dx%d %s, stamped workgroup %s: %s' % (k, cin, cout, n, h, w, prec, int(os.environ.get('TA_CONV_PROBE', '0')) >> 8, ctx.conv_counts()))

"""Per-op HIP-event table of the detector alone (TA_PROFILE_OPS=1; lanes are serialised while profiling).
    python tools/detector_profile.py [n h w] [precision]      default 32 416 739 f16x3 (the 1080p working size); C2 = 32 640 640"""
import os
import sys

os.environ['TA_PROFILE_OPS'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import lib, pack, synth, weights                         # noqa: E402

a = sys.argv[1:]
shape = tuple(int(x) for x in a[:3]) if len(a) >= 3 else (32, 416, 739)
prec = a[3] if len(a) >= 4 else (a[0] if len(a) == 1 else 'f16x3')
ctx = lib.Context(0)
m = lib.Model(ctx, pack.pack_retinaface(weights.make_retinaface_state(), prec))
fr = ctx.upload(synth.frames(1, *shape))
for _ in range(3):                                                       # the last table is the warm one
    m.forward_frames(fr)

import os, sys
os.environ['TA_PROFILE_OPS'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import lib, pack, synth, weights
ctx = lib.Context(0)
m = lib.Model(ctx, pack.pack_retinaface(weights.make_retinaface_state(), 'f16x3'))
fr = ctx.upload(synth.frames(1, 32, 416, 739))
for _ in range(3):
    m.forward_frames(fr)

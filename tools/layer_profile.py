"""Per-op time of the three networks at the 1080p working sizes of bench.py (one forward each, TA_PROFILE_OPS=1 makes the
library print a HIP-event table per forward on stderr).   python tools/layer_profile.py [precision]"""
import os
import sys

os.environ['TA_PROFILE_OPS'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                                      # noqa: E402
from terran_amd import lib, pack, synth, weights                         # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
ctx = lib.Context(0)
for name, packer, sd, shape in (('openpose', pack.pack_openpose, weights.make_openpose_decoder_state(), (32, 184, 327)),
                                ('retinaface', pack.pack_retinaface, weights.make_retinaface_state(), (32, 416, 739))):
    m = lib.Model(ctx, packer(sd, prec))
    fr = ctx.upload(synth.frames(1, *shape))
    for _ in range(2):                                                   # the second table is the warm one
        print('==== %s %s' % (name, prec), file=sys.stderr, flush=True)
        m.forward_frames(fr)
    m.free()
m = lib.Model(ctx, pack.pack_arcface(weights.make_arcface_state(), prec))
crops = np.random.default_rng(2).integers(0, 256, (64, 3, 112, 112), dtype=np.uint8)
for _ in range(2):
    print('==== arcface %s' % prec, file=sys.stderr, flush=True)
    m.forward_crops(crops)

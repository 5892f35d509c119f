"""Would hipGraphs buy anything?  Replays each network's op program (the launches of one forward at the 1080p working size
and at the BASELINE sizes) as plain stream launches and as a hipGraph captured from those very launches
(ta_model_graph_probe): GPU time per replay and host enqueue time per replay, both ways.

    python tools/graph_probe.py [f16x3]          # on the GPU box
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import lib, pack, runtime, synth, weights   # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
    ctx = runtime.get_context(0)
    cases = [('retinaface', 'RetinaFace 32 x 416 x 739 (1080p step)', lambda m: m.forward_frames(ctx.upload(synth.frames(1, 32, 416, 739)))),
             ('retinaface', 'RetinaFace 32 x 640 x 640 (C2)', lambda m: m.forward_frames(ctx.upload(synth.frames(1, 32, 640, 640)))),
             ('retinaface', 'RetinaFace 1 x 416 x 739 (one frame)', lambda m: m.forward_frames(ctx.upload(synth.frames(1, 1, 416, 739)))),
             ('openpose', 'OpenPose 32 x 184 x 327 (1080p step)', lambda m: m.forward_frames(ctx.upload(synth.frames(3, 32, 184, 327)))),
             ('openpose', 'OpenPose 1 x 184 x 327 (one frame)', lambda m: m.forward_frames(ctx.upload(synth.frames(3, 1, 184, 327)))),
             ('arcface', 'ArcFace 64 crops', lambda m: m.forward_crops(np.random.default_rng(2).integers(0, 256, (64, 3, 112, 112), dtype=np.uint8))),
             ('arcface', 'ArcFace 2 crops', lambda m: m.forward_crops(np.random.default_rng(2).integers(0, 256, (2, 3, 112, 112), dtype=np.uint8)))]
    progs = {}
    print('%-40s %5s | GPU ms / replay: streams  graph | host enqueue ms / replay: streams  graph | capture+instantiate ms' % (prec, 'ops'))
    for kind, name, fwd in cases:
        if kind not in progs:
            progs[kind] = getattr(pack, 'pack_' + kind)(getattr(weights, 'make_%s_state' % kind)(), prec)
        m = lib.Model(ctx, progs[kind])
        fwd(m)
        ctx.sync()
        g = m.graph_probe(20)
        print('%-40s %5d | %24.3f %6.3f | %32.3f %6.3f | %8.1f' % (name, len(progs[kind].ops), g[0], g[1], g[2], g[3], g[4]))
        m.free()


if __name__ == '__main__':
    main()

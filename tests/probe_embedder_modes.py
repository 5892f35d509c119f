"""GPU measurement (uses the oracle as the referee: lives under tests/): the embedder's arithmetic modes side by side --
unit-embedding error against the oracle on the benign and on the wild-statistics ArcFace weights (64 crops, half noise half
smooth), and the network's time at 64 / 256 / 320 crops (HIP events around every conv launch).
    python tests/probe_embedder_modes.py [modes...]     ->  profiles/r05_embedder_modes.txt"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arcface_pre, nets                     # noqa: E402
from terran_amd import ArcFace, runtime, weights         # noqa: E402
from tests import wild_weights                           # noqa: E402

torch.set_num_threads(64)
modes = sys.argv[1:] or ['f32', 'f16x3', 'f16x2', 'f16']
ctx = runtime.get_context(0)
rng = np.random.default_rng(5)
crops = rng.integers(0, 256, (64, 3, 112, 112), dtype=np.uint8)
crops[32:] = wild_weights._calib_frames(77, 32, 112, 112)[..., ::-1].transpose(0, 3, 1, 2)
for stats in ('benign', 'wild'):
    sd = weights.make_arcface_state() if stats == 'benign' else wild_weights.MAKERS['arcface']()
    ref = arcface_pre.l2_normalize(nets.arcface_forward(sd, torch.from_numpy(crops.astype(np.float32))).numpy())
    for mode in modes:
        a = ArcFace(device=0, state=sd, precision=mode)
        e = a.embed_crops(crops)
        d = np.abs(e - ref)
        line = '%-6s weights, %-5s: unit embeddings vs oracle max %.3g (noise crops %.3g, smooth %.3g) rms %.3g, max cosine distance %.3g, range fallbacks %d' % (
            stats, mode, d.max(), d[:32].max(), d[32:].max(), np.sqrt((d * d).mean()), 1.0 - (e * ref).sum(1).min(), a.fallbacks)
        if stats == 'benign':
            for n in (64, 256, 320):
                c = rng.integers(0, 256, (n, 3, 112, 112), dtype=np.uint8)
                for _ in range(2):
                    a.embed_crops(c)
                ctx.sync()
                ctx.profile_reset()
                ctx.profile(True)
                for _ in range(3):
                    a.embed_crops(c)
                ms, launches, work = ctx.profile_read(0)
                ctx.profile(False)
                line += ' | %d crops: %.3f ms, %.0f TF' % (n, ms / 3, work / (ms * 1e-3) / 1e12)
        print(line, flush=True)
        a.model.free()
    runtime.clear_pack_memo()

"""CPU emulation (no GPU): how far does the ArcFace embedding move when every conv / FC operand is rounded to ONE half float
(a single-MFMA embedder), to one bfloat16, or to the hi + lo pair the f16x3 mode carries?  The oracle's network with its
operands rounded in front of every contraction, float32 accumulation.  Measurement script (uses the oracle: lives under tests/).
    python tests/probe_embed_precision.py [seeds]     ->  DESIGN.md section 7   (`seeds`: four other weight seeds, noise and
                                                       smooth crops, single half only)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from oracle import nets, arcface_pre
from terran_amd import weights
torch.set_num_threads(32)
sd = weights.make_arcface_state()
rng = np.random.default_rng(3)
x = torch.from_numpy(rng.integers(0, 256, (16, 3, 112, 112)).astype(np.float32))
ref = arcface_pre.l2_normalize(nets.arcface_forward(sd, x).numpy())
orig_conv, orig_lin = F.conv2d, F.linear
def mk(rnd):
    def conv(x, w, *a, **k): return orig_conv(rnd(x), rnd(w), *a, **k)
    def lin(x, w, *a, **k): return orig_lin(rnd(x), rnd(w), *a, **k)
    return conv, lin
def split16(t):            # hi + lo of two halves: what f16x3 carries (22 bits)
    hi = t.half().float(); return hi + (t - hi).half().float()
modes = {'f16 single (hi only)': lambda t: t.half().float(),
         'bf16 single': lambda t: t.bfloat16().float(),
         'f16x3 operands (hi+lo, ll dropped ~)': split16}
for name, rnd in modes.items():
    F.conv2d, F.linear = mk(rnd)
    try:
        got = arcface_pre.l2_normalize(nets.arcface_forward(sd, x).numpy())
    finally:
        F.conv2d, F.linear = orig_conv, orig_lin
    d = np.abs(got - ref)
    cos = 1 - (got * ref).sum(1)
    print('%-40s max |d component| %.2e  mean %.2e  max cosine distance to f32 %.2e' % (name, d.max(), d.mean(), cos.max()))

if 'seeds' in sys.argv[1:]:
    r16 = modes['f16 single (hi only)']
    for seed in (11, 222, 3333, 44444):
        sd2 = weights.make_arcface_state(seed)
        rng2 = np.random.default_rng(seed)
        noise = rng2.integers(0, 256, (4, 3, 112, 112)).astype(np.float32)
        low = rng2.normal(128, 50, (4, 3, 7, 7)).astype(np.float32)
        smooth = np.clip(np.kron(low, np.ones((16, 16), np.float32)), 0, 255).round()
        x2 = torch.from_numpy(np.concatenate([noise, smooth]))
        ref2 = arcface_pre.l2_normalize(nets.arcface_forward(sd2, x2).numpy())
        F.conv2d, F.linear = mk(r16)
        try:
            got2 = arcface_pre.l2_normalize(nets.arcface_forward(sd2, x2).numpy())
        finally:
            F.conv2d, F.linear = orig_conv, orig_lin
        d2 = np.abs(got2 - ref2)
        print('seed %d: max |d| noise crops %.2e, smooth crops %.2e; max cosine distance %.2e'
              % (seed, d2[:4].max(), d2[4:].max(), (1 - (got2 * ref2).sum(1)).max()))

"""CPU emulation (no GPU): how far does the ArcFace embedding move when every conv / FC operand is rounded to ONE half float
(a single-MFMA embedder), to one bfloat16, or to the hi + lo pair the f16x3 mode carries?  The oracle's network with its
operands rounded in front of every contraction, float32 accumulation.  Measurement script (uses the oracle: lives under tests/).
    python tests/probe_embed_precision.py      ->  DESIGN.md section 7"""
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

"""RMS / max error of the pose network's output maps against a FLOAT64 evaluation, per arithmetic mode (GPU box).

    python tests/probe_map_error.py [wild|benign] [frames]
Where the decision flips of tests/test_gpu_decisions_vs_oracle.py come from: how far each device mode's PAF / heat maps
(and the torch-CPU float32 oracle's) are from the float64 maps, relative to the maps' RMS.  Pack-time knobs can be
varied through the environment for experiments (a probe under tests/: it calls the oracle, which only tests may): TA_CH_SPREAD, TA_ACT_TARGET_LOG2 (terran_amd/pack.py)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import lib, pack, runtime, synth, weights   # noqa: E402
from oracle import nets                                     # noqa: E402  (checker only)

if os.environ.get('TA_CH_SPREAD'):
    pack._CH_SPREAD = int(os.environ['TA_CH_SPREAD'])
if os.environ.get('TA_ACT_TARGET_LOG2'):
    pack._ACT_TARGET_LOG2 = int(os.environ['TA_ACT_TARGET_LOG2'])


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'wild'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    if kind == 'wild':
        from tests import wild_weights
        sd = wild_weights.MAKERS['openpose']()
    else:
        sd = weights.make_openpose_state()
    frames = synth.frames(4000, n, 184, 327)
    x64 = torch.from_numpy(np.transpose(frames, (0, 3, 1, 2)).astype(np.float64) / 255.0 - 0.5)
    sd64 = {k: (np.asarray(v).astype(np.float64) if np.asarray(v).dtype.kind == 'f' else np.asarray(v)) for k, v in sd.items()}
    p64, h64 = [t.numpy() for t in nets.openpose_forward(sd64, x64)]
    p32, h32 = [t.numpy() for t in nets.openpose_forward(sd, x64.float())]
    ref = np.concatenate([p64, h64], 1)
    rms = float(np.sqrt((ref * ref).mean()))

    def row(name, got):
        d = got.astype(np.float64) - ref
        print('%-22s rms err / rms(ref) %.3g   max err / rms(ref) %.3g' % (name, np.sqrt((d * d).mean()) / rms, np.abs(d).max() / rms))
    print('%s weights, %d frames 184 x 327, rms of the float64 maps %.4g; spread %d, target 2^%d' %
          (kind, n, rms, pack._CH_SPREAD, pack._ACT_TARGET_LOG2))
    row('torch-CPU float32', np.concatenate([p32, h32], 1))
    ctx = runtime.get_context(0)
    fr = ctx.upload(frames)
    for prec in ('f32', 'f16x3', 'bf16x3'):
        m = lib.Model(ctx, pack.pack_openpose(sd, prec))
        m.forward_frames(fr)
        row('device ' + prec, np.concatenate([m.read('pafs'), m.read('heatmaps')], 1))
        m.free()


if __name__ == '__main__':
    main()

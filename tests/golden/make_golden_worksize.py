"""Golden fixtures at the WORKING sizes of BASELINE configs[4] and on weights with trained-looking statistics, generated
FROM THE REFERENCE MODULES (container-only, like make_golden.py):  python tests/golden/make_golden_worksize.py

`oracle/nets.py` and `oracle/openpose_post.py` read the product's own layer tables (terran_amd/arch.py), so a table error
shared by oracle and product is only caught by a fixture the reference produced.  The tiny-input fixtures of
make_golden.py cover every layer; these add
  * one 184 x 327 OpenPose forward (the 1080p working size of the pose network): all 57 maps,
  * one 416 x 739 RetinaFace forward: the stride-32 / stride-16 head tensors in full, the stride-8 ones as per-channel
    sums + every 4th pixel (the full set is 0.8 MB),
and, for the wild-statistics weights of tests/wild_weights.py (BatchNorm gains / variances and channel gains spread over
orders of magnitude -- the regime of a trained checkpoint), the reference modules' outputs on small inputs: the oracle must
restate the reference there too (BatchNorm with running_var ~ 1e-7, negative gammas, PReLU slopes anywhere in [0, 1]).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
os.environ.setdefault('TERRAN_HOME', '/tmp/terran_home')

from tests.golden import ref_import as R   # noqa: E402
from tests.golden.make_golden import _load, save    # noqa: E402
from tests import wild_weights               # noqa: E402
from terran_amd import weights, synth        # noqa: E402


def main():
    RM = R.ref('terran.face.detection.retinaface.model')
    AM = R.ref('terran.face.recognition.arcface.model')
    PM = R.ref('terran.pose.openpose.model')

    def retina(st, img):
        x = torch.from_numpy(img.astype(np.float32)).permute(0, 3, 1, 2).flip(1).contiguous()
        with torch.no_grad():
            return [o.numpy() for o in _load(RM.RetinaFace, st)(x)]

    def pose(st, img):
        xp = torch.from_numpy((np.transpose(img, (0, 3, 1, 2)).astype(np.float32) / 255.0 - 0.5))
        with torch.no_grad():
            paf, hm = _load(PM.BodyPoseModel, st)(xp)
        return paf.numpy(), hm.numpy()

    # ---- working sizes, seeded benign weights
    img = synth.frames(31, 1, 184, 327)
    paf, hm = pose(weights.make_openpose_state(), img)
    save('worksize_openpose.npz', frames_seed=31, shape=np.array([1, 184, 327]), pafs=paf, heatmaps=hm,
         note='BodyPoseModel module on synth.frames(31, 1, 184, 327), seed %d' % weights.SEED_OPENPOSE)
    img = synth.frames(32, 1, 416, 739)
    outs = retina(weights.make_retinaface_state(), img)
    kw = {}
    for i, o in enumerate(outs):
        if i < 6:                                     # stride 32, 16: in full
            kw['out%d' % i] = o
        else:                                         # stride 8: per-channel sums (float64) + every 4th pixel
            kw['out%d_sum' % i] = o.astype(np.float64).sum((0, 2, 3))
            kw['out%d_abs_sum' % i] = np.abs(o.astype(np.float64)).sum((0, 2, 3))
            kw['out%d_every4' % i] = o[:, :, ::4, ::4].copy()
    save('worksize_retinaface.npz', frames_seed=32, shape=np.array([1, 416, 739]),
         note='RetinaFace module on synth.frames(32, 1, 416, 739) (BGR 0..255), seed %d; order cls/bbox/lmk x stride 32, 16, 8'
              % weights.SEED_RETINAFACE, **kw)

    # ---- wild-statistics weights (tests/wild_weights.py), small inputs, reference modules
    st = wild_weights.wild_retinaface_state()
    img = synth.frames(33, 1, 96, 128)
    outs = retina(st, img)
    save('wild_retinaface.npz', frames_seed=33, shape=np.array([1, 96, 128]), note='RetinaFace module, tests.wild_weights.wild_retinaface_state()',
         **{'out%d' % i: o for i, o in enumerate(outs)})
    st = wild_weights.wild_arcface_state()
    crops = np.random.default_rng(34).integers(0, 256, (2, 3, 112, 112), dtype=np.uint8)
    with torch.no_grad():
        emb = _load(AM.FaceResNet100, st)(torch.from_numpy(crops.astype(np.float32))).numpy()
    save('wild_arcface.npz', crops=crops, embeddings=emb, note='FaceResNet100 module, tests.wild_weights.wild_arcface_state()')
    st = wild_weights.wild_openpose_state()
    img = synth.frames(35, 1, 64, 96)
    paf, hm = pose(st, img)
    save('wild_openpose.npz', frames_seed=35, shape=np.array([1, 64, 96]), pafs=paf, heatmaps=hm,
         note='BodyPoseModel module, tests.wild_weights.wild_openpose_state()')


if __name__ == '__main__':
    main()

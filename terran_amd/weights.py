"""Seeded synthetic weights with the reference's `state_dict` key names/shapes.

There is no network in the build/bench environment and the pretrained `.pth`
files (terran/checkpoint.py:49-52,73-76,98-101) are not on disk, so every test,
fixture and bench in this repo uses random-init weights of the exact
architecture, generated here from a numpy `default_rng(seed)` stream (bit-stable
across machines).  A real checkpoint is the same `{key: array}` mapping, so
`load_state(path)` and these generators are interchangeable.

Initialisation (chosen so activations stay O(1) through ~100 layers):
  conv / linear : N(0, gain^2 / fan_in), gain^2 = 2 (ReLU nets) or 2/(1+0.25^2) (PReLU)
  BN            : gamma=1 (0.35 on the residual-closing BNs of ArcFace), beta~N(0,0.1),
                  running_mean~N(0,0.05), running_var~U(0.8,1.2)
  PReLU         : 0.25
  conv bias     : N(0, 0.05)
"""
import numpy as np

from . import arch

SEED_RETINAFACE = 100
SEED_ARCFACE = 101
SEED_OPENPOSE = 102


def _bn(rng, out, key, c, gamma=1.0):
    out[key + '.weight'] = np.full((c,), gamma, np.float32) + rng.normal(0, 0.02, c).astype(np.float32)
    out[key + '.bias'] = rng.normal(0, 0.1, c).astype(np.float32)
    out[key + '.running_mean'] = rng.normal(0, 0.05, c).astype(np.float32)
    out[key + '.running_var'] = rng.uniform(0.8, 1.2, c).astype(np.float32)
    out[key + '.num_batches_tracked'] = np.zeros((), np.int64)


def _fill(rng, specs, gain2, out_gain2=None, cls_bias_shift=0.0):
    out = {}
    for key, shape, kind in specs:
        if kind in ('conv', 'dw', 'linear', 'conv_out'):
            fan_in = int(np.prod(shape[1:]))
            g2 = gain2 if kind != 'conv_out' or out_gain2 is None else out_gain2
            out[key] = (rng.standard_normal(shape, dtype=np.float32)
                        * np.float32(np.sqrt(g2 / fan_in)))
        elif kind == 'bias':
            out[key] = rng.normal(0, 0.05, shape).astype(np.float32)
        elif kind == 'prelu':
            out[key] = np.full(shape, 0.25, np.float32)
        elif kind == 'bn':
            _bn(rng, out, key, shape[0])
        elif kind == 'bn_res':
            _bn(rng, out, key, shape[0], gamma=0.35)
        else:
            raise ValueError(kind)
    return out


def make_retinaface_state(seed=SEED_RETINAFACE, fg_bias=-4.0):
    """RetinaFace-mnet weights.  The detector eats raw 0..255 pixels
    (retinaface/wrapper.py:144-146), so the stem is scaled by 1/128 to keep
    activations O(1); the head weights are scaled so that box deltas stay small
    (exp() finite) and `fg_bias` shifts the foreground logits so that only a few
    percent of the anchors pass the 0.5 threshold (a random head would fire on
    half of them)."""
    rng = np.random.default_rng(seed)
    st = _fill(rng, arch.retinaface_param_specs(), gain2=2.0)
    st['base.first_conv_block.0.weight'] *= np.float32(1.0 / 128.0)
    A = arch.RETINA_NUM_ANCHORS
    for s in (8, 16, 32):
        st['outputs.cls_stride%d.weight' % s] *= np.float32(2.5)
        st['outputs.bbox_stride%d.weight' % s] *= np.float32(0.15)
        st['outputs.landmark_stride%d.weight' % s] *= np.float32(0.25)
        b = st['outputs.cls_stride%d.bias' % s]
        b[A:] += np.float32(fg_bias)
    return st


def make_arcface_state(seed=SEED_ARCFACE):
    rng = np.random.default_rng(seed)
    return _fill(rng, arch.arcface_param_specs(), gain2=2.0 / (1.0 + 0.25 ** 2))


def make_openpose_state(seed=SEED_OPENPOSE, out_scale=0.35):
    """OpenPose weights. `out_scale` scales the std of the per-stage output convs
    (the 38/19-channel 1x1s) so that heat-maps are O(0.3) like a trained model's
    rather than O(1) noise (keeps the peak count per part realistic)."""
    rng = np.random.default_rng(seed)
    return _fill(rng, arch.openpose_param_specs(), gain2=2.0, out_gain2=2.0 * out_scale ** 2)


def load_state(path):
    """Load a real Terran checkpoint (`torch.save(state_dict)`) as {key: ndarray}."""
    import torch
    sd = torch.load(path, map_location='cpu')
    return {k: v.numpy() for k, v in sd.items()}

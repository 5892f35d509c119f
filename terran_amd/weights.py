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


# ----------------------------------------------------------------------------------------------------
# OpenPose weights with an embedded DECODER path (end-to-end pose tests and bench.py)
# ----------------------------------------------------------------------------------------------------
# Random-weight OpenPose maps are structureless: no person ever assembles, so every end-to-end comparison of the
# pose path would be on empty lists.  `make_openpose_decoder_state` keeps the seeded random network and re-purposes a
# few channels of the VGG front and of stage 6 as an exact "pixel un-shuffle": a frame produced by
# `synth.pose_code_frames` carries, in the 8 x 8 pixel block of every map cell, the 57 heat-map / PAF values of that
# cell (R = data, G / B = row / column position codes), and the re-purposed channels route them to the network's
# outputs:
#   conv1_1/1_2 + pool1 : data AND (row parity == a) AND (column parity == b) -> 4 channels at 1/2 resolution
#   conv2_1/2_2 + pool2 : the same on the next position bit -> 16 channels at 1/4
#   conv3_1     + pool3 : ... -> 64 channels at 1/8 (one per position inside the 8 x 8 block)
#   conv3_2 .. conv4_4  : pass-through; feat[q] = data of in-block position q = 8 dy + dx
#   stage 6 (both branches): centre-tap pass-through of feat[0..37] -> PAF (2 d - 1), feat[38..56] -> heat-map (d),
#                         plus `noise` x the branch's random hidden channels (everything else stays random weights)
# ANDs are ReLU(d + s_y + s_x - 2) with s in {0, 1} recovered from the position codes by ReLU pairs, so the decode is
# exact up to float rounding in every arithmetic mode.  Stages 1-5 and all other channels keep their random weights
# (and never read the decoder channels), so the convolutions stay numerically busy.
def make_openpose_decoder_state(seed=SEED_OPENPOSE, noise=0.05):
    st = make_openpose_state(seed)
    ALPHA, BETA = 255.0 / 32.0, 0.5 * 255.0 / 32.0 - 0.5          # v = ALPHA x + BETA = position code 0..7

    def layer(key):
        return st[key + '.weight'], st[key + '.bias']

    def claim(key, n_out, ins):
        """Zero the decoder rows 0..n_out-1 and the random rows' taps on the decoder inputs `ins`."""
        W, b = layer(key)
        W[:n_out] = 0.0
        b[:n_out] = 0.0
        if ins is not None:
            W[n_out:, ins] = 0.0
        return W, b, W.shape[2] // 2                               # centre tap

    # ---- level 1 -----------------------------------------------------------------------------------
    W, b, c = claim('model0.conv1_1', 7, None)
    W[0, 0, c, c], b[0] = 1.0, 0.5                                  # D  = ReLU(x_R + 0.5) = data
    for o, ch, off in ((1, 1, 0.0), (2, 2, 0.0), (3, 1, -3.0), (4, 1, -4.0), (5, 2, -3.0), (6, 2, -4.0)):
        W[o, ch, c, c], b[o] = ALPHA, BETA + off                    # Vy, Vx, Ay, By, Ax, Bx
    W, b, c = claim('model0.conv1_2', 6, slice(0, 7))
    for a in (0, 1):
        for bb in (0, 1):
            o = a * 2 + bb
            W[o, 0, c, c] = 1.0
            sy, sx = (1.0 if a else -1.0), (1.0 if bb else -1.0)
            W[o, 3, c, c], W[o, 4, c, c] = sy, -sy                  # [y0 == a]: +-(Ay - By) (+1 in the bias for a = 0)
            W[o, 5, c, c], W[o, 6, c, c] = sx, -sx
            b[o] = -2.0 + (0.0 if a else 1.0) + (0.0 if bb else 1.0)
    W[4, 1, c, c], W[4, 3, c, c], W[4, 4, c, c] = 1.0, -4.0, 4.0    # V'y = Vy - 4 y0
    W[5, 2, c, c], W[5, 5, c, c], W[5, 6, c, c] = 1.0, -4.0, 4.0
    # ---- level 2 -----------------------------------------------------------------------------------
    W, b, c = claim('model0.conv2_1', 10, slice(0, 6))
    for o in range(6):
        W[o, o, c, c] = 1.0                                         # D_ab, V'y, V'x pass through
    for o, ch, off in ((6, 4, -1.0), (7, 4, -2.0), (8, 5, -1.0), (9, 5, -2.0)):
        W[o, ch, c, c], b[o] = 1.0, off                             # A'y, B'y, A'x, B'x
    W, b, c = claim('model0.conv2_2', 18, slice(0, 10))
    for ab in range(4):
        for a1 in (0, 1):
            for b1 in (0, 1):
                o = ab * 4 + a1 * 2 + b1
                W[o, ab, c, c] = 1.0
                sy, sx = (1.0 if a1 else -1.0), (1.0 if b1 else -1.0)
                W[o, 6, c, c], W[o, 7, c, c] = sy, -sy
                W[o, 8, c, c], W[o, 9, c, c] = sx, -sx
                b[o] = -2.0 + (0.0 if a1 else 1.0) + (0.0 if b1 else 1.0)
    W[16, 4, c, c], W[16, 6, c, c], W[16, 7, c, c] = 1.0, -2.0, 2.0   # y2 = V'y - 2 y1
    W[17, 5, c, c], W[17, 8, c, c], W[17, 9, c, c] = 1.0, -2.0, 2.0
    # ---- level 3 -----------------------------------------------------------------------------------
    W, b, c = claim('model0.conv3_1', 64, slice(0, 18))
    for i16 in range(16):
        for a2 in (0, 1):
            for b2 in (0, 1):
                o = i16 * 4 + a2 * 2 + b2
                W[o, i16, c, c] = 1.0
                W[o, 16, c, c] = 1.0 if a2 else -1.0
                W[o, 17, c, c] = 1.0 if b2 else -1.0
                b[o] = -2.0 + (0.0 if a2 else 1.0) + (0.0 if b2 else 1.0)
    for key in ('conv3_2', 'conv3_3', 'conv3_4'):
        W, b, c = claim('model0.' + key, 64, slice(0, 64))
        for o in range(64):
            W[o, o, c, c] = 1.0
    # channel ((a,b),(a1,b1),(a2,b2)) holds in-block position dy = a + 2 a1 + 4 a2, dx = b + 2 b1 + 4 b2
    W, b, c = claim('model0.conv4_1', 57, slice(0, 64))
    for q in range(57):
        dy, dx = q // 8, q % 8
        a, a1, a2, bb, b1, b2 = dy & 1, (dy >> 1) & 1, dy >> 2, dx & 1, (dx >> 1) & 1, dx >> 2
        W[q, ((a * 2 + bb) * 4 + a1 * 2 + b1) * 4 + a2 * 2 + b2, c, c] = 1.0
    for key in ('conv4_2', 'conv4_3_CPM', 'conv4_4_CPM'):
        W, b, c = claim('model0.' + key, 57, slice(0, 57))
        for o in range(57):
            W[o, o, c, c] = 1.0
    # ---- stage 6: feat[0..37] -> PAF, feat[38..56] -> heat-map ------------------------------------------
    for br, n_out, f0 in ((1, 38, 0), (2, 19, 38)):
        p = 'model6_%d.' % br
        L = 'L%d' % br
        W, b, c = claim(p + 'Mconv1_stage6_' + L, n_out, slice(57, 57 + 57))   # stage input = cat[PAF38, HM19, feat128]
        for o in range(n_out):
            W[o, 57 + f0 + o, c, c] = 1.0
        for i in range(2, 7):
            W, b, c = claim(p + 'Mconv%d_stage6_%s' % (i, L), n_out, slice(0, n_out))
            for o in range(n_out):
                W[o, o, c, c] = 1.0
        W, b = layer(p + 'Mconv7_stage6_' + L)
        W *= np.float32(noise)                                      # the random hidden channels become low-level noise
        b[:] = 0.0
        W[:, :n_out] = 0.0
        for o in range(n_out):
            W[o, o, 0, 0] = 2.0 if br == 1 else 1.0                 # PAF = 2 d - 1, heat-map = d
        if br == 1:
            b[:] = -1.0
    return st

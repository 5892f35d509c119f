"""-m gpu: the `f16x3` mode's half-float range guard.

Activations between convs are stored as two half floats (hi + lo, 22 significant bits); |x| > 65504 does not fit.  The
conv epilogues that write that format raise a device flag, the call fails with TA_E_RANGE (never with numbers), and the
wrapper classes re-run the batch on an exact-f32 model of the same weights.  The reference has no such limit
(float32 activations, openpose/model.py:27-141); with the seeded and with real weights activations stay below ~1e2.
"""
import numpy as np
import pytest

from terran_amd import pack, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from terran_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


def _two_convs(precision, gain, c1=64, split_mid=True):
    """frames -> conv3x3 (x gain) -> conv3x3 (x 1 / gain): the middle tensor grows with `gain`, the output does not."""
    rng = np.random.default_rng(3)
    P = pack.Program(pack.MODEL_OPENPOSE, precision)
    t0 = P.tensor(4, 1)
    P.input_tensor = t0
    t1 = P.tensor(c1, 1, name='mid', f32=not split_mid)
    W1 = rng.normal(0, 0.3, (c1, 3, 3, 3)).astype(np.float32) * np.float32(gain)
    b1 = rng.normal(0, 0.1, c1).astype(np.float32) * np.float32(gain)
    P.conv(t0, t1, W1, b1, act=pack.ACT_RELU)
    t2 = P.tensor(c1, 0, name='out', f32=True)
    W2 = rng.normal(0, 0.05, (c1, c1, 3, 3)).astype(np.float32) / np.float32(gain)
    b2 = rng.normal(0, 0.1, c1).astype(np.float32)
    P.conv(t1, t2, W2, b2)
    P.outputs = [t2]
    return P


def _forward_checked(ctx, model, frames):
    """Network only (debug tap), then the flag exactly as the task entry points read it."""
    model.forward_frames(frames)
    return ctx.lib.ta_debug_range_check(ctx.h)


def test_range_flag_is_raised_by_the_epilogue_and_cleared(ctx):
    from terran_amd import lib
    frames = ctx.upload(synth.frames(9, 2, 24, 40))
    ok = lib.Model(ctx, _two_convs('f16x3', 1.0))
    assert _forward_checked(ctx, ok, frames) == lib.OK
    ref = ok.read('out')
    # 2^20 x larger middle tensor: ~1e5 > 65504 -> the generic kernel's epilogue (Cin = 3 stem) raises the flag
    big = lib.Model(ctx, _two_convs('f16x3', 2.0 ** 20))
    assert _forward_checked(ctx, big, frames) == lib.E_RANGE
    assert 'half-float range' in ctx.last_error()
    # the flag does not stick: the in-range program is clean again, and bit-identical to its first run
    assert _forward_checked(ctx, ok, frames) == lib.OK
    assert np.array_equal(ok.read('out'), ref)
    # a float32 middle tensor is checked as well: the conv that reads it splits it into half floats in registers
    wide = lib.Model(ctx, _two_convs('f16x3', 2.0 ** 20, split_mid=False))
    assert _forward_checked(ctx, wide, frames) == lib.E_RANGE
    # the other modes never raise it
    for prec in ('f32', 'bf16x3'):
        assert _forward_checked(ctx, lib.Model(ctx, _two_convs(prec, 2.0 ** 20)), frames) == lib.OK


@pytest.mark.parametrize('mode', ['f16x3', 'f16'])
def test_split_role_epilogues_raise_the_flag(ctx, mode):
    """The lean (conv_drain_fast) and the generic LDS-staged drains of the split-role kernel: a 64 -> 64 conv whose
    OUTPUT is a split tensor beyond the range."""
    from terran_amd import lib
    rng = np.random.default_rng(4)
    frames = ctx.upload(synth.frames(10, 2, 24, 40))
    for act, gain, last_gain, expect in ((pack.ACT_RELU, 1.0, 1.0, lib.OK), (pack.ACT_RELU, 2.0 ** 22, None, lib.E_RANGE),
                                         (pack.ACT_NONE, -2.0 ** 22, None, lib.E_RANGE), (pack.ACT_PRELU, 2.0 ** 22, None, lib.E_RANGE),
                                         (pack.ACT_RELU, 1.0, 2.0 ** 22, lib.E_RANGE)):      # last: the float32 output, generic drain
        P = pack.Program(pack.MODEL_OPENPOSE, mode)               # 'f16': the same flag on 2-byte half-float tensors
        t0 = P.tensor(4, 1)
        P.input_tensor = t0
        t1 = P.tensor(64, 1, name='mid')
        P.conv(t0, t1, rng.normal(0, 0.3, (64, 3, 3, 3)).astype(np.float32), np.zeros(64, np.float32), act=pack.ACT_RELU)
        t2 = P.tensor(64, 1, name='big')                         # split format: read by the next conv
        W2 = rng.normal(0, 0.05, (64, 64, 3, 3)).astype(np.float32) * np.float32(gain)
        kw = dict(prelu=np.full(64, 0.25, np.float32)) if act == pack.ACT_PRELU else {}
        P.conv(t1, t2, W2, np.zeros(64, np.float32), act=act, **kw)
        t3 = P.tensor(64, 0, name='out', f32=True)
        P.conv(t2, t3, rng.normal(0, 0.05, (64, 64, 3, 3)).astype(np.float32) * np.float32(last_gain or 1.0 / abs(gain)), np.zeros(64, np.float32))
        P.outputs = [t3]
        ctx.conv_counts(reset=True)
        m = lib.Model(ctx, P)
        rc = _forward_checked(ctx, m, frames)
        assert rc == expect, (act, gain, last_gain, rc, ctx.conv_counts())
        assert ctx.conv_counts().get('lean_epilogue', 0) >= 1


def test_openpose_wrapper_falls_back_to_f32(states):
    """conv1_1 scaled by 2^24 and conv1_2 by 2^-24 (powers of two, ReLU is positively homogeneous: the network computes
    the same numbers, bit for bit in float32) -- but conv1_1's output reaches ~1e7.  The f16x3 wrapper must hand the batch
    to its exact-f32 twin and return exactly what an f32 wrapper returns."""
    from terran_amd import OpenPose
    sd = dict(states('openpose_decoder'))
    g = np.float32(2.0 ** 24)
    sd['model0.conv1_1.weight'] = np.asarray(sd['model0.conv1_1.weight'], np.float32) * g
    sd['model0.conv1_1.bias'] = np.asarray(sd['model0.conv1_1.bias'], np.float32) * g
    sd['model0.conv1_2.weight'] = np.asarray(sd['model0.conv1_2.weight'], np.float32) / g
    frames = synth.pose_code_frames(81, 3, 96, 128, 3)
    a = OpenPose(device=0, short_side=96, state=sd, precision='f16x3')
    b = OpenPose(device=0, short_side=96, state=sd, precision='f32')
    ra, rb = a.call(frames), b.call(frames)
    assert a.fallbacks == 1 and b.fallbacks == 0
    assert [len(p) for p in ra] == [len(p) for p in rb] and sum(len(p) for p in ra) >= 3
    for pa, pb in zip(ra, rb):
        for x, y in zip(pa, pb):
            assert np.array_equal(x['keypoints'], y['keypoints']) and x['score'] == y['score']
    # an ordinary batch afterwards runs on the f16x3 model again
    c = OpenPose(device=0, short_side=96, state=states('openpose_decoder'), precision='f16x3')
    c.call(frames)
    assert c.fallbacks == 0


@pytest.mark.parametrize('mode', ['f16x3', 'f16'])
def test_arcface_wrapper_falls_back_to_f32(states, mode):
    from terran_amd import ArcFace
    sd = dict(states('arcface'))
    g = np.float32(2.0 ** 22)
    # the stem's BatchNorm scale x 2^22 and the first unit's leading BatchNorm / 2^22: same network, huge stem output
    for k in ('weight', 'bias'):
        sd['initial_layer.1.' + k] = np.asarray(sd['initial_layer.1.' + k], np.float32) * g
    sd['stages.0.0.body.0.running_mean'] = np.asarray(sd['stages.0.0.body.0.running_mean'], np.float32) * g
    sd['stages.0.0.body.0.running_var'] = np.asarray(sd['stages.0.0.body.0.running_var'], np.float32) * g * g
    crops = np.random.default_rng(8).integers(0, 256, (5, 3, 112, 112), dtype=np.uint8)
    a = ArcFace(device=0, state=sd, precision=mode)
    b = ArcFace(device=0, state=sd, precision='f32')
    ea, eb = a.embed_crops(crops), b.embed_crops(crops)
    assert a.fallbacks == 1 and np.array_equal(ea, eb) and np.isfinite(ea).all()


def test_retinaface_wrapper_falls_back_to_f32(states):
    """The detector's refiner runs on the split-half MFMA in the f16x3 mode: a lateral map beyond 65504 (its BatchNorm
    scaled by 2^24 here) must send the batch to the exact-f32 twin -- same detections as an f32 wrapper, bit for bit."""
    from terran_amd import RetinaFace
    sd = dict(states('retinaface'))
    g = np.float32(2.0 ** 24)
    for k in ('weight', 'bias'):
        sd['refiner.conv_stride8.1.' + k] = np.asarray(sd['refiner.conv_stride8.1.' + k], np.float32) * g
    sd['refiner.aggr_stride8.0.weight'] = np.asarray(sd['refiner.aggr_stride8.0.weight'], np.float32) / g
    frames = synth.frames(12, 3, 96, 128)
    a = RetinaFace(device=0, state=sd, precision='f16x3')
    b = RetinaFace(device=0, state=sd, precision='f32')
    ra, rb = a.call(frames), b.call(frames)
    assert a.fallbacks == 1 and b.fallbacks == 0
    assert [len(x) for x in ra] == [len(x) for x in rb]
    for x, y in zip(ra, rb):
        for p, q in zip(x, y):
            assert np.array_equal(p['bbox'], q['bbox']) and p['score'] == q['score']
    c = RetinaFace(device=0, state=states('retinaface'), precision='f16x3')
    c.call(frames)
    assert c.fallbacks == 0

"""-m gpu: the `f16x3` mode's half-float range guard.

Activations between convs are stored as two half floats (hi + lo, 22 significant bits); |x| > 65504 does not fit.  Every
tensor is stored times a pack-time power of two that puts its EXPECTED maximum near 2^10 (pack.Program.tensor_scales), so
weights whose activations are merely large or small run as they are; what still leaves the range (64 x beyond the
expectation) raises a device flag in the epilogue that stores it -- whatever that op's own arithmetic mode, and also for
the depthwise intermediate of a dw + pw block that is split into half floats in registers -- the call fails with TA_E_RANGE
(never with numbers), and the wrapper classes re-run the batch on an exact-f32 model of the same weights.  The reference
has no such limit (float32 activations, openpose/model.py:27-141).

The programs below provoke the condition with `forced_scale` (a tensor stored times a deliberately wrong power of two) or
with the pack-time switch TERRAN_AMD_NO_ACT_SCALES (every tensor stored unscaled, the round-3 behaviour).
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


def _two_convs(precision, gain, c1=64, split_mid=True, mid_scale=None):
    """frames -> conv3x3 (x gain) -> conv3x3 (x 1 / gain): the middle tensor grows with `gain`, the output does not.
    mid_scale: store the middle tensor times 2^mid_scale instead of what the packer would choose."""
    rng = np.random.default_rng(3)
    P = pack.Program(pack.MODEL_OPENPOSE, precision)
    t0 = P.tensor(4, 1)
    P.input_tensor = t0
    P.input_stats = (np.array([-0.05] * 3 + [0.0]), np.array([0.08] * 3 + [0.0]))
    t1 = P.tensor(c1, 1, name='mid', f32=False)
    if not split_mid:
        P.tensors[t1] = (P.tensors[t1][0] + 4, P.tensors[t1][1], P.tensors[t1][2])     # 68 channels: not a multiple of 32 -> float32 storage
    W1 = rng.normal(0, 0.3, (c1, 3, 3, 3)).astype(np.float32) * np.float32(gain)
    b1 = rng.normal(0, 0.1, c1).astype(np.float32) * np.float32(gain)
    P.conv(t0, t1, W1, b1, act=pack.ACT_RELU)
    t2 = P.tensor(c1, 0, name='out', f32=True)
    W2 = rng.normal(0, 0.05, (c1, c1, 3, 3)).astype(np.float32) / np.float32(gain)
    b2 = rng.normal(0, 0.1, c1).astype(np.float32)
    P.conv(t1, t2, W2, b2)
    P.outputs = [t2]
    if mid_scale is not None:
        P.forced_scale[t1] = mid_scale
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
    # a 2^20 x larger middle tensor (~1e5 unscaled) is nothing special any more: the packer stores it times 2^-k ...
    P = _two_convs('f16x3', 2.0 ** 20)
    scaled = lib.Model(ctx, P)
    assert P.scales[1].max() <= -8
    assert _forward_checked(ctx, scaled, frames) == lib.OK
    assert np.allclose(scaled.read('out'), ref, rtol=0, atol=2e-6 * np.abs(ref).max())
    assert np.allclose(scaled.read('mid') / 2.0 ** 20, ok.read('mid'), rtol=0, atol=1e-6 * np.abs(ok.read('mid')).max())
    # ... but stored times 2^14 too much (a tensor 2^14 beyond what its weights predict) the generic kernel's epilogue
    # (Cin = 3 stem) raises the flag
    big = lib.Model(ctx, _two_convs('f16x3', 1.0, mid_scale=int(P.scales[1].max()) + 20 + 14))
    assert _forward_checked(ctx, big, frames) == lib.E_RANGE
    assert 'half-float range' in ctx.last_error()
    # the flag does not stick: the in-range program is clean again, and bit-identical to its first run
    assert _forward_checked(ctx, ok, frames) == lib.OK
    assert np.array_equal(ok.read('out'), ref)
    # a float32 middle tensor is checked as well: the conv that reads it splits it into half floats in registers
    Pw = _two_convs('f16x3', 1.0, split_mid=False)
    Pw.blob()
    wide = lib.Model(ctx, _two_convs('f16x3', 1.0, split_mid=False, mid_scale=int(Pw.scales[1].max()) + 14))
    assert Pw.tensor_formats()[1] == pack.FMT_F32
    assert _forward_checked(ctx, wide, frames) == lib.E_RANGE
    # the other modes never raise it (and store everything unscaled)
    for prec in ('f32', 'bf16x3'):
        assert _forward_checked(ctx, lib.Model(ctx, _two_convs(prec, 2.0 ** 20)), frames) == lib.OK


@pytest.mark.parametrize('mode', ['f16x3', 'f16', 'f16x2'])
def test_split_role_epilogues_raise_the_flag(ctx, mode):
    """The lean (conv_drain_fast) and the generic LDS-staged drains of the split-role kernel: a 64 -> 64 conv whose
    OUTPUT is a split tensor stored 2^14 beyond the range its weights predict (forced_scale)."""
    from terran_amd import lib
    rng = np.random.default_rng(4)
    frames = ctx.upload(synth.frames(10, 2, 24, 40))
    for act, sign, over_mid, over_out, expect in ((pack.ACT_RELU, 1.0, 0, 0, lib.OK), (pack.ACT_RELU, 1.0, 14, 0, lib.E_RANGE),
                                                  (pack.ACT_NONE, -1.0, 14, 0, lib.E_RANGE), (pack.ACT_PRELU, 1.0, 14, 0, lib.E_RANGE),
                                                  (pack.ACT_RELU, 1.0, 0, 2.0 ** 20, lib.OK)):   # last: a float32 RESULT no op reads may hold
                                                                                                 # anything (nothing splits it into half floats)
        P = pack.Program(pack.MODEL_OPENPOSE, mode)               # 'f16': the same flag on 2-byte half-float tensors
        t0 = P.tensor(4, 1)
        P.input_tensor = t0
        P.input_stats = (np.array([-0.05] * 3 + [0.0]), np.array([0.08] * 3 + [0.0]))
        t1 = P.tensor(64, 1, name='mid')
        P.conv(t0, t1, rng.normal(0, 0.3, (64, 3, 3, 3)).astype(np.float32), np.zeros(64, np.float32), act=pack.ACT_RELU)
        t2 = P.tensor(64, 1, name='big')                         # split format: read by the next conv
        W2 = rng.normal(0, 0.05, (64, 64, 3, 3)).astype(np.float32) * np.float32(sign)
        kw = dict(prelu=np.full(64, 0.25, np.float32)) if act == pack.ACT_PRELU else {}
        P.conv(t1, t2, W2, np.zeros(64, np.float32), act=act, **kw)
        t3 = P.tensor(64, 0, name='out', f32=True)
        P.conv(t2, t3, rng.normal(0, 0.05, (64, 64, 3, 3)).astype(np.float32) * np.float32(over_out or 1.0), np.zeros(64, np.float32))
        P.outputs = [t3]
        if over_mid:
            P.forced_scale[t2] = int(P.tensor_scales()[t2].max()) + over_mid
        ctx.conv_counts(reset=True)
        m = lib.Model(ctx, P)
        rc = _forward_checked(ctx, m, frames)
        assert rc == expect, (act, sign, over_mid, over_out, rc, ctx.conv_counts())
        assert ctx.conv_counts().get('lean_epilogue', 0) >= 1


def _dwpw_program(dw_prec, pw_gain=1.0, mid_over=0, out_over=0):
    """frames -> conv (4 -> 64, exact f32) -> [dw3x3 -> 1x1] (f32 or f16x3) -> conv 64 -> 64 (f16x3) -> float32 out: the
    shape of the detector's base, where an exact-f32 block hands a float32 tensor to a split-half op."""
    rng = np.random.default_rng(6)
    P = pack.Program(pack.MODEL_OPENPOSE, 'f16x3')
    t0 = P.tensor(4, 1)
    P.input_tensor = t0
    P.input_stats = (np.array([-0.05] * 3 + [0.0]), np.array([0.08] * 3 + [0.0]))
    t1 = P.tensor(64, 1)
    P.conv(t0, t1, rng.normal(0, 0.3, (64, 3, 3, 3)).astype(np.float32), np.zeros(64, np.float32), act=pack.ACT_RELU, precision='f32')
    t2 = P.tensor(64, 1, name='feat')
    P.dwpw(t1, t2, rng.normal(0, 0.3, (64, 1, 3, 3)).astype(np.float32), rng.normal(0, 0.1, 64).astype(np.float32),
           rng.normal(0, 0.1, (64, 64, 1, 1)).astype(np.float32) * np.float32(pw_gain), rng.normal(0, 0.1, 64).astype(np.float32) * np.float32(pw_gain),
           precision=dw_prec)
    t3 = P.tensor(64, 0, name='out', f32=True)
    P.conv(t2, t3, rng.normal(0, 0.05, (64, 64, 1, 1)).astype(np.float32) / np.float32(pw_gain), np.zeros(64, np.float32), precision='f16x3')
    P.outputs = [t3]
    if mid_over or out_over:
        ref = _dwpw_program(dw_prec, pw_gain)                    # what the packer would choose
        ref.blob()
        if mid_over:
            P.forced_scale[('mid', 1)] = int(ref.mid_scales[1].max()) + mid_over
        if out_over:
            P.forced_scale[t2] = int(ref.scales[t2].max()) + out_over
    return P


def test_dwpw_blocks_are_range_checked(ctx):
    """a value above 65504 written by an EXACT-F32 producer into a float32 tensor (the detector's base blocks), or
    the depthwise intermediate of an f16x3 dw + pw block, used to be split into hi = +inf, lo = -inf in registers: NaN sums,
    masked to 0 by the ReLU, no flag.  Every store of a program with half-float ops is checked now, the depthwise rows too,
    and the ReLU keeps a NaN a NaN."""
    from terran_amd import lib
    frames = ctx.upload(synth.frames(14, 2, 24, 40))
    for dw_prec in ('f32', 'f16x3'):
        ok = lib.Model(ctx, _dwpw_program(dw_prec))
        assert _forward_checked(ctx, ok, frames) == lib.OK
        ref = ok.read('out')
        assert np.isfinite(ref).all() and np.abs(ref).max() > 0
        # a 2^16 x larger block output: the packer stores it scaled, same numbers
        g = lib.Model(ctx, _dwpw_program(dw_prec, pw_gain=2.0 ** 16))
        assert _forward_checked(ctx, g, frames) == lib.OK
        assert np.allclose(g.read('out'), ref, rtol=0, atol=3e-6 * np.abs(ref).max())
        # the block's OUTPUT (a float32 tensor, written by an exact-f32 op when dw_prec == 'f32') 2^14 beyond its expectation
        bad = lib.Model(ctx, _dwpw_program(dw_prec, out_over=14))
        assert _forward_checked(ctx, bad, frames) == lib.E_RANGE, dw_prec
        assert _forward_checked(ctx, ok, frames) == lib.OK
    # the depthwise INTERMEDIATE of the split-half block
    bad = lib.Model(ctx, _dwpw_program('f16x3', mid_over=14))
    assert _forward_checked(ctx, bad, frames) == lib.E_RANGE


def _no_scales(monkeypatch):
    """Pack as round 3 did: every tensor stored unscaled, so that re-parametrised weights DO leave the half-float range."""
    monkeypatch.setenv('TERRAN_AMD_NO_ACT_SCALES', '1')


def test_openpose_wrapper_falls_back_to_f32(states, monkeypatch):
    """conv1_1 scaled by 2^24 and conv1_2 by 2^-24 (powers of two, ReLU is positively homogeneous: the network computes
    the same numbers, bit for bit in float32) -- conv1_1's output reaches ~1e7.  With activation scales (the default) the
    f16x3 wrapper just runs it: no fallback, the exact-f32 wrapper's people.  Stored unscaled (TERRAN_AMD_NO_ACT_SCALES) the
    tensor leaves the half-float range, and the wrapper must hand the batch to its exact-f32 twin and return exactly what
    an f32 wrapper returns."""
    from terran_amd import OpenPose
    sd = dict(states('openpose_decoder'))
    g = np.float32(2.0 ** 24)
    sd['model0.conv1_1.weight'] = np.asarray(sd['model0.conv1_1.weight'], np.float32) * g
    sd['model0.conv1_1.bias'] = np.asarray(sd['model0.conv1_1.bias'], np.float32) * g
    sd['model0.conv1_2.weight'] = np.asarray(sd['model0.conv1_2.weight'], np.float32) / g
    frames = synth.pose_code_frames(81, 3, 96, 128, 3)
    b = OpenPose(device=0, short_side=96, state=sd, precision='f32')
    rb = b.call(frames)
    s = OpenPose(device=0, short_side=96, state=sd, precision='f16x3')
    rs = s.call(frames)
    assert s.fallbacks == 0 and b.fallbacks == 0
    _no_scales(monkeypatch)
    a = OpenPose(device=0, short_side=96, state=sd, precision='f16x3')
    with pytest.warns(RuntimeWarning, match='TA_E_RANGE'):           # the first fallback of a model is announced
        ra = a.call(frames)
    assert a.fallbacks == 1
    for r in (ra, rs):
        assert [len(p) for p in r] == [len(p) for p in rb] and sum(len(p) for p in r) >= 3
        for pa, pb in zip(r, rb):
            for x, y in zip(pa, pb):
                assert np.array_equal(x['keypoints'], y['keypoints'])
    for pa, pb in zip(ra, rb):
        for x, y in zip(pa, pb):
            assert x['score'] == y['score']                    # the fallback IS the f32 model
    # an ordinary batch afterwards runs on the f16x3 model again
    c = OpenPose(device=0, short_side=96, state=states('openpose_decoder'), precision='f16x3')
    c.call(frames)
    assert c.fallbacks == 0


@pytest.mark.parametrize('mode', ['f16x3', 'f16', 'f16x2'])
def test_arcface_wrapper_falls_back_to_f32(states, mode, monkeypatch):
    from terran_amd import ArcFace
    sd = dict(states('arcface'))
    g = np.float32(2.0 ** 22)
    # the stem's BatchNorm scale x 2^22 and the first unit's leading BatchNorm / 2^22: same network, huge stem output
    for k in ('weight', 'bias'):
        sd['initial_layer.1.' + k] = np.asarray(sd['initial_layer.1.' + k], np.float32) * g
    sd['stages.0.0.body.0.running_mean'] = np.asarray(sd['stages.0.0.body.0.running_mean'], np.float32) * g
    sd['stages.0.0.body.0.running_var'] = np.asarray(sd['stages.0.0.body.0.running_var'], np.float32) * g * g
    crops = np.random.default_rng(8).integers(0, 256, (5, 3, 112, 112), dtype=np.uint8)
    b = ArcFace(device=0, state=sd, precision='f32')
    eb = b.embed_crops(crops)
    s = ArcFace(device=0, state=sd, precision=mode)               # activation scales: runs as it is
    es = s.embed_crops(crops)
    assert s.fallbacks == 0 and np.abs(es - eb).max() < (1e-3 if mode in ('f16', 'f16x2') else 2e-6)
    _no_scales(monkeypatch)
    a = ArcFace(device=0, state=sd, precision=mode)
    ea = a.embed_crops(crops)
    assert a.fallbacks == 1 and np.array_equal(ea, eb) and np.isfinite(ea).all()


def test_retinaface_wrapper_falls_back_to_f32(states, monkeypatch):
    """The detector's refiner runs on the split-half MFMA in the f16x3 mode: a lateral map beyond 65504 when stored unscaled
    (its BatchNorm scaled by 2^24 here) must send the batch to the exact-f32 twin -- same detections as an f32 wrapper, bit
    for bit; with activation scales it runs on the split-half MFMA and finds the same detections."""
    from terran_amd import RetinaFace
    sd = dict(states('retinaface'))
    g = np.float32(2.0 ** 24)
    for k in ('weight', 'bias'):
        sd['refiner.conv_stride8.1.' + k] = np.asarray(sd['refiner.conv_stride8.1.' + k], np.float32) * g
    sd['refiner.aggr_stride8.0.weight'] = np.asarray(sd['refiner.aggr_stride8.0.weight'], np.float32) / g
    frames = synth.frames(12, 3, 96, 128)
    b = RetinaFace(device=0, state=sd, precision='f32')
    rb = b.call(frames)
    s = RetinaFace(device=0, state=sd, precision='f16x3')
    rs = s.call(frames)
    assert s.fallbacks == 0 and b.fallbacks == 0
    _no_scales(monkeypatch)
    a = RetinaFace(device=0, state=sd, precision='f16x3')
    ra = a.call(frames)
    assert a.fallbacks == 1
    assert [len(x) for x in ra] == [len(x) for x in rb] == [len(x) for x in rs]
    for x, y, z in zip(ra, rb, rs):
        for p, q, r in zip(x, y, z):
            assert np.array_equal(p['bbox'], q['bbox']) and p['score'] == q['score']
            assert np.array_equal(r['bbox'], q['bbox']) and abs(float(r['score']) - float(q['score'])) < 1e-5
    c = RetinaFace(device=0, state=states('retinaface'), precision='f16x3')
    c.call(frames)
    assert c.fallbacks == 0


def test_range_error_wins_over_post_processing_errors(states, monkeypatch):
    """when overflowed (garbage) maps make the grouping itself fail, the caller must still get TA_E_RANGE -- the
    error the wrappers act on -- and the flag must not survive into the next call on the context."""
    from terran_amd import OpenPose, lib
    _no_scales(monkeypatch)
    sd = dict(states('openpose'))
    g = np.float32(2.0 ** 30)
    sd['model0.conv1_1.weight'] = np.asarray(sd['model0.conv1_1.weight'], np.float32) * g
    sd['model0.conv1_1.bias'] = np.asarray(sd['model0.conv1_1.bias'], np.float32) * g
    sd['model0.conv1_2.weight'] = np.asarray(sd['model0.conv1_2.weight'], np.float32) / g
    frames = synth.frames(15, 2, 96, 128)
    a = OpenPose(device=0, short_side=96, state=sd, precision='f16x3')
    a.call(frames)
    assert a.fallbacks == 1
    ok = OpenPose(device=0, short_side=96, state=states('openpose'), precision='f16x3', ctx=a.ctx)
    ok.call(frames)                                              # same context, next call: no stale TA_E_RANGE
    assert ok.fallbacks == 0

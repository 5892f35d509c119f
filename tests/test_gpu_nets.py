"""-m gpu: the three networks on the HIP path (through the C ABI) against the oracle.

BASELINE.json's bar is "within 1e-3 fp32"; these tests hold the kernels to ~10x the error they actually measure
(relative to max|reference| of each tensor): f32 = exact-f32 MFMA, only the summation order differs from torch
(measured <= 3e-6 after ~100 layers); bf16x3 drops the lo*lo product terms (measured <= 2.3e-5).  Every comparison
prints the error it achieved.
"""
import numpy as np
import pytest
import torch

from tests.util import golden
from terran_amd import pack, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from terran_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


NET_TOL = {'f32': 3e-5, 'f16x3': 3e-5, 'bf16x3': 2e-4}
_prec = ['f32']


def _close(a, b, tol=None, what=''):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    tol = NET_TOL[_prec[0]] if tol is None else tol
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert a.shape == b.shape, (what, a.shape, b.shape)
    print('  %-16s max abs err %.2e / scale %.2e = %.2e (tol %.0e)' % (what, err, scale, err / scale, tol))
    assert err <= tol * scale, '%s: max abs err %.3e (scale %.3e)' % (what, err, scale)
    return err


PRECISIONS = ['f32', 'f16x3', 'bf16x3']      # both parity-grade modes; 'bf16' (throughput mode) is not expected to pass


@pytest.mark.parametrize('precision', PRECISIONS)
def test_openpose_net(ctx, states, precision):
    from terran_amd import lib
    _prec[0] = precision
    from oracle import nets
    sd = states('openpose')
    m = lib.Model(ctx, pack.pack_openpose(sd, precision))
    g = golden('nets_openpose.npz')
    for images in (g['images'], synth.frames(40, 2, 72, 104)):
        fr = ctx.upload(images)
        m.forward_frames(fr)
        x = torch.from_numpy(np.transpose(images, (0, 3, 1, 2)).astype(np.float32) / 255.0 - 0.5)
        taps = {}
        paf, hm = nets.openpose_forward(sd, x, taps)
        _close(m.read('feat'), taps['feat'].numpy(), what='feat')
        # the stage tensors ping-pong between two concat buffers: only stage 5 (and 6) survive
        _close(m.read('stage5_paf'), taps['stage5_paf'].numpy(), what='stage5 paf')
        _close(m.read('stage5_hm'), taps['stage5_hm'].numpy(), what='stage5 hm')
        e1 = _close(m.read('pafs'), paf.numpy(), what='pafs')
        e2 = _close(m.read('heatmaps'), hm.numpy(), what='heatmaps')
        print('openpose', precision, 'max err', e1, e2)
    g_p, g_h = g['pafs'], g['heatmaps']
    fr = ctx.upload(g['images'])
    m.forward_frames(fr)
    _close(m.read('pafs'), g_p, what='golden pafs')           # vs the reference module's own output
    _close(m.read('heatmaps'), g_h, what='golden heatmaps')
    assert m.read('heatmaps').min() >= 0.0                     # stage-6 heat-map ReLU quirk


@pytest.mark.parametrize('precision', PRECISIONS)
def test_arcface_net(ctx, states, precision):
    from terran_amd import lib
    _prec[0] = precision
    from oracle import nets
    sd = states('arcface')
    m = lib.Model(ctx, pack.pack_arcface(sd, precision))
    g = golden('nets_arcface.npz')
    crops = np.concatenate([g['crops'], np.random.default_rng(41).integers(0, 256, (3, 3, 112, 112), dtype=np.uint8)])
    m.forward_crops(crops)
    taps = {}
    emb = nets.arcface_forward(sd, torch.from_numpy(crops.astype(np.float32)), taps).numpy()
    _close(m.read('stem'), taps['stem'].numpy(), what='stem')
    for s in (1, 2, 3, 4):
        _close(m.read('stage%d' % s), taps['stage%d' % s].numpy(), what='stage%d' % s)
    out = m.read('embedding')[:, :, 0, 0]
    e = _close(out, emb, what='embedding')
    _close(out[:2], g['embeddings'], what='golden embedding')
    print('arcface', precision, 'max err', e, 'scale', np.abs(emb).max())


def test_retinaface_lanes_equal_the_serial_program(ctx, states, monkeypatch):
    """The context modules + heads of the stride-32 / 16 levels run on side streams (op lanes): every head must come out
    bit-identical to the program that keeps them in order on the main stream, call after call (fork / join per forward)."""
    from terran_amd import lib
    sd = states('retinaface')
    images = synth.frames(61, 3, 150, 203)
    fr = ctx.upload(images)
    lanes = lib.Model(ctx, pack.pack_retinaface(sd, 'f16x3'))
    monkeypatch.setenv('TERRAN_AMD_NO_DETECTOR_LANES', '1')
    serial = lib.Model(ctx, pack.pack_retinaface(sd, 'f16x3'))
    assert {(op['variant'] >> 17) & 3 for op in pack.pack_retinaface(sd, 'f16x3').ops} == {0}
    monkeypatch.delenv('TERRAN_AMD_NO_DETECTOR_LANES')
    assert {(op['variant'] >> 17) & 3 for op in pack.pack_retinaface(sd, 'f16x3').ops} == {0, 1, 2}
    serial.forward_frames(fr)
    want = {k: serial.read(k) for k in ('head32', 'head16', 'head8', 'ctx32_7x7', 'ctx16_3x3')}
    for _ in range(3):
        lanes.forward_frames(fr)
        for k, v in want.items():
            assert np.array_equal(lanes.read(k), v), k


def test_lane_sharing_a_tensor_with_the_main_stream_is_refused(ctx):
    """The loader checks what makes lanes safe: an op outside a lane may not read what the lane writes."""
    from terran_amd import lib
    rng = np.random.default_rng(5)
    P = pack.Program(pack.MODEL_OPENPOSE, 'f32')
    t0 = P.tensor(4, 1)
    P.input_tensor = t0
    t1, t2, t3 = P.tensor(32, 1), P.tensor(32, 1), P.tensor(32, 0)
    w = lambda co, ci, k: rng.normal(0, 0.1, (co, ci, k, k)).astype(np.float32)
    P.conv(t0, t1, w(32, 3, 3), np.zeros(32, np.float32))
    P.lane = 1
    P.conv(t1, t2, w(32, 32, 3), np.zeros(32, np.float32))
    P.lane = 0
    P.conv(t2, t3, w(32, 32, 1), np.zeros(32, np.float32))          # reads the lane's output on the main stream
    P.outputs = [t3]
    with pytest.raises(lib.TerranAmdError):
        lib.Model(ctx, P)


@pytest.mark.parametrize('shape', [(1, 27, 123), (1, 29, 125), (2, 57, 249), (1, 5, 7), (1, 56, 248), (1, 31, 126), (3, 16, 16)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_retinaface_front_tile_edges(ctx, states, shape):
    """rf_stem_kernel works on 14 x 62 output tiles from aligned dwords of the frame rows: maps of exactly one tile / one
    pixel more, every row misalignment (W * 3 mod 4 = 0..3), frames smaller than a tile; the stride-8 / 16 / 32 features
    (everything downstream of the front) against the oracle."""
    from terran_amd import lib
    _prec[0] = 'f32'
    from oracle import nets
    sd = states('retinaface')
    m = lib.Model(ctx, pack.pack_retinaface(sd, 'f32', fused=True))
    images = synth.frames(50 + shape[1], *shape)
    m.forward_frames(ctx.upload(images))
    x = torch.from_numpy(images.astype(np.float32)).permute(0, 3, 1, 2).flip(1).contiguous()
    taps = {}
    nets.retinaface_forward(sd, x, taps)
    for s_ in (8, 16, 32):
        _close(m.read('feat%d' % s_), taps['feat%d' % s_].numpy(), what='feat%d' % s_)


def test_arcface_single_half_mode_within_the_embedding_bar(ctx, states):
    """precision='f16': ONE f16 MFMA per product (operands rounded to 11 bits, f32 accumulate) -- a tolerance mode for the
    embedder only, which takes no discrete decision.  north_star's bar for embeddings is 1e-3 on the unit-norm vector;
    measured 3.6e-4 worst component with 2-byte half-float activations (TA_FMT_F16; the CPU emulation tests/probe_embed_precision.py predicts 3.0e-4; one bfloat16 per
    operand gives 2e-3).  The same packers map 'f16' to 'f16x3' for the detector and the pose network."""
    from terran_amd import ArcFace, lib
    from oracle import nets, pipeline
    sd = states('arcface')
    crops = np.random.default_rng(43).integers(0, 256, (12, 3, 112, 112), dtype=np.uint8)
    ref = nets.arcface_forward(sd, torch.from_numpy(crops.astype(np.float32))).numpy()
    ref /= np.linalg.norm(ref, axis=1, keepdims=True)
    m = lib.Model(ctx, pack.pack_arcface(sd, 'f16'))
    assert {op['prec'] for op in pack.pack_arcface(sd, 'f16').ops if op['type'] == pack.OP_CONV} == {4}
    m.forward_crops(crops)
    e = m.read('embedding')[:, :, 0, 0]
    e = e / np.linalg.norm(e, axis=1, keepdims=True)
    err = float(np.abs(e - ref).max())
    cos = float((1.0 - (e * ref).sum(1)).max())
    print('f16 embedder: max component error %.2e, max cosine distance to the oracle %.2e' % (err, cos))
    assert err <= 1e-3 and cos <= 1e-5
    assert err >= 1e-5                                   # ... and it really is the 11-bit arithmetic, not a fallback
    images = np.random.default_rng(44).integers(0, 256, (3, 112, 112, 3), dtype=np.uint8)
    got = ArcFace(device=0, state=sd, precision='f16').call(images)
    want = pipeline.arcface_call(sd, images)
    assert np.abs(got - want).max() <= 1e-3
    for packer, sdn in ((pack.pack_retinaface, 'retinaface'), (pack.pack_openpose, 'openpose')):
        precs = {op['prec'] for op in packer(states(sdn), 'f16').ops if op['type'] == pack.OP_CONV}
        assert 4 not in precs and 3 in precs


@pytest.mark.parametrize('fused', [True, False], ids=['fused', 'layerwise'])
@pytest.mark.parametrize('precision', PRECISIONS)
def test_retinaface_net(ctx, states, precision, fused):
    from terran_amd import lib
    _prec[0] = precision
    from oracle import nets
    sd = states('retinaface')
    m = lib.Model(ctx, pack.pack_retinaface(sd, precision, fused=fused))
    g = golden('nets_retinaface.npz')
    for images in (g['images'], synth.frames(42, 2, 75, 101), synth.frames(43, 1, 33, 250)):   # odd sizes: ceil strides + upsample crop
        fr = ctx.upload(images)
        m.forward_frames(fr)
        x = torch.from_numpy(images.astype(np.float32)).permute(0, 3, 1, 2).flip(1).contiguous()
        taps = {}
        outs = [o.numpy() for o in nets.retinaface_forward(sd, x, taps)]
        if not fused:                                               # the fused front never materialises the 8-channel maps
            _close(m.read('stem'), taps['stem'].numpy(), what='stem')
        for s in (8, 16, 32):
            _close(m.read('feat%d' % s), taps['feat%d' % s].numpy(), what='feat%d' % s)
            _close(m.read('p%d' % s), taps['p%d' % s].numpy(), what='p%d' % s)
            ctx_ref = taps['ctx%d' % s].numpy()
            _close(m.read('ctx%d_3x3' % s), ctx_ref[:, :32], what='ctx3x3')
            _close(m.read('ctx%d_5x5' % s), ctx_ref[:, 32:48], what='ctx5x5')
            _close(m.read('ctx%d_7x7' % s), ctx_ref[:, 48:], what='ctx7x7')
        for i, s in enumerate((32, 16, 8)):
            head = m.read('head%d' % s)               # raw logits | bbox | landmarks
            cls, bbox, lmk = head[:, 0:4], head[:, 4:12], head[:, 12:32]
            prob = outs[3 * i]
            ref_fg = prob[:, 2:4]
            mine_fg = 1.0 / (1.0 + np.exp(cls[:, 0:2].astype(np.float64) - cls[:, 2:4].astype(np.float64)))
            _close(mine_fg, ref_fg, what='fg prob s%d' % s)
            _close(bbox, outs[3 * i + 1], what='bbox s%d' % s)
            _close(lmk, outs[3 * i + 2], what='lmk s%d' % s)


@pytest.mark.parametrize('precision', PRECISIONS)
def test_worksize_nets_vs_reference_goldens(ctx, states, precision):
    """The device at the working sizes of BASELINE configs[4] against what the REFERENCE modules produced there
    (tests/golden/worksize_*.npz: one 184 x 327 OpenPose forward, one 416 x 739 RetinaFace head set) -- not against the
    oracle, which shares the product's layer tables."""
    from terran_amd import lib
    _prec[0] = precision
    g = golden('worksize_openpose.npz')
    n, h, w = (int(v) for v in g['shape'])
    m = lib.Model(ctx, pack.pack_openpose(states('openpose'), precision))
    m.forward_frames(ctx.upload(synth.frames(int(g['frames_seed']), n, h, w)))
    _close(m.read('pafs'), g['pafs'], what='golden pafs 184x327')
    _close(m.read('heatmaps'), g['heatmaps'], what='golden heatmaps 184x327')
    m.free()
    g = golden('worksize_retinaface.npz')
    n, h, w = (int(v) for v in g['shape'])
    m = lib.Model(ctx, pack.pack_retinaface(states('retinaface'), precision))
    m.forward_frames(ctx.upload(synth.frames(int(g['frames_seed']), n, h, w)))
    for i, s in enumerate((32, 16, 8)):
        head = m.read('head%d' % s)
        fg = 1.0 / (1.0 + np.exp(head[:, 0:2].astype(np.float64) - head[:, 2:4].astype(np.float64)))
        for j, mine in enumerate((fg, head[:, 4:12], head[:, 12:32])):
            k = 3 * i + j
            if k < 6:
                _close(mine, g['out%d' % k][:, 2:4] if j == 0 else g['out%d' % k], what='golden out%d' % k)
            else:
                ref4 = g['out%d_every4' % k]
                _close(mine[:, :, ::4, ::4], ref4[:, 2:4] if j == 0 else ref4, what='golden out%d (every 4th pixel)' % k)
                if j:
                    sums = mine.astype(np.float64).sum((0, 2, 3))
                    assert np.abs(sums - g['out%d_sum' % k]).max() <= NET_TOL[precision] * g['out%d_abs_sum' % k].max()
    m.free()


@pytest.mark.parametrize('precision', PRECISIONS + ['f16', 'f16x2'])
def test_wild_statistics_nets_vs_reference_goldens(ctx, states, precision):
    """The device on weights with trained-looking statistics against the REFERENCE modules run on the same weights
    (tests/golden/wild_*.npz).  'f16' = the opt-in single-half embedder: measured outside north_star's 1e-3 on these weights
    (DESIGN section 4), held to 5e-3 here; its detector / pose programs are f16x3's."""
    from terran_amd import lib
    _prec[0] = 'f16x3' if precision in ('f16', 'f16x2') else precision
    wt = {'f32': 3e-4, 'f16x3': 3e-4, 'bf16x3': 1e-3}[_prec[0]]      # ill-conditioned weights: measured 1.0e-4 (fg prob, stride 8) in the
                                                                     # exact-f32 mode and in f16x3 alike, ~5 x the benign nets' distance
    g = golden('wild_retinaface.npz')
    n, h, w = (int(v) for v in g['shape'])
    m = lib.Model(ctx, pack.pack_retinaface(states('wild_retinaface'), precision))
    m.forward_frames(ctx.upload(synth.frames(int(g['frames_seed']), n, h, w)))
    for i, s in enumerate((32, 16, 8)):
        head = m.read('head%d' % s)
        fg = 1.0 / (1.0 + np.exp(head[:, 0:2].astype(np.float64) - head[:, 2:4].astype(np.float64)))
        _close(fg, g['out%d' % (3 * i)][:, 2:4], tol=wt, what='wild fg prob s%d' % s)
        _close(head[:, 4:12], g['out%d' % (3 * i + 1)], tol=wt, what='wild bbox s%d' % s)
        _close(head[:, 12:32], g['out%d' % (3 * i + 2)], tol=wt, what='wild lmk s%d' % s)
    assert ctx.lib.ta_debug_range_check(ctx.h) == lib.OK
    m.free()
    g = golden('wild_openpose.npz')
    n, h, w = (int(v) for v in g['shape'])
    m = lib.Model(ctx, pack.pack_openpose(states('wild_openpose'), precision))
    m.forward_frames(ctx.upload(synth.frames(int(g['frames_seed']), n, h, w)))
    _close(m.read('pafs'), g['pafs'], tol=wt, what='wild golden pafs')
    _close(m.read('heatmaps'), g['heatmaps'], tol=wt, what='wild golden heatmaps')
    assert ctx.lib.ta_debug_range_check(ctx.h) == lib.OK
    m.free()
    g = golden('wild_arcface.npz')
    m = lib.Model(ctx, pack.pack_arcface(states('wild_arcface'), precision))
    m.forward_crops(g['crops'])
    out = m.read('embedding')[:, :, 0, 0]
    _close(out, g['embeddings'], tol=2e-2 if precision in ('f16', 'f16x2') else wt, what='wild golden embedding')
    unit = lambda e: e / np.sqrt((e.astype(np.float64) ** 2).sum(1, keepdims=True))
    err = float(np.abs(unit(out) - unit(g['embeddings'])).max())
    print('  wild arcface %s: unit embeddings max abs err %.2e' % (precision, err))
    assert err <= {'f32': 2e-5, 'f16x3': 2e-5, 'bf16x3': 2e-4, 'f16': 5e-3, 'f16x2': 1e-3}[precision]      # f16x2 (the default embedder): INSIDE north_star's bar on these weights too
    assert ctx.lib.ta_debug_range_check(ctx.h) == lib.OK
    m.free()

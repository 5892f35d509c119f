"""-m gpu: unit tests of the implicit-GEMM conv kernel through a two-op program
(uint8 frame -> 3x3 conv to C1 channels -> conv under test), against torch CPU conv2d."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from terran_amd import pack, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from terran_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


CASES = [
    # c1, cout, k, stride, halo_in, in_off, cin_used, out_off, out_total, act, res, out2
    dict(c1=128, cout=128, k=3),
    dict(c1=128, cout=512, k=1),
    dict(c1=128, cout=38, k=1, out_total=192, out_off=128, cout_p=40),
    dict(c1=512, cout=19, k=1, out_total=192, out_off=168, cout_p=20),
    dict(c1=128, cout=64, k=3, halo=3),
    dict(c1=192, cout=128, k=7, halo=3),
    dict(c1=192, cout=128, k=3, halo=3, in_off=0, cin_used=128),
    dict(c1=64, cout=64, k=3, stride=2),
    dict(c1=64, cout=128, k=1, stride=2, pad=0),
    dict(c1=64, cout=16, k=3, in_off=32, cin_used=16),
    dict(c1=8, cout=16, k=1),
    dict(c1=16, cout=8, k=3),
    dict(c1=64, cout=64, k=3, act=2, res=True, out2=True),
    dict(c1=32, cout=32, k=1, act=1, res=True),
    dict(c1=256, cout=256, k=3, n=3, h=14, w=14),
    dict(c1=256, cout=256, k=3, groups=2),
    dict(c1=256, cout=256, k=7, halo=3, groups=2, n=3, h=23, w=40),
    dict(c1=384, cout=256, k=1, groups=2, in_off=128, cin_used=256, act=1),
    # a per-channel affine of the INPUT in front of the zero-padded 3x3 conv, folded into weights + nine border-class biases
    # (ArcFace's BatchNorm-before-conv, arcface/model.py:12-14): split-role kernel (lean and generic drains), generic kernel,
    # maps of one row / one column (every pixel is a border pixel of two classes at once)
    dict(c1=128, cout=128, k=3, act=2, in_affine=True, n=3, h=14, w=14),
    dict(c1=64, cout=64, k=3, act=2, in_affine=True, n=2, h=7, w=9),
    dict(c1=256, cout=256, k=3, act=0, in_affine=True, n=2, h=1, w=5),
    dict(c1=32, cout=16, k=3, act=1, in_affine=True, h=6, w=1),
    dict(c1=16, cout=32, k=3, act=2, in_affine=True, h=5, w=5),
]


# float32 MFMA is an exact fmaf chain; bf16x3 drops the lo*lo term (~1e-5 of sum|x*w|); bf16 keeps 8 bits
TOLS = {'f32': 2e-4, 'f16x3': 2e-4, 'bf16x3': 6e-4, 'bf16': 6e-2}     # f16x3: 22-bit operands, float32-grade


@pytest.mark.parametrize('precision', ['f32', 'f16x3', 'bf16x3', 'bf16'])
@pytest.mark.parametrize('case', CASES, ids=lambda c: '-'.join('%s%s' % kv for kv in c.items()))
def test_conv(ctx, case, precision):
    from terran_amd import lib
    tol = TOLS[precision]
    rng = np.random.default_rng(7)
    c1, cout, k = case['c1'], case['cout'], case['k']
    stride = case.get('stride', 1)
    padv = case.get('pad', k // 2)
    halo = case.get('halo', max(padv, 0))
    in_off = case.get('in_off', 0)
    cin_used = case.get('cin_used', c1)
    out_off = case.get('out_off', 0)
    out_total = case.get('out_total', (cout + 3) // 4 * 4)
    act = case.get('act', 0)
    n, h, w = case.get('n', 2), case.get('h', 19), case.get('w', 23)

    P = pack.Program(pack.MODEL_OPENPOSE, precision)
    t0 = P.tensor(4, 1)
    P.input_tensor = t0
    t1 = P.tensor(c1, halo, name='mid')
    W1 = rng.normal(0, 0.3, (c1, 3, 3, 3)).astype(np.float32)
    b1 = rng.normal(0, 0.1, c1).astype(np.float32)
    P.conv(t0, t1, W1, b1, act=pack.ACT_RELU)
    groups = case.get('groups', 1)
    W2 = rng.normal(0, 1.0 / np.sqrt(cin_used // groups * k * k), (cout, cin_used // groups, k, k)).astype(np.float32)
    b2 = rng.normal(0, 0.1, cout).astype(np.float32)
    t2 = P.tensor(out_total, 0, name='out')
    kw = {}
    prelu = scale2 = shift2 = None
    if act == 2:
        prelu = rng.uniform(0.1, 0.4, cout).astype(np.float32)
        kw['prelu'] = prelu
    tres = -1
    if case.get('res'):
        # residual = a third tensor produced from the input by another conv (same spatial size as out)
        tres = P.tensor((cout + 3) // 4 * 4, 0, name='res')
        Wr = rng.normal(0, 0.3, (cout, 3, 3, 3)).astype(np.float32)
        br = rng.normal(0, 0.1, cout).astype(np.float32)
        P.conv(t0, tres, Wr, br, stride=stride, pad=1)
        kw['res'] = tres
    if case.get('out2'):
        t3 = P.tensor((cout + 3) // 4 * 4, 1, name='out2')
        scale2 = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        shift2 = rng.normal(0, 0.2, cout).astype(np.float32)
        kw.update(out2=t3, scale2=scale2, shift2=shift2)
    aff = None
    if case.get('in_affine'):
        aff = (rng.uniform(0.5, 1.5, cin_used).astype(np.float32), rng.normal(0, 0.5, cin_used).astype(np.float32))
        kw['in_affine'] = aff
    if groups > 1:
        kw['groups'] = groups
    else:
        kw['cin_p'] = cin_used
    P.conv(t1, t2, W2, b2, stride=stride, pad=padv, act=act, in_ch_off=in_off,
           out_ch_off=out_off, cout_p=case.get('cout_p'), **kw)
    P.outputs = [t2]
    m = lib.Model(ctx, P)

    images = synth.frames(3, n, h, w)
    m.forward_frames(ctx.upload(images))
    x = torch.from_numpy(np.transpose(images, (0, 3, 1, 2)).astype(np.float32) / 255.0 - 0.5)
    mid = F.relu(F.conv2d(x, torch.from_numpy(W1), torch.from_numpy(b1), padding=1))
    np.testing.assert_allclose(m.read('mid'), mid.numpy(), rtol=tol, atol=tol)
    xin = mid[:, in_off:in_off + cin_used]
    if aff is not None:                              # the reference order: affine first, THEN zero padding
        xin = xin * torch.from_numpy(aff[0])[None, :, None, None] + torch.from_numpy(aff[1])[None, :, None, None]
    y = F.conv2d(xin, torch.from_numpy(W2), torch.from_numpy(b2), stride=stride, padding=padv, groups=groups)
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = F.prelu(y, torch.from_numpy(prelu))
    if case.get('res'):
        r = F.conv2d(x, torch.from_numpy(Wr), torch.from_numpy(br), stride=stride, padding=1)
        np.testing.assert_allclose(m.read('res')[:, :cout], r.numpy(), rtol=tol, atol=tol)
        y = y + r
    got = m.read('out')[:, out_off:out_off + cout]
    np.testing.assert_allclose(got, y.numpy(), rtol=tol, atol=2 * tol)
    full = m.read('out')
    mask = np.ones(out_total, bool)
    mask[out_off:out_off + cout] = False
    assert np.all(full[:, mask] == 0.0), 'conv wrote outside its channel slice'
    if case.get('out2'):
        z = y * torch.from_numpy(scale2)[None, :, None, None] + torch.from_numpy(shift2)[None, :, None, None]
        np.testing.assert_allclose(m.read('out2')[:, :cout], z.numpy(), rtol=tol, atol=2 * tol)

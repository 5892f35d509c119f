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
TOLS = {'f32': 2e-4, 'f16x3': 2e-4, 'bf16x3': 6e-4, 'bf16': 6e-2, 'f16': 6e-3, 'f16x2': 4e-3}     # f16x3: 22-bit operands, float32-grade; f16: 11 bits


@pytest.mark.parametrize('precision', ['f32', 'f16x3', 'bf16x3', 'bf16', 'f16', 'f16x2'])      # f16x2: f16x3's tensors, activations enter as their hi half (11 bits)
@pytest.mark.parametrize('case', CASES, ids=lambda c: '-'.join('%s%s' % kv for kv in c.items()))
def test_conv(ctx, case, precision):
    from terran_amd import lib
    if precision == 'f16' and case.get('groups', 1) > 1:
        pytest.skip('the single-half mode has no grouped convs (its tensors are whole 64-channel blocks)')
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


F16_CASES = [
    dict(c1=128, cout=128, k=3),                                  # half-float tensor in, half-float tensor out
    dict(c1=64, cout=128, k=1, stride=2, pad=0),                  # ONE K slab of 64 channels (ArcFace's first shortcuts)
    dict(c1=64, cout=64, k=3, stride=2, act=2, res=True),
    dict(c1=128, cout=38, k=1, out_total=192, out_off=128, cout_p=40),   # 4-channel groups into a slice of a half-float tensor
    dict(c1=256, cout=256, k=3, n=3, h=14, w=14, act=2, in_affine=True),
    dict(c1=192, cout=128, k=7, halo=3),
    dict(c1=512, cout=512, k=3, n=2, h=7, w=7, k_split=2),       # fixed K split: partial sums + splitk_reduce into the half tensor
]


@pytest.mark.parametrize('case', F16_CASES, ids=lambda c: '-'.join('%s%s' % kv for kv in c.items()))
def test_conv_half_float_tensors_against_rounded_operands(ctx, case):
    """precision='f16' on TA_FMT_F16 tensors (2 bytes per element, K slabs of 64 channels, one f16 MFMA per product): the
    products of half floats are exact in float32, so against a reference whose operands are rounded to half floats where
    the kernels round them (weights x 2^s at pack time, activations when a tensor is stored) only the summation order and
    the final rounding of the stored tensor are left: 1e-3 of the output scale here, against 6e-3 for the unrounded
    reference of test_conv."""
    from terran_amd import lib
    rng = np.random.default_rng(11)
    c1, cout, k = case['c1'], case['cout'], case['k']
    stride, padv = case.get('stride', 1), case.get('pad', k // 2)
    halo = case.get('halo', max(padv, 0))
    out_off, out_total, act = case.get('out_off', 0), case.get('out_total', (cout + 3) // 4 * 4), case.get('act', 0)
    n, h, w = case.get('n', 2), case.get('h', 19), case.get('w', 23)
    r16 = lambda t: t.half().float()
    P = pack.Program(pack.MODEL_OPENPOSE, 'f16')
    t0 = P.tensor(4, 1)
    P.input_tensor = t0
    t1 = P.tensor(c1, halo, name='mid')
    W1 = rng.normal(0, 0.3, (c1, 3, 3, 3)).astype(np.float32)
    b1 = rng.normal(0, 0.1, c1).astype(np.float32)
    P.conv(t0, t1, W1, b1, act=pack.ACT_RELU)
    W2 = rng.normal(0, 1.0 / np.sqrt(c1 * k * k), (cout, c1, k, k)).astype(np.float32)
    b2 = rng.normal(0, 0.1, cout).astype(np.float32)
    t2 = P.tensor(out_total, 0, name='out')
    kw = {}
    prelu = None
    if act == 2:
        prelu = rng.uniform(0.1, 0.4, cout).astype(np.float32)
        kw['prelu'] = prelu
    if case.get('res'):
        tres = P.tensor(cout, 0, name='res')
        Wr = rng.normal(0, 0.3, (cout, 3, 3, 3)).astype(np.float32)
        br = rng.normal(0, 0.1, cout).astype(np.float32)
        P.conv(t0, tres, Wr, br, stride=stride, pad=1)
        kw['res'] = tres
    aff = None
    if case.get('in_affine'):
        aff = (rng.uniform(0.5, 1.5, c1).astype(np.float32), rng.normal(0, 0.5, c1).astype(np.float32))
        kw['in_affine'] = aff
    P.conv(t1, t2, W2, b2, stride=stride, pad=padv, act=act, out_ch_off=out_off, cout_p=case.get('cout_p'),
           k_split=case.get('k_split', 0), **kw)
    P.outputs = [t2]
    fmts = P.tensor_formats()
    assert fmts[t1] == pack.FMT_F16 and (fmts[t2] == pack.FMT_F16) == (out_total % 64 == 0)
    m = lib.Model(ctx, P)
    assert P.ops[-1]['n_slabs'] == k * k * c1 // 64
    images = synth.frames(3, n, h, w)
    m.forward_frames(ctx.upload(images))
    x = torch.from_numpy(np.transpose(images, (0, 3, 1, 2)).astype(np.float32) / 255.0 - 0.5)

    def wq(W):                                   # weights as packed: x 2^s (largest in [2^13, 2^14)), rounded to half, scaled back
        s = 13 - int(np.floor(np.log2(np.abs(W).max())))
        return torch.from_numpy(np.ldexp(np.ldexp(W.astype(np.float32), s).astype(np.float16).astype(np.float32), -s))
    mid = r16(F.relu(F.conv2d(r16(x), wq(W1), torch.from_numpy(b1), padding=1)))
    got_mid = m.read('mid')
    scale = float(mid.abs().max())
    assert np.abs(got_mid - mid.numpy()).max() <= 1e-3 * scale
    xin = mid
    Wq = W2
    if aff is not None:                          # folded into the weights (rounded AFTER the fold) + border-class biases
        Wf, b16 = pack.fold_input_affine(W2, b2, *aff)
        y = F.conv2d(xin, wq(Wf.astype(np.float32)), None, stride=stride, padding=padv)
        yy = torch.from_numpy(np.zeros((), np.float32)) + y
        cls = np.zeros((h, w), int)
        for yy_ in range(h):
            for xx_ in range(w):
                cy = 3 if h == 1 else (0 if yy_ == 0 else (2 if yy_ == h - 1 else 1))
                cx = 3 if w == 1 else (0 if xx_ == 0 else (2 if xx_ == w - 1 else 1))
                cls[yy_, xx_] = cy * 4 + cx
        y = yy + torch.from_numpy(b16[cls].astype(np.float32)).permute(2, 0, 1)[None]
    else:
        y = F.conv2d(xin, wq(Wq), torch.from_numpy(b2), stride=stride, padding=padv)
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = F.prelu(y, torch.from_numpy(prelu))
    if case.get('res'):
        r = F.conv2d(r16(x), wq(Wr), torch.from_numpy(br), stride=stride, padding=1)
        if fmts[tres] == pack.FMT_F16:
            r = r16(r)
        y = y + r
    if fmts[t2] == pack.FMT_F16:
        y = r16(y)
    got = m.read('out')[:, out_off:out_off + cout]
    oscale = max(1.0, float(y.abs().max()))
    err = float(np.abs(got - y.numpy()).max())
    print('half-float conv %s: max err %.2e (scale %.2f)' % (case, err, oscale))
    assert err <= 1e-3 * oscale
    full = m.read('out')
    mask = np.ones(out_total, bool)
    mask[out_off:out_off + cout] = False
    assert np.all(full[:, mask] == 0.0), 'conv wrote outside its channel slice'

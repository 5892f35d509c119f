"""-m gpu: every implicit-GEMM kernel variant, pinned per layer (ta_op_desc.variant), on layers of the sizes bench.py
and the BASELINE configs run (C4: 16x46x82 maps, C5: 32x23x40 maps, C3: 256 crops), against torch-CPU conv2d fed with the
very activations the kernel consumed (the producing layer's output read back through the debug tap).

`ctx.conv_counts()` proves which kernel ran.  Errors are reported relative to max|reference| and bounded at ~10x what
the kernels measure: f32 = exact-f32 MFMA (only the summation order differs from torch), bf16x3 drops the lo*lo term.
DESIGN.md section 4 maps each variant to the test ids of this file.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from terran_amd import pack, synth

pytestmark = pytest.mark.gpu

# measured (MI355X, this file): f32 <= 3.7e-6 (K = 9408 products per output: summation order), bf16x3 <= 5.6e-6 of max|ref|
TOL = {'f32': 2e-5, 'f16x3': 2e-5, 'bf16x3': 5e-5}


@pytest.fixture(scope='module')
def ctx():
    from terran_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


LAYERS = {
    # OpenPose stage conv Mconv1 at C4 size: 7x7, 185 (padded 192) -> 128, 16 x 46 x 82 maps        (model.py:58-95)
    'pose7x7_192to128_c4': dict(n=16, h=46, w=82, c1=192, cout=128, k=7),
    # the same layer at the 1080p size of bench.py (32 x 23 x 40 maps)
    'pose7x7_192to128_c5': dict(n=32, h=23, w=40, c1=192, cout=128, k=7),
    # PAF | heat-map branch pair as ONE grouped conv (pack.pack_openpose), C5 and C4 sizes
    'pose7x7_grouped_c5': dict(n=32, h=23, w=40, c1=256, cout=256, k=7, groups=2),
    'pose7x7_grouped_c4': dict(n=16, h=46, w=82, c1=256, cout=256, k=7, groups=2),
    # ArcFace stage 3 body conv at C3 size: 3x3 256 -> 256, 256 crops x 14 x 14                      (arcface/model.py:11-35)
    'arc3x3_256_c3': dict(n=256, h=14, w=14, c1=256, cout=256, k=3),
    # ArcFace unit-closing conv: stride 2, PReLU-free, residual + next unit's BatchNorm as second output
    'arc3x3_s2_res_out2': dict(n=64, h=56, w=56, c1=128, cout=128, k=3, stride=2, res=True, out2=True),
    'arc3x3_prelu': dict(n=64, h=28, w=28, c1=128, cout=128, k=3, act=2),
    # OpenPose VGG conv1_2 at the 1080p size: 3x3 64 -> 64 on 184 x 327 maps                           (model.py:41-57)
    'vgg3x3_64_c5': dict(n=8, h=184, w=327, c1=64, cout=64, k=3, act=1),
    # conv + ReLU + 2x2 max-pool in one launch (pack.py `pool=True`; conv1_2 / conv2_2 / conv3_4 of the VGG front), odd sizes
    'vgg3x3_64_pool_c5': dict(n=8, h=184, w=327, c1=64, cout=64, k=3, act=1, pool=True),
    'vgg3x3_128_pool_c5': dict(n=8, h=92, w=163, c1=128, cout=128, k=3, act=1, pool=True),
    'vgg3x3_256_pool_c5': dict(n=32, h=46, w=81, c1=256, cout=256, k=3, act=1, pool=True),
    # small / ragged shapes through the specialised epilogues (tile tails, maps narrower than a drain pass, tiny pooled maps)
    'small_pool': dict(n=3, h=7, w=10, c1=64, cout=64, k=3, act=1, pool=True),
    'small_pool_128': dict(n=5, h=9, w=5, c1=128, cout=128, k=3, act=1, pool=True),
    'small_res_out2': dict(n=3, h=5, w=3, c1=128, cout=128, k=3, res=True, out2=True),
    'small_prelu_256': dict(n=2, h=3, w=2, c1=256, cout=256, k=3, act=2),
    'small_plain_1x1': dict(n=1, h=1, w=1, c1=256, cout=128, k=1),
    # OpenPose stage output conv: 1x1 128 -> 38 into a channel slice of the 192-channel concat tensor
    'pose1x1_to_slice': dict(n=16, h=46, w=82, c1=128, cout=38, k=1, out_total=192, out_off=128, cout_p=40),
}

CASES = [
    # (layer, variant, mid tensor pinned to float32?)
    ('pose7x7_192to128_c4', 'split_2x2', False), ('pose7x7_192to128_c4', 'split_2x4', False),
    ('pose7x7_192to128_c4', 'split_2x2_p8', False), ('pose7x7_192to128_c4', 'split_1x4', False),
    ('pose7x7_192to128_c4', 'pipe64', False), ('pose7x7_192to128_c4', 'pipe64', True),
    ('pose7x7_192to128_c4', 'pipe128', True), ('pose7x7_192to128_c4', 'generic', True),
    ('pose7x7_192to128_c5', 'split_2x2', False), ('pose7x7_192to128_c5', 'split_2x4', False),
    ('pose7x7_192to128_c5', 'auto', False),
    ('pose7x7_grouped_c5', 'split_2x4', False), ('pose7x7_grouped_c5', 'split_2x2', False),
    ('pose7x7_grouped_c5', 'split_1x4', False), ('pose7x7_grouped_c5', 'auto', False),
    ('pose7x7_grouped_c4', 'split_2x4', False), ('pose7x7_grouped_c4', 'auto', False),
    ('arc3x3_256_c3', 'split_2x2', False), ('arc3x3_256_c3', 'split_2x4', False), ('arc3x3_256_c3', 'pipe64', False),
    ('arc3x3_256_c3', 'auto', False),
    ('arc3x3_s2_res_out2', 'split_2x2', False), ('arc3x3_s2_res_out2', 'split_2x4', False),
    ('arc3x3_s2_res_out2', 'pipe64', True), ('arc3x3_s2_res_out2', 'generic', True),
    ('arc3x3_prelu', 'split_2x4', False), ('arc3x3_prelu', 'auto', False),
    ('vgg3x3_64_c5', 'split_1x4', False), ('vgg3x3_64_c5', 'pipe64', False), ('vgg3x3_64_c5', 'auto', False),
    ('pose1x1_to_slice', 'generic', True), ('pose1x1_to_slice', 'auto', False),
    ('vgg3x3_64_pool_c5', 'split_1x4', False), ('vgg3x3_64_pool_c5', 'auto', False),
    ('vgg3x3_128_pool_c5', 'split_2x2', False), ('vgg3x3_128_pool_c5', 'split_2x4', False), ('vgg3x3_128_pool_c5', 'split_1x4', False),
    ('vgg3x3_128_pool_c5', 'auto', False),
    ('vgg3x3_256_pool_c5', 'split_2x4', False), ('vgg3x3_256_pool_c5', 'split_2x2_p8', False), ('vgg3x3_256_pool_c5', 'auto', False),
    ('small_pool', 'split_1x4', False), ('small_pool', 'auto', False), ('small_pool_128', 'split_2x2', False), ('small_pool_128', 'split_2x4', False),
    ('small_res_out2', 'split_2x2', False), ('small_res_out2', 'split_2x4', False), ('small_prelu_256', 'split_2x2', False),
    ('small_prelu_256', 'split_2x4', False), ('small_plain_1x1', 'split_2x2', False),
    # the 2-stage, two-workgroups-per-CU variants (short-K layers): same tiles, K order and epilogues
    ('vgg3x3_64_c5', 'split_1x4_w2', False), ('vgg3x3_64_pool_c5', 'split_1x4_w2', False), ('small_pool', 'split_1x4_w2', False),
    ('pose7x7_grouped_c5', 'split_1x4_w2', False), ('arc3x3_256_c3', 'split_2x2_w2', False), ('arc3x3_s2_res_out2', 'split_2x2_w2', False),
    ('vgg3x3_128_pool_c5', 'split_2x2_w2', False), ('small_res_out2', 'split_2x2_w2', False), ('small_plain_1x1', 'split_2x2_w2', False),
]

_ref_cache = {}


def _weights(L, rng):
    c1, cout, k, groups = L['c1'], L['cout'], L['k'], L.get('groups', 1)
    W1 = rng.normal(0, 0.3, (c1, 3, 3, 3)).astype(np.float32)
    b1 = rng.normal(0, 0.1, c1).astype(np.float32)
    W2 = rng.normal(0, 1.0 / np.sqrt(c1 // groups * k * k), (cout, c1 // groups, k, k)).astype(np.float32)
    b2 = rng.normal(0, 0.1, cout).astype(np.float32)
    return W1, b1, W2, b2


@pytest.mark.parametrize('precision', ['f32', 'f16x3', 'bf16x3'])
@pytest.mark.parametrize('layer,variant,mid_f32', CASES, ids=['%s-%s%s' % (a, b, '-f32in' if c else '') for a, b, c in CASES])
def test_conv_variant_at_bench_size(ctx, layer, variant, mid_f32, precision):
    _run_case(ctx, layer, variant, mid_f32, precision, split_io=False)


# The same layers with output / shortcut / second output kept in the pre-split bf16 format the networks use between
# convs: this is what routes the split-role kernels through their compile-time specialised epilogue (`lean_epilogue`
# in the counts).  Reading a split tensor back returns hi + lo: 16 mantissa bits, so the bound is 2^-16 wider.
SPLIT_CASES = [(layer, v) for layer, v, f in CASES
               if v.startswith('split') and not f and LAYERS[layer]['cout'] % 32 == 0 and 'out_total' not in LAYERS[layer]]


@pytest.mark.parametrize('precision', ['f16x3', 'bf16x3'])
@pytest.mark.parametrize('layer,variant', SPLIT_CASES, ids=['%s-%s' % c for c in SPLIT_CASES])
def test_conv_variant_split_format_tensors(ctx, layer, variant, precision):
    _run_case(ctx, layer, variant, False, precision, split_io=True)


def _run_case(ctx, layer, variant, mid_f32, precision, split_io):
    from terran_amd import lib
    L = LAYERS[layer]
    rng = np.random.default_rng(11)
    n, h, w, c1, cout, k = L['n'], L['h'], L['w'], L['c1'], L['cout'], L['k']
    stride, groups, act = L.get('stride', 1), L.get('groups', 1), L.get('act', 0)
    out_off, out_total = L.get('out_off', 0), L.get('out_total', cout)
    W1, b1, W2, b2 = _weights(L, rng)
    P = pack.Program(pack.MODEL_OPENPOSE, precision)
    t0 = P.tensor(4, 1)
    P.input_tensor = t0
    t1 = P.tensor(c1, k // 2, name='mid', f32=mid_f32)
    P.conv(t0, t1, W1, b1, act=pack.ACT_RELU)
    t2 = P.tensor(out_total, 0, name='out', f32=not split_io)
    kw = dict(variant=lib.CONV_VARIANTS[variant], groups=groups)
    prelu = scale2 = shift2 = Wr = br = None
    if act == 2:
        prelu = rng.uniform(0.1, 0.4, cout).astype(np.float32)
        kw['prelu'] = prelu
    if L.get('res'):
        tres = P.tensor(cout, 0, name='res', f32=not split_io)
        Wr = rng.normal(0, 0.3, (cout, 3, 3, 3)).astype(np.float32)
        br = rng.normal(0, 0.1, cout).astype(np.float32)
        P.conv(t0, tres, Wr, br, stride=stride, pad=1)
        kw['res'] = tres
    if L.get('out2'):
        t3 = P.tensor(cout, 1, name='out2', f32=not split_io)
        scale2 = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        shift2 = rng.normal(0, 0.2, cout).astype(np.float32)
        kw.update(out2=t3, scale2=scale2, shift2=shift2)
    if L.get('pool'):
        kw['pool'] = True
    P.conv(t1, t2, W2, b2, stride=stride, act=act, out_ch_off=out_off, cout_p=L.get('cout_p'), **kw)
    P.outputs = [t2]
    m = lib.Model(ctx, P)
    images = synth.frames(5, n, h, w)
    fr = ctx.upload(images)
    ctx.conv_counts(reset=True)
    m.forward_frames(fr)
    counts = ctx.conv_counts()
    if variant != 'auto':                                             # the pinned kernel is the one that ran
        helpers = 1 + (1 if L.get('res') else 0)                      # the 3 -> c1 stem (and the residual's producer) are `generic`
        assert counts.get(variant, 0) == (1 + helpers if variant == 'generic' else 1), counts
    # the specialised epilogue (conv_drain_fast) runs exactly when the tensors are in the split format and the epilogue is
    # one of the five it is compiled for; float32 tensors and the other kernels take the generic drain
    lean = (split_io and (act == 1 if L.get('pool') else ((not L.get('res') and not L.get('out2')) or
                                                          (L.get('res') and L.get('out2') and act == 0))))
    assert counts.get('lean_epilogue', 0) == (1 if lean else 0), (counts, L)
    mid = torch.from_numpy(m.read('mid'))                             # exactly what the conv under test consumed
    key = (layer, precision, mid_f32, split_io)
    if key not in _ref_cache:
        y = F.conv2d(mid, torch.from_numpy(W2), torch.from_numpy(b2), stride=stride, padding=k // 2, groups=groups)
        if act == 1:
            y = F.relu(y)
        elif act == 2:
            y = F.prelu(y, torch.from_numpy(prelu))
        if L.get('res'):
            y = y + torch.from_numpy(m.read('res'))
        if L.get('pool'):
            y = F.max_pool2d(y, 2, 2, 0)                               # floor: the odd last row / column drops out
        _ref_cache[key] = (mid.numpy().copy(), y.numpy())
    mid_ref, y = _ref_cache[key]
    assert np.array_equal(mid.numpy(), mid_ref)                       # same input as the cached reference saw
    scale = float(np.abs(y).max())
    got = m.read('out')
    err = float(np.abs(got[:, out_off:out_off + cout] - y).max()) / scale
    print('%s %s %s: kernels %s, max err %.2e of max|ref|' % (layer, variant, precision, counts, err))
    tol = TOL[precision] + ((2.0 ** -16 if precision == 'bf16x3' else 2.0 ** -21) if split_io else 0.0)   # hi + lo: 16 / 22 bits
    assert err <= tol, err
    if out_total != cout:
        mask = np.ones(out_total, bool)
        mask[out_off:out_off + cout] = False
        assert np.all(got[:, mask] == 0.0), 'conv wrote outside its channel slice'
    if L.get('out2'):
        z = y * scale2[None, :, None, None] + shift2[None, :, None, None]
        assert float(np.abs(m.read('out2') - z).max()) / max(scale, float(np.abs(z).max())) <= tol
    m.free()
    fr.free()


@pytest.mark.parametrize('precision', ['f32', 'f16x3', 'bf16x3'])
@pytest.mark.parametrize('variant', ['auto', 'split_2x2', 'split_1x4', 'pipe64'])
def test_fc_25088_to_512_at_c3_size(ctx, variant, precision):
    """ArcFace's Flatten + Linear 25088 -> 512 (arcface/model.py:79-85) as the 1x1 conv over the (N,1,1,25088) view, 256
    crops: the K-split path (32 fixed K ranges + ordered reduction) under `auto` / split variants, one pass under pipe64."""
    from terran_amd import lib
    rng = np.random.default_rng(12)
    n = 256
    P = pack.Program(pack.MODEL_OPENPOSE, precision)
    t0 = P.tensor(4, 1)
    P.input_tensor = t0
    Z = P.tensor(512, 0, name='z')
    W1 = rng.normal(0, 0.3, (512, 3, 3, 3)).astype(np.float32)
    b1 = rng.normal(0, 0.1, 512).astype(np.float32)
    P.conv(t0, Z, W1, b1, act=pack.ACT_RELU)
    A = P.tensor(7 * 7 * 512, 0, alias_of=Z)
    Wl = rng.normal(0, 1.0 / np.sqrt(25088), (512, 25088)).astype(np.float32)
    bl = rng.normal(0, 0.1, 512).astype(np.float32)
    f = np.arange(7 * 7 * 512)
    ch_pos = (f % 49) * 512 + f // 49                               # (C,H,W) flatten order -> NHWC position
    E = P.tensor(512, 0, name='emb', f32=True)
    P.conv(A, E, Wl.reshape(512, 25088, 1, 1), bl, ch_pos=ch_pos, pad=0, variant=lib.CONV_VARIANTS[variant])
    P.outputs = [E]
    m = lib.Model(ctx, P)
    fr = ctx.upload(synth.frames(6, n, 7, 7))
    ctx.conv_counts(reset=True)
    m.forward_frames(fr)
    counts = ctx.conv_counts()
    z = m.read('z')                                                  # (n,512,7,7)
    ref = z.reshape(n, -1).astype(np.float64) @ Wl.astype(np.float64).T + bl
    got = m.read('emb')[:, :, 0, 0]
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    print('fc %s %s: kernels %s, max err %.2e of max|ref|' % (variant, precision, counts, err))
    assert err <= TOL[precision], err
    if variant != 'auto':
        assert counts.get(variant, 0) == 1, counts
    # batch composition must not change a single bit (fixed K ranges): the first 100 crops alone
    fr2 = ctx.upload(synth.frames(6, n, 7, 7)[:100])
    m.forward_frames(fr2)
    assert np.array_equal(m.read('emb')[:, :, 0, 0], got[:100])
    m.free()


def test_pinned_variant_that_cannot_run_the_layer_is_an_error(ctx):
    from terran_amd import lib
    rng = np.random.default_rng(1)
    P = pack.Program(pack.MODEL_OPENPOSE, 'f32')
    t0 = P.tensor(4, 1)
    P.input_tensor = t0
    t1 = P.tensor(16, 0, name='out')
    P.conv(t0, t1, rng.normal(0, 0.3, (16, 3, 3, 3)).astype(np.float32), np.zeros(16, np.float32),
           variant=lib.CONV_VARIANTS['split_2x4'])                  # Cin = 4: only the table-driven kernel can
    P.outputs = [t1]
    m = lib.Model(ctx, P)
    with pytest.raises(lib.TerranAmdError) as e:
        m.forward_frames(ctx.upload(synth.frames(1, 1, 16, 16)))
    assert e.value.code == lib.E_INVALID and 'variant' in str(e.value)


# ---- window-resident pixel operand (conv_igemm_win): same tiles, same K order, same epilogue as the streaming split-role kernel ----
WIN_LAYERS = {
    # ArcFace stage 3 / stage 2 / stage 1 bodies at 64 crops; tiles cross image boundaries (196 / 784 / 3136 pixels per image)
    'arc14_256': dict(n=64, h=14, w=14, c1=256, cout=256, k=3, act=2),
    'arc28_128': dict(n=64, h=28, w=28, c1=128, cout=128, k=3, act=2),
    'arc56_64': dict(n=16, h=56, w=56, c1=64, cout=64, k=3, act=2),
    'arc14_res': dict(n=37, h=14, w=14, c1=256, cout=256, k=3, res=True),
    # ragged: maps narrower than a tile row run, several images per tile, a last tile with a handful of pixels
    'tiny_5x3': dict(n=41, h=5, w=3, c1=128, cout=128, k=3, act=1),
    'tiny_2x7': dict(n=29, h=2, w=7, c1=128, cout=128, k=3, act=1),
    # 5 x 5 (halo 2) and a grouped 3 x 3
    'k5_16x20': dict(n=5, h=16, w=20, c1=128, cout=128, k=5, act=1),
    'k3_grouped': dict(n=4, h=9, w=11, c1=256, cout=256, k=3, groups=2, act=1),
    # VGG-like 3 x 3 at the 1080p pose size (23 x 40): the widest map the 128 x 256 patch capacity takes
    'vgg23x40_256': dict(n=8, h=23, w=40, c1=256, cout=256, k=3, act=1),
}
WIN_CASES = [('arc14_256', '2x2'), ('arc14_256', '2x4'), ('arc28_128', '2x2'), ('arc28_128', '2x4'), ('arc56_64', '1x4'), ('arc14_res', '2x2'),
             ('arc14_res', '2x4'), ('tiny_5x3', '2x2'), ('tiny_2x7', '2x2'), ('k5_16x20', '2x2'), ('k3_grouped', '2x2'), ('k3_grouped', '2x4'),
             ('vgg23x40_256', '2x2'), ('vgg23x40_256', '2x4')]


@pytest.mark.parametrize('precision', ['f16x3', 'f16x2'])
@pytest.mark.parametrize('layer,tile', WIN_CASES, ids=['%s-%s' % c for c in WIN_CASES])
def test_window_kernel_equals_streaming_kernel(ctx, layer, tile, precision):
    """conv_igemm_win keeps the pixel operand of a channel block resident in LDS instead of streaming it per filter tap; tile
    mapping, K order, MFMA order and epilogue are conv_igemm_split's, so every output BIT must equal that kernel's (which the
    tests above hold against torch).  Also: the automatic choice takes the window kernel wherever it is eligible."""
    from terran_amd import lib
    L = WIN_LAYERS[layer]
    rng = np.random.default_rng(23)
    n, h, w, c1, cout, k = L['n'], L['h'], L['w'], L['c1'], L['cout'], L['k']
    W1, b1, W2, b2 = _weights(L, rng)
    prelu = rng.uniform(0.1, 0.4, cout).astype(np.float32)
    Wr, br = rng.normal(0, 0.3, (cout, 3, 3, 3)).astype(np.float32), rng.normal(0, 0.1, cout).astype(np.float32)
    fr = ctx.upload(synth.frames(6, n, h, w))
    outs = {}
    for variant in ('split_' + tile, 'win_' + tile, 'auto'):
        P = pack.Program(pack.MODEL_OPENPOSE, precision)
        t0 = P.tensor(4, 1)
        P.input_tensor = t0
        t1 = P.tensor(c1, k // 2, name='mid')
        P.conv(t0, t1, W1, b1, act=pack.ACT_RELU)
        t2 = P.tensor(cout, 1, name='out')                       # split format (a conv reads it), halo 1
        kw = dict(variant=lib.CONV_VARIANTS[variant], groups=L.get('groups', 1), act=L.get('act', 0))
        if L.get('act') == 2:
            kw['prelu'] = prelu
        if L.get('res'):
            tres = P.tensor(cout, 0, name='res')
            P.conv(t0, tres, Wr, br, pad=1)
            kw['res'] = tres
        P.conv(t1, t2, W2, b2, **kw)
        t3 = P.tensor(32, 0, name='sink', f32=True)              # keeps `out` in the split format
        P.conv(t2, t3, rng.normal(0, 0.05, (32, cout, 3, 3)).astype(np.float32), np.zeros(32, np.float32))
        P.outputs = [t3]
        m = lib.Model(ctx, P)
        ctx.conv_counts(reset=True)
        m.forward_frames(fr)
        counts = ctx.conv_counts()
        assert ctx.lib.ta_debug_range_check(ctx.h) == lib.OK
        if variant != 'auto':
            assert counts.get(variant, 0) >= 1, (variant, counts)
        else:
            assert any(kname.startswith('win_') for kname in counts), counts      # eligible -> chosen
        outs[variant] = (m.read('out'), counts)
        m.free()
    fr.free()
    ref = outs['split_' + tile][0]
    assert np.abs(ref).max() > 0
    assert np.array_equal(outs['win_' + tile][0], ref), (layer, tile, float(np.abs(outs['win_' + tile][0] - ref).max()))
    assert np.array_equal(outs['auto'][0], ref)


# ---- the detector's [depthwise 3x3 -> 1x1] block: rf_dwpw_kernel (written for this block) against conv_dwpw (generic tiles) ----------
def _dwpw_block_program(C, cout, stride, split_out):
    """frames -> conv 3x3 (4 -> C, exact f32) -> [dw3x3 (stride) -> 1x1 C -> cout] (f16x3) [-> 1x1 conv (f16x3): the block's output is then
    stored pre-split] -> float32 out."""
    rng = np.random.default_rng(1000 * C + 10 * cout + stride)
    P = pack.Program(pack.MODEL_OPENPOSE, 'f16x3')
    t0 = P.tensor(4, 1)
    P.input_tensor = t0
    P.input_stats = (np.array([-0.05] * 3 + [0.0]), np.array([0.08] * 3 + [0.0]))
    t1 = P.tensor(C, 1)
    P.conv(t0, t1, rng.normal(0, 0.3, (C, 3, 3, 3)).astype(np.float32), rng.normal(0, 0.1, C).astype(np.float32), act=pack.ACT_RELU, precision='f32')
    t2 = P.tensor(cout, 0, name='block', f32=not split_out)
    P.dwpw(t1, t2, rng.normal(0, 0.3, (C, 1, 3, 3)).astype(np.float32), rng.normal(0, 0.1, C).astype(np.float32),
           rng.normal(0, 2.0 / np.sqrt(C), (cout, C, 1, 1)).astype(np.float32), rng.normal(0, 0.1, cout).astype(np.float32),
           stride=stride, precision='f16x3')
    if split_out:
        t3 = P.tensor(32, 0, name='out', f32=True)
        P.conv(t2, t3, rng.normal(0, 0.05, (32, cout, 1, 1)).astype(np.float32), np.zeros(32, np.float32), precision='f16x3')
        P.outputs = [t3]
    else:
        P.outputs = [t2]
    return P


@pytest.mark.parametrize('C,cout,stride,split_out', [(64, 64, 1, False), (64, 128, 2, False), (128, 128, 1, False), (128, 256, 2, False),
                                                     (256, 256, 1, True), (32, 32, 1, False), (32, 64, 2, False), (96, 40, 1, False),
                                                     (128, 128, 1, True), (64, 24, 1, False)])
def test_rf_dwpw_kernel_equals_the_generic_kernel(ctx, monkeypatch, C, cout, stride, split_out):
    """Same products in the same order: every output bit of the block must be the same whichever kernel ran it -- on maps that
    do not fill their tiles (13 x 24, 5 x 3), across image boundaries (tiles of 64 / 128 raster-consecutive pixels), with cout
    tiles of 32 / 64 / 128 and partial ones (40, 24), stride 1 and 2, float32 and pre-split outputs."""
    from terran_amd import lib
    prog = _dwpw_block_program(C, cout, stride, split_out)
    tap = 'out' if split_out else 'block'
    for n, h, w in ((3, 13, 24), (2, 5, 3), (1, 40, 40), (5, 20, 20)):
        frames = ctx.upload(synth.frames(7 + n, n, h, w))
        outs, kernels = {}, {}
        for generic in (False, True):
            if generic:
                monkeypatch.setenv('TA_DWPW_GENERIC', '1')
            else:
                monkeypatch.delenv('TA_DWPW_GENERIC', raising=False)
            m = lib.Model(ctx, prog)
            ctx.kernel_work(reset=True)
            m.forward_frames(frames)
            outs[generic] = m.read(tap).copy()
            kernels[generic] = sorted(k for k in ctx.kernel_work() if 'dwpw' in k)
            assert ctx.lib.ta_debug_range_check(ctx.h) == lib.OK
            m.free()
        frames.free()
        assert all(k.startswith('rf_dwpw_kernel') for k in kernels[False]) and kernels[False], kernels
        assert all(k.startswith('conv_dwpw') for k in kernels[True]) and kernels[True], kernels
        assert np.isfinite(outs[False]).all() and np.abs(outs[False]).max() > 0
        assert np.array_equal(outs[False], outs[True]), (C, cout, stride, split_out, n, h, w, float(np.abs(outs[False] - outs[True]).max()))

"""TEST INFRASTRUCTURE: seeded weights whose STATISTICS look like a trained checkpoint's instead of an initialiser's.

`terran_amd.weights.make_*_state` draws BatchNorms with gamma = 1 +- 0.02, running_var ~ U(0.8, 1.2), running_mean ~ N(0, 0.05):
every folded layer then has rows of the same magnitude and every tensor is O(1) -- the regime in which the split-half
arithmetic (`f16x3`) trivially keeps its 22 bits.  The released checkpoints (terran/checkpoint.py:49-52,73-76,98-101) are
trained networks: per-channel BatchNorm gains and variances spread over orders of magnitude, conv rows likewise, PReLU slopes
anywhere in [0, 1].  No checkpoint is reachable offline, so these generators produce that regime synthetically:

  BatchNorm'd networks (RetinaFace, ArcFace): every conv gets a per-output-channel gain ~ logU[1e-3, 10]; its BatchNorm gets
      gamma ~ +- logU[0.05, 8], beta ~ N(0, 0.3 |gamma|) and running statistics MEASURED on a calibration batch walked through the
      network layer by layer (like a trained net's: they belong to the activations that actually arrive), times a per-channel
      mismatch: running_var x logU[1/4, 4], running_mean + N(0, 0.5 sigma).  So running_var spans ~ logU[1e-7, 1e3] and every
      layer's output is O(gamma) whatever came before: activations stay finite through 100 layers.  PReLU slopes ~ U[0, 1].
  OpenPose (no BatchNorm): a function-preserving re-parametrisation of the seeded network: hidden channel c of every
      conv -> ReLU -> conv chain is scaled by g_c ~ logU[1/64, 64] (row c of the producer and its bias times g_c, column c of
      every consumer divided by g_c; max-pool and ReLU commute with a positive gain).  The maps the network computes are the
      same up to rounding; the tensors in between span four orders of magnitude per channel.  `pow2=True` draws powers of two
      (bit-exact re-parametrisation: the decoder weights keep their exact plateaus and ties).

Everything is seeded (numpy default_rng + torch-CPU float32 convs on fixed inputs).  torch is needed to GENERATE; the states
are plain {key: ndarray} dicts like any other.  Used by tests/test_gpu_wild_weights.py and tests/probe_wild_weights.py.
"""
import numpy as np

from terran_amd import arch, synth, weights


def _logu(rng, lo, hi, n):
    return np.exp(rng.uniform(np.log(lo), np.log(hi), n))


class _Walker:
    """Walks a network on a calibration batch, making every conv / BatchNorm pair wild as it goes."""

    def __init__(self, sd, rng):
        import torch
        self.torch, self.F = torch, torch.nn.functional
        self.sd, self.rng = sd, rng

    def t(self, key):
        return self.torch.from_numpy(np.ascontiguousarray(self.sd[key]))

    def conv_bn(self, x, key_conv, key_bn, eps, bias=False, res_gamma=1.0, **kw):
        """conv (+ bias) -> BatchNorm with wild, data-matched statistics.  Returns the BatchNorm output."""
        torch, F, sd, rng = self.torch, self.F, self.sd, self.rng
        W = sd[key_conv + '.weight']
        c = W.shape[0]
        gain = _logu(rng, 1e-3, 10.0, c).astype(np.float32)
        sd[key_conv + '.weight'] = (W * gain.reshape(-1, 1, 1, 1)).astype(np.float32)
        if bias:
            sd[key_conv + '.bias'] = (sd[key_conv + '.bias'] * gain).astype(np.float32)
        with torch.no_grad():
            z = F.conv2d(x, self.t(key_conv + '.weight'), self.t(key_conv + '.bias') if bias else None, **kw)
        return self.bn(z, key_bn, eps, res_gamma)

    def bn(self, z, key_bn, eps, res_gamma=1.0):
        torch, F, sd, rng = self.torch, self.F, self.sd, self.rng
        c = z.shape[1]
        dims = [0] + list(range(2, z.dim()))
        mean = z.mean(dims).numpy().astype(np.float64)
        var = z.var(dims, unbiased=False).numpy().astype(np.float64) + 1e-30
        gamma = _logu(rng, 0.05, 8.0, c) * np.where(rng.uniform(size=c) < 0.08, -1.0, 1.0) * res_gamma
        sd[key_bn + '.weight'] = gamma.astype(np.float32)
        sd[key_bn + '.bias'] = (rng.normal(0, 0.3, c) * np.abs(gamma)).astype(np.float32)
        sd[key_bn + '.running_var'] = (var * _logu(rng, 0.25, 4.0, c)).astype(np.float32)
        sd[key_bn + '.running_mean'] = (mean + rng.normal(0, 0.5, c) * np.sqrt(var)).astype(np.float32)
        with torch.no_grad():
            return F.batch_norm(z, self.t(key_bn + '.running_mean'), self.t(key_bn + '.running_var'), self.t(key_bn + '.weight'),
                                self.t(key_bn + '.bias'), False, 0.0, eps)

    def prelu(self, x, key):
        self.sd[key] = self.rng.uniform(0.0, 1.0, self.sd[key].shape).astype(np.float32)
        return self.F.prelu(x, self.t(key))


def _calib_frames(seed, n, h, w):
    """Half smooth frames, half noisy ones: uint8 (n, h, w, 3)."""
    rng = np.random.default_rng(seed)
    fr = synth.frames(seed, n, h, w).astype(np.float32)
    fr[n // 2:] += rng.normal(0, 40.0, fr[n // 2:].shape)
    return np.clip(fr, 0, 255).astype(np.uint8)


def wild_arcface_state(seed=301):
    """ArcFace IR-ResNet100 (arcface/model.py:4-97) with trained-looking statistics."""
    import torch
    F = torch.nn.functional
    rng = np.random.default_rng(seed)
    sd = weights.make_arcface_state(seed)
    wk = _Walker(sd, rng)
    eps = arch.ARC_BN_EPS
    crops = _calib_frames(seed, 6, 112, 112)[..., ::-1].transpose(0, 3, 1, 2)                  # BGR CHW, like the wrapper's crops
    x = (torch.from_numpy(np.ascontiguousarray(crops)).float() - arch.ARC_MEAN) * arch.ARC_STD
    out = wk.conv_bn(x, 'initial_layer.0', 'initial_layer.1', eps, padding=1)
    out = wk.prelu(out, 'initial_layer.2.weight')
    for st, u, cin, cout, stride, sc in arch.arcface_units():
        p = 'stages.%d.%d' % (st, u)
        y = wk.bn(out, p + '.body.0', eps)                        # the BatchNorm in front of the zero-padded conv (model.py:12-14)
        y = wk.conv_bn(y, p + '.body.1', p + '.body.2', eps, padding=1)
        y = wk.prelu(y, p + '.body.3.weight')
        y = wk.conv_bn(y, p + '.body.4', p + '.body.5', eps, res_gamma=0.35, stride=stride, padding=1)
        s = wk.conv_bn(out, p + '.shortcut.0', p + '.shortcut.1', eps, res_gamma=0.35, stride=stride) if sc else out
        out = y + s
    out = wk.bn(out, 'final_layer.0', eps)
    g = _logu(rng, 1e-2, 10.0, 512).astype(np.float32)
    sd['final_layer.3.weight'] = sd['final_layer.3.weight'] * g[:, None]
    sd['final_layer.3.bias'] = sd['final_layer.3.bias'] * g
    with torch.no_grad():
        z = F.linear(out.flatten(1), wk.t('final_layer.3.weight'), wk.t('final_layer.3.bias'))
    wk.bn(z, 'final_layer.4', eps)
    return sd


def wild_retinaface_state(seed=300, fg_bias=-4.0):
    """RetinaFace-mnet (retinaface/model.py:53-316) with trained-looking statistics; the heads (no BatchNorm) are re-scaled
    so that scores, box deltas and landmark offsets keep the magnitudes of `weights.make_retinaface_state`."""
    import torch
    F = torch.nn.functional
    rng = np.random.default_rng(seed)
    sd = weights.make_retinaface_state(seed, fg_bias=fg_bias)
    wk = _Walker(sd, rng)
    eps = arch.RETINA_BASE_BN_EPS
    fr = _calib_frames(seed, 4, 160, 224)
    x = torch.from_numpy(np.ascontiguousarray(fr[..., ::-1].transpose(0, 3, 1, 2))).float()    # BGR 0..255 (wrapper.py:144-146)

    def cbr(kc, kb, x, e=eps, bias=False, **kw):
        return F.relu(wk.conv_bn(x, kc, kb, e, bias=bias, **kw))
    out = cbr('base.first_conv_block.0', 'base.first_conv_block.1', x, stride=2, padding=1)
    out = cbr('base.first_conv_block.3', 'base.first_conv_block.4', out, padding=1, groups=8)
    feats = []
    for si, scale in enumerate(arch.RETINA_SCALES):
        for bi, (cin, cout, stride, both) in enumerate(scale):
            p = 'base.scales.%d.%d' % (si, bi)
            conv = cbr(p + '.conv_block.0', p + '.conv_block.1', out)
            out = cbr(p + '.sep_block.0', p + '.sep_block.1', conv, stride=stride, padding=1, groups=cout)
            if both:
                feats.append(conv)
    p = 'base.final_conv.0'
    conv = cbr(p + '.conv_block.0', p + '.conv_block.1', out)
    out = cbr(p + '.sep_block.0', p + '.sep_block.1', conv, padding=1, groups=256)
    feats.append(cbr('base.final_conv.1', 'base.final_conv.2', out))
    s8, s16, s32 = feats
    e2 = arch.RETINA_REFINER_BN_EPS

    def cbr2(p, x, padding=0):
        return cbr(p + '.0', p + '.1', x, e2, bias=True, padding=padding)
    p8, p16, p32 = cbr2('refiner.conv_stride8', s8), cbr2('refiner.conv_stride16', s16), cbr2('refiner.conv_stride32', s32)
    up32 = F.interpolate(p32, scale_factor=2)[:, :, :p16.shape[2], :p16.shape[3]]
    p16 = cbr2('refiner.aggr_stride16', p16 + up32, padding=1)
    up16 = F.interpolate(p16, scale_factor=2)[:, :, :p8.shape[2], :p8.shape[3]]
    p8 = cbr2('refiner.aggr_stride8', p8 + up16, padding=1)

    def context(p, x):
        def c(name, i, x):
            return cbr('%s.%s.%d' % (p, name, i), '%s.%s.%d' % (p, name, i + 1), x, e2, bias=True, padding=1)
        red = c('dimension_reducer', 0, x)
        return torch.cat([c('context_3x3', 0, x), c('context_5x5', 0, red), c('context_7x7', 3, c('context_7x7', 0, red))], 1)
    ctx = {8: context('refiner.context_stride8', p8), 16: context('refiner.context_stride16', p16),
           32: context('refiner.context_stride32', p32)}
    A = arch.RETINA_NUM_ANCHORS
    for s in (8, 16, 32):
        for head, target in (('cls', 2.5), ('bbox', 0.2), ('landmark', 0.35)):
            key = 'outputs.%s_stride%d' % (head, s)
            with torch.no_grad():
                z = F.conv2d(ctx[s], wk.t(key + '.weight'))
            sd[key + '.weight'] = (sd[key + '.weight'] * np.float32(target / max(float(z.std()), 1e-12))).astype(np.float32)
    return sd


def _openpose_rescale(sd, rng, lo, hi, pow2):
    """Per-channel gains on every hidden tensor of the OpenPose graph (openpose/model.py:27-141), function-preserving."""
    def gains(c):
        if pow2:
            return np.ldexp(1.0, rng.integers(int(np.log2(lo)), int(np.log2(hi)) + 1, c)).astype(np.float32)
        return _logu(rng, lo, hi, c).astype(np.float32)

    def scale_out(key, g):
        sd[key + '.weight'] = (sd[key + '.weight'] * g.reshape(-1, 1, 1, 1)).astype(np.float32)
        sd[key + '.bias'] = (sd[key + '.bias'] * g).astype(np.float32)

    def scale_in(key, g, ch0=0):
        W = sd[key + '.weight'].copy()
        W[:, ch0:ch0 + len(g)] = W[:, ch0:ch0 + len(g)] / g.reshape(1, -1, 1, 1)
        sd[key + '.weight'] = W.astype(np.float32)
    convs = [it for it in arch.OPENPOSE_MODEL0 if it[0] != 'pool']
    for (name, cin, cout, k), nxt in zip(convs, convs[1:] + [None]):
        g = gains(cout)
        scale_out('model0.' + name, g)
        if nxt is not None:
            scale_in('model0.' + nxt[0], g)
        else:                                                     # feat (128 ch): stage 1 reads it alone, stages 2..6 behind PAF38 | HM19
            for b in (1, 2):
                scale_in('model1_%d.%s' % (b, arch.openpose_stage_layers(1, b)[0][0]), g)
                for t in range(2, 7):
                    scale_in('model%d_%d.%s' % (t, b, arch.openpose_stage_layers(t, b)[0][0]), g, ch0=57)
    for t in range(1, 7):
        for b in (1, 2):
            layers = arch.openpose_stage_layers(t, b)
            for (name, cin, cout, k, relu), nxt in zip(layers[:-1], layers[1:]):
                assert relu                                        # every hidden layer of a branch ends in a ReLU
                g = gains(cout)
                scale_out('model%d_%d.%s' % (t, b, name), g)
                scale_in('model%d_%d.%s' % (t, b, nxt[0]), g)
    return sd


def wild_openpose_state(seed=302, lo=1.0 / 64, hi=64.0):
    """Seeded random OpenPose with wild per-channel gains on every hidden tensor (arbitrary positive gains)."""
    return _openpose_rescale(weights.make_openpose_state(seed), np.random.default_rng(seed), lo, hi, pow2=False)


def wild_openpose_decoder_state(seed=weights.SEED_OPENPOSE, lo=1.0 / 64, hi=64.0):
    """The decoder network of the end-to-end pose tests (people DO assemble) under power-of-two channel gains: float32
    computes bit for bit the maps of the plain decoder network (exact plateaus and ties survive), while every hidden tensor's
    channels sit anywhere in 2^-6 .. 2^6 of their usual magnitude."""
    return _openpose_rescale(weights.make_openpose_decoder_state(seed), np.random.default_rng(seed + 1), lo, hi, pow2=True)


MAKERS = {'retinaface': wild_retinaface_state, 'arcface': wild_arcface_state, 'openpose': wild_openpose_state,
          'openpose_decoder': wild_openpose_decoder_state}


def describe(sd, kind):
    """Spread of what makes the weights 'wild': BatchNorm gamma / running_var, folded row norms, PReLU slopes."""
    out = {}
    gs = [np.abs(v).ravel() for k, v in sd.items() if k.endswith('.weight') and (k[:-7] + '.running_var') in sd]
    g = np.concatenate(gs) if gs else np.zeros(0)
    v = np.concatenate([sd[k].ravel() for k in sd if k.endswith('.running_var')]) if g.size else np.zeros(0)
    if g.size:
        out['bn_gamma_min_max'] = (float(g.min()), float(g.max()))
        out['bn_running_var_min_max'] = (float(v.min()), float(v.max()))
    rows = []
    for k, w in sd.items():
        if k.endswith('.weight') and getattr(w, 'ndim', 0) == 4:
            r = np.abs(w.reshape(w.shape[0], -1)).max(1)
            r = r[r > 0]
            if r.size:
                rows.append(float(r.max() / r.min()))
    out['largest_row_max_ratio_within_a_conv'] = max(rows) if rows else None
    sl = [v for k, v in sd.items() if '.body.3.' in k or k == 'initial_layer.2.weight']
    if sl:
        sl = np.concatenate([x.ravel() for x in sl])
        out['prelu_slope_min_max'] = (float(sl.min()), float(sl.max()))
    return out

"""ORACLE (test infrastructure, never the product path): CPU fp32 restatement of
the three network graphs of the hot path, written functionally over a
`{state_dict key: ndarray}` mapping with torch-CPU primitives.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package.  The HIP path in `terran_amd/` never does.

Restates (does not import) the reference graphs:
  retinaface_forward : terran/face/detection/retinaface/model.py:53-112 (base),
                       168-245 (refiner + context), 248-316 (heads + softmax)
  arcface_forward    : terran/face/recognition/arcface/model.py:4-35 (Unit), 38-97
  openpose_forward   : terran/pose/openpose/model.py:27-141

Pinned against the imported reference modules by `tests/golden/make_golden.py`
(container-only) -> `tests/golden/*.npz`, and re-checked by
`tests/test_oracle_golden.py` everywhere.
"""
import numpy as np
import torch
import torch.nn.functional as F

from terran_amd import arch


def _t(sd, key):
    v = sd[key]
    return v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))


def _bn(sd, key, x, eps):
    return F.batch_norm(x, _t(sd, key + '.running_mean'), _t(sd, key + '.running_var'),
                        _t(sd, key + '.weight'), _t(sd, key + '.bias'), False, 0.0, eps)


def _conv(sd, key, x, stride=1, padding=0, groups=1, bias=False):
    b = _t(sd, key + '.bias') if bias else None
    return F.conv2d(x, _t(sd, key + '.weight'), b, stride, padding, 1, groups)


# ----------------------------------------------------------------------------
# RetinaFace
# ----------------------------------------------------------------------------
def retinaface_forward(sd, x, taps=None):
    """x: (N,3,H,W) float32 BGR 0..255 (wrapper.py:144-146).  Returns the nine
    outputs in the reference order [cls32,bbox32,lmk32, cls16,.., cls8,..]
    (model.py:304-316).  `taps`, if a dict, receives named intermediates."""
    eps = arch.RETINA_BASE_BN_EPS

    def cbr(key_conv, key_bn, x, **kw):
        return F.relu(_bn(sd, key_bn, _conv(sd, key_conv, x, **kw), eps))

    with torch.no_grad():
        out = cbr('base.first_conv_block.0', 'base.first_conv_block.1', x, stride=2, padding=1)
        out = cbr('base.first_conv_block.3', 'base.first_conv_block.4', out, padding=1, groups=8)
        if taps is not None:
            taps['stem'] = out
        feats = []
        for si, scale in enumerate(arch.RETINA_SCALES):
            for bi, (cin, cout, stride, both) in enumerate(scale):
                p = 'base.scales.%d.%d' % (si, bi)
                conv = cbr(p + '.conv_block.0', p + '.conv_block.1', out)
                out = cbr(p + '.sep_block.0', p + '.sep_block.1', conv,
                          stride=stride, padding=1, groups=cout)
                if both:
                    feats.append(conv)
        p = 'base.final_conv.0'
        conv = cbr(p + '.conv_block.0', p + '.conv_block.1', out)
        out = cbr(p + '.sep_block.0', p + '.sep_block.1', conv, padding=1, groups=256)
        out = cbr('base.final_conv.1', 'base.final_conv.2', out)
        feats.append(out)
        s8, s16, s32 = feats
        if taps is not None:
            taps['feat8'], taps['feat16'], taps['feat32'] = s8, s16, s32

        eps = arch.RETINA_REFINER_BN_EPS

        def cbr2(p, x, padding=0):
            return F.relu(_bn(sd, p + '.1', _conv(sd, p + '.0', x, padding=padding, bias=True), eps))

        p8 = cbr2('refiner.conv_stride8', s8)
        p16 = cbr2('refiner.conv_stride16', s16)
        p32 = cbr2('refiner.conv_stride32', s32)
        up32 = F.interpolate(p32, scale_factor=2)[:, :, :p16.shape[2], :p16.shape[3]]
        p16 = cbr2('refiner.aggr_stride16', p16 + up32, padding=1)
        up16 = F.interpolate(p16, scale_factor=2)[:, :, :p8.shape[2], :p8.shape[3]]
        p8 = cbr2('refiner.aggr_stride8', p8 + up16, padding=1)
        if taps is not None:
            taps['p8'], taps['p16'], taps['p32'] = p8, p16, p32

        def context(p, x):
            def c(name, i, x):
                return F.relu(_bn(sd, '%s.%s.%d' % (p, name, i + 1),
                                  _conv(sd, '%s.%s.%d' % (p, name, i), x, padding=1, bias=True), eps))
            red = c('dimension_reducer', 0, x)
            c3 = c('context_3x3', 0, x)
            c5 = c('context_5x5', 0, red)
            c7 = c('context_7x7', 3, c('context_7x7', 0, red))
            return torch.cat([c3, c5, c7], dim=1)

        ctx = {8: context('refiner.context_stride8', p8),
               16: context('refiner.context_stride16', p16),
               32: context('refiner.context_stride32', p32)}     # un-aggregated lateral, model.py:243
        if taps is not None:
            taps['ctx8'], taps['ctx16'], taps['ctx32'] = ctx[8], ctx[16], ctx[32]

        outs = []
        for s in arch.RETINA_STRIDES:
            cls = _conv(sd, 'outputs.cls_stride%d' % s, ctx[s], bias=True)
            N, A, H, W = cls.shape
            # softmax over the channel pair (a, a+A): model.py:283-294
            prob = F.softmax(cls.contiguous().view(N, 2, -1, W), dim=1).view(N, A, H, W)
            bbox = _conv(sd, 'outputs.bbox_stride%d' % s, ctx[s], bias=True)
            lmk = _conv(sd, 'outputs.landmark_stride%d' % s, ctx[s], bias=True)
            outs += [prob, bbox, lmk]
    return outs


# ----------------------------------------------------------------------------
# ArcFace
# ----------------------------------------------------------------------------
def arcface_forward(sd, x, taps=None):
    """x: (N,3,112,112) float32 with uint8 values, BGR (arcface/wrapper.py:72,166-172).
    Returns (N,512) un-normalised embeddings (model.py:87-97)."""
    eps = arch.ARC_BN_EPS
    with torch.no_grad():
        out = (x - arch.ARC_MEAN) * arch.ARC_STD
        out = _conv(sd, 'initial_layer.0', out, padding=1)
        out = F.prelu(_bn(sd, 'initial_layer.1', out, eps), _t(sd, 'initial_layer.2.weight'))
        if taps is not None:
            taps['stem'] = out
        for st, u, cin, cout, stride, sc in arch.arcface_units():
            p = 'stages.%d.%d' % (st, u)
            y = _bn(sd, p + '.body.0', out, eps)
            y = _conv(sd, p + '.body.1', y, padding=1)
            y = F.prelu(_bn(sd, p + '.body.2', y, eps), _t(sd, p + '.body.3.weight'))
            y = _conv(sd, p + '.body.4', y, stride=stride, padding=1)
            y = _bn(sd, p + '.body.5', y, eps)
            if sc:
                s = _bn(sd, p + '.shortcut.1', _conv(sd, p + '.shortcut.0', out, stride=stride), eps)
            else:
                s = out
            out = y + s
            if taps is not None and u == arch.ARC_UNITS[st] - 1:
                taps['stage%d' % (st + 1)] = out
        out = _bn(sd, 'final_layer.0', out, eps)
        out = out.flatten(1)                     # (C,H,W) order
        out = F.linear(out, _t(sd, 'final_layer.3.weight'), _t(sd, 'final_layer.3.bias'))
        out = F.batch_norm(out, _t(sd, 'final_layer.4.running_mean'), _t(sd, 'final_layer.4.running_var'),
                           _t(sd, 'final_layer.4.weight'), _t(sd, 'final_layer.4.bias'), False, 0.0, eps)
    return out


# ----------------------------------------------------------------------------
# OpenPose
# ----------------------------------------------------------------------------
def openpose_forward(sd, x, taps=None):
    """x: (N,3,H,W) float32 RGB/255-0.5 (openpose/wrapper.py:116-122).
    Returns (pafs (N,38,h,w), heatmaps (N,19,h,w)) of the 6th stage."""
    with torch.no_grad():
        out = x
        for item in arch.OPENPOSE_MODEL0:
            if item[0] == 'pool':
                out = F.max_pool2d(out, 2, 2, 0)
            else:
                name, cin, cout, k = item
                out = F.relu(_conv(sd, 'model0.' + name, out, padding=k // 2, bias=True))
        feat = out
        if taps is not None:
            taps['feat'] = feat
        inp = feat
        for t in range(1, 7):
            outs = []
            for b in (1, 2):
                y = inp
                for name, cin, cout, k, relu in arch.openpose_stage_layers(t, b):
                    y = _conv(sd, 'model%d_%d.%s' % (t, b, name), y, padding=k // 2, bias=True)
                    if relu:
                        y = F.relu(y)
                outs.append(y)
            if taps is not None:
                taps['stage%d_paf' % t], taps['stage%d_hm' % t] = outs
            if t < 6:
                inp = torch.cat([outs[0], outs[1], feat], 1)
    return outs[0], outs[1]

"""Tools: per op, the largest |x| a program STORES (ta_model_debug_amax) against the packer's bound (pack.Program.expected_amax):
where does an activation leave the range the packer expected?

    python tools/amax_debug.py arcface [f16x3] [--wild | --stem-gain 22]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import lib, pack, runtime, synth, weights            # noqa: E402


def report(kind, sd, prec, run, top=12):
    ctx = runtime.get_context(0)
    P = getattr(pack, 'pack_' + kind)(sd, prec)
    P.blob()
    m = lib.Model(ctx, P)
    m.amax_collect(True)
    run(m)
    got = m.amax_read(len(P.ops))
    rc = ctx.lib.ta_debug_range_check(ctx.h)
    rows = []
    for i, op in enumerate(P.ops):
        if op['type'] not in (pack.OP_CONV, pack.OP_DWPW, pack.OP_RFSTEM):
            continue
        sl = slice(op['out_ch_off'], op['out_ch_off'] + op['cout'])
        a = P.scales[op['out']][sl]
        bound = P.expected_amax(op['out'], per_channel=True)[sl] * 2.0 ** a
        rows.append((got[i, 0] / max(bound.max(), 1e-30), i, op['type'], op['cin'], op['cout'], op['kh'], float(got[i, 0]), float(bound.max()),
                     int(a.min()), int(a.max())))
    bad = [(i, got[i, 0], got[i, 1]) for i in range(len(P.ops)) if not (np.isfinite(got[i]).all() and got[i].max() <= 65504.0)]
    print('ops that stored inf / NaN / > 65504:', bad)
    rows = [r for r in rows if np.isfinite(r[0])]
    rows.sort(reverse=True)
    print('%s %s: range check rc = %d; ops by stored max / expected stored bound' % (kind, prec, rc))
    for r in rows[:top]:
        print('  ratio %8.3g  op %3d type %d cin %4d cout %4d k%d  stored max %10.4g  bound %8.4g  exponents %d..%d' % r)
    m.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('kind')
    ap.add_argument('prec', nargs='?', default='f16x3')
    ap.add_argument('--wild', action='store_true')
    ap.add_argument('--stem-gain', type=int, default=0)
    args = ap.parse_args()
    if args.wild:
        from tests import wild_weights
        sd = wild_weights.MAKERS[args.kind]()
    else:
        sd = getattr(weights, 'make_%s_state' % args.kind)()
    if args.kind == 'arcface':
        if args.stem_gain:
            sd = dict(sd)
            g = np.float32(2.0 ** args.stem_gain)
            for k in ('weight', 'bias'):
                sd['initial_layer.1.' + k] = np.asarray(sd['initial_layer.1.' + k], np.float32) * g
            sd['stages.0.0.body.0.running_mean'] = np.asarray(sd['stages.0.0.body.0.running_mean'], np.float32) * g
            sd['stages.0.0.body.0.running_var'] = np.asarray(sd['stages.0.0.body.0.running_var'], np.float32) * g * g
        crops = np.random.default_rng(8).integers(0, 256, (5, 3, 112, 112), dtype=np.uint8)
        report('arcface', sd, args.prec, lambda m: m.forward_crops(crops))
    else:
        fr = runtime.get_context(0).upload(synth.frames(4000, 2, 184, 327))
        report(args.kind, sd, args.prec, lambda m: m.forward_frames(fr))


if __name__ == '__main__':
    main()

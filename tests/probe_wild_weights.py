"""Probe (GPU box): how do the arithmetic modes behave on weights with trained-looking statistics (tests/wild_weights.py)?

    python tests/probe_wild_weights.py [--frames N] > profiles/r04_wild_weights.txt

Per network and mode: decisions that differ from the ORACLE (same counting as tests/test_gpu_decisions_vs_oracle.py),
embedding error, TA_E_RANGE fallbacks, and how well the packer's EXPECTED maximum of every tensor (pack.Program.expected_amax,
the basis of the activation scales) matches the maximum the device actually stored (ta_model_debug_amax).

Trained-looking statistics make the networks ill-conditioned enough that two float32 evaluations of one network (the
oracle's torch-CPU convs and the device's exact-f32 MFMA: different summation orders, BatchNorm applied vs folded) decide a
fraction of a percent of the near-ties differently.  To tell arithmetic quality from that noise every row is ALSO counted
against a FLOAT64 evaluation of the same network (the oracle's graphs run in double, post-processing unchanged): `vs f64` is
the distance to the truth, and the float32 oracle itself gets a row.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from terran_amd import lib, pack, runtime, synth, weights                       # noqa: E402
from tests import wild_weights                                                    # noqa: E402
from tests import test_gpu_decisions_vs_oracle as D                                # noqa: E402

MODES = ('f32', 'f16x3', 'f16')


def amax_report(kind, sd, prec, run):
    """Expected vs stored maxima per op output.  run(model) executes one forward of a lib.Model on representative input."""
    ctx = runtime.get_context(0)
    P = getattr(pack, 'pack_' + kind)(sd, prec)
    P.blob()
    m = lib.Model(ctx, P)
    m.amax_collect(True)
    run(m)
    got = m.amax_read(len(P.ops))
    ratios, spread = [], []
    for i, op in enumerate(P.ops):
        if op['type'] not in (pack.OP_CONV, pack.OP_DWPW, pack.OP_RFSTEM) or got[i, 0] == 0:
            continue
        sl = slice(op['out_ch_off'], op['out_ch_off'] + op['cout'])
        a = P.scales[op['out']][sl]
        bound = P.expected_amax(op['out'], per_channel=True)[sl] * 2.0 ** a        # stored units: <= 2^10 by construction
        if bound.max() > 0:
            ratios.append(got[i, 0] / bound.max())
            spread.append(int(a.max() - a.min()))
    m.amax_collect(False)
    m.free()
    ratios = np.array(ratios)
    return dict(ops=len(ratios), stored_max=float(max(got[:, 0].max(), got[:, 1].max())), headroom_to_65504=float(65504.0 / max(got[:, 0].max(), got[:, 1].max())),
                measured_over_expected_min=float(ratios.min()), measured_over_expected_median=float(np.median(ratios)),
                measured_over_expected_max=float(ratios.max()), exponents_min_max=(int(min(x.min() for x in P.scales)), int(max(x.max() for x in P.scales))),
                largest_exponent_spread_within_a_tensor=int(max(spread)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=64, help='frames per task at the small size (16 more at the working size)')
    ap.add_argument('--stats', default='wild')
    args = ap.parse_args()
    B = D.BATCH
    for stats in args.stats.split(','):
        mk = (lambda k: wild_weights.MAKERS[k]()) if stats == 'wild' else (lambda k: getattr(weights, 'make_%s_state' % k)())
        sd_r, sd_a, sd_p, sd_d = mk('retinaface'), mk('arcface'), mk('openpose'), mk('openpose_decoder')
        print('==== %s weights ====' % stats)
        if stats == 'wild':
            for k, sd in (('retinaface', sd_r), ('arcface', sd_a), ('openpose', sd_p)):
                print('  %s: %s' % (k, wild_weights.describe(sd, k)))
        # ---- detector and pose: decisions against the float64 evaluation (and against the float32 oracle)
        from terran_amd import ArcFace
        if stats == 'wild':
            tot = D.wild_table(lambda name: {'wild_retinaface': sd_r, 'wild_openpose': sd_p, 'wild_openpose_decoder': sd_d}[name],
                               modes=MODES)
            for task, rows in tot.items():
                print('  TOTAL %s, flips against the float64 evaluation: %s' % (task, rows))
        fr = runtime.get_context(0).upload(synth.frames(6000, 4, 416, 739))
        print('  retinaface f16x3 amax:', amax_report('retinaface', sd_r, 'f16x3', lambda m: m.forward_frames(fr)))
        fr.free()
        fr = runtime.get_context(0).upload(synth.frames(4000, 4, 184, 327))
        print('  openpose f16x3 amax:', amax_report('openpose', sd_p, 'f16x3', lambda m: m.forward_frames(fr)))
        fr.free()
        # ---- embeddings
        from oracle import arcface_pre, nets
        rng = np.random.default_rng(5)
        crops = rng.integers(0, 256, (64, 3, 112, 112), dtype=np.uint8)
        crops[32:] = wild_weights._calib_frames(77, 32, 112, 112)[..., ::-1].transpose(0, 3, 1, 2)    # smooth, image-like crops
        ref = arcface_pre.l2_normalize(nets.arcface_forward(sd_a, torch.from_numpy(crops.astype(np.float32))).numpy())
        e64 = nets.arcface_forward(D.f64_state(sd_a), torch.from_numpy(crops.astype(np.float64))).numpy()
        truth = e64 / np.sqrt((e64 * e64).sum(1, keepdims=True))
        print('  arcface 64 crops oracle (f32): max |d| of unit embeddings vs f64 %.3g' % np.abs(ref - truth).max())
        for mode in MODES:
            a = ArcFace(device=0, state=sd_a, precision=mode)
            e = a.embed_crops(crops)
            print('  arcface 64 crops device %-5s: max |d| of unit embeddings vs f64 %.3g, vs oracle %.3g (noise crops %.3g, smooth crops %.3g), '
                  'max cosine distance %.3g, range fallbacks %d' % (mode, np.abs(e - truth).max(), np.abs(e - ref).max(), np.abs(e - ref)[:32].max(),
                                                                   np.abs(e - ref)[32:].max(), 1.0 - (e * ref).sum(1).min(), a.fallbacks))
        for mode in ('f16x3', 'f16'):
            print('  arcface %s amax:' % mode, amax_report('arcface', sd_a, mode, lambda m: m.forward_crops(crops)))
        runtime.clear_pack_memo()


if __name__ == '__main__':
    main()

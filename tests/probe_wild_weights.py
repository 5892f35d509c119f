"""Probe (GPU box): how do the arithmetic modes behave on weights with trained-looking statistics (tests/wild_weights.py)?

    python tests/probe_wild_weights.py [--frames N] > profiles/r04_wild_weights.txt

Per network and mode: decisions that differ from the ORACLE (same counting as tests/test_gpu_decisions_vs_oracle.py),
embedding error, TA_E_RANGE fallbacks, and how well the packer's EXPECTED maximum of every tensor (pack.Program.expected_amax,
the basis of the activation scales) matches the maximum the device actually stored (ta_model_debug_amax).
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
    ratios, worst = [], None
    for i, op in enumerate(P.ops):
        if op['type'] not in (pack.OP_CONV, pack.OP_DWPW) or got[i, 0] == 0:
            continue
        exp = P.expected_amax(op['out']) * 2.0 ** P.scales[op['out']]
        # the op wrote a slice: compare with the expectation of the whole tensor (what the scale was chosen from)
        r = got[i, 0] / exp if exp > 0 else np.inf
        ratios.append(r)
        if worst is None or got[i, 0] > worst[1]:
            worst = (i, float(got[i, 0]), float(exp), P.scales[op['out']])
    m.amax_collect(False)
    m.free()
    ratios = np.array(ratios)
    return dict(ops=len(ratios), stored_max=float(max(got[:, 0].max(), got[:, 1].max())), headroom_to_65504=float(65504.0 / max(got[:, 0].max(), got[:, 1].max())),
                measured_over_expected_min=float(ratios.min()), measured_over_expected_median=float(np.median(ratios)),
                measured_over_expected_max=float(ratios.max()), scales_min_max=(int(min(P.scales)), int(max(P.scales))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=64, help='frames per task at the small size (16 more at the working size)')
    ap.add_argument('--stats', default='wild,benign')
    args = ap.parse_args()
    B = D.BATCH
    for stats in args.stats.split(','):
        mk = (lambda k: wild_weights.MAKERS[k]()) if stats == 'wild' else (lambda k: getattr(weights, 'make_%s_state' % k)())
        sd_r, sd_a, sd_p, sd_d = mk('retinaface'), mk('arcface'), mk('openpose'), mk('openpose_decoder')
        print('==== %s weights ====' % stats)
        if stats == 'wild':
            for k, sd in (('retinaface', sd_r), ('arcface', sd_a), ('openpose', sd_p)):
                print('  %s: %s' % (k, wild_weights.describe(sd, k)))
        # ---- detector
        from oracle import pipeline
        from terran_amd import ArcFace, OpenPose, RetinaFace
        for case, (gen, n) in (('208x277', (lambda k: synth.frames(1000 + k, B, 208, 277), args.frames)),
                               ('416x739', (lambda k: synth.frames(6000 + k, B, 416, 739), 16))):
            ref = []
            for k in range(0, n, B):
                ref += [D._det_keys(d) for d in pipeline.retinaface_call(sd_r, gen(k))]
            for mode in MODES:
                model = RetinaFace(device=0, state=sd_r, precision=mode)
                tot = dict(images=0, dets=0, ddets=0, images_reordered=0)
                for k in range(0, n, B):
                    for g, r in zip(model.call(gen(k)), ref[k:k + B]):
                        g = D._det_keys(g)
                        tot['images'] += 1
                        tot['dets'] += len(r)
                        tot['ddets'] += len(set(g) ^ set(r))
                        tot['images_reordered'] += int(set(g) == set(r) and g != r)
                tot['range_fallbacks'] = model.fallbacks
                print('  retinaface %s %-5s vs oracle: %s' % (case, mode, tot))
        fr = runtime.get_context(0).upload(synth.frames(6000, 4, 416, 739))
        print('  retinaface f16x3 amax:', amax_report('retinaface', sd_r, 'f16x3', lambda m: m.forward_frames(fr)))
        fr.free()
        # ---- pose
        for case, (sd, gen, short, n) in (('random 96x128', (sd_p, lambda k: synth.frames(2000 + k, B, 96, 128), 96, args.frames)),
                                          ('people 96x128', (sd_d, lambda k: synth.pose_code_frames(3000 + k, B, 96, 128, 3), 96, args.frames)),
                                          ('random 184x327', (sd_p, lambda k: synth.frames(4000 + k, B, 184, 327), 184, 16)),
                                          ('people 184x327', (sd_d, lambda k: synth.pose_code_frames(5000 + k, B, 184, 327, 4), 184, 16))):
            ref = []
            for k in range(0, n, B):
                ref += D._pose_sets_oracle(sd, gen(k), short)
            for mode in MODES[:2]:                                        # the pose network of the f16 mode IS the f16x3 program
                model = OpenPose(device=0, short_side=short, state=sd, precision=mode)
                tot = dict(frames=0, peaks=0, dpeaks=0, conns=0, dconns=0, humans=0, dhumans=0)
                for k in range(0, n, B):
                    for (gp, gc, gh), (rp, rc, rh) in zip(D._pose_sets_device(model, gen(k)), ref[k:k + B]):
                        tot['frames'] += 1
                        tot['peaks'] += len(rp)
                        tot['dpeaks'] += len(gp ^ rp)
                        tot['conns'] += len(rc)
                        tot['dconns'] += len(gc ^ rc)
                        tot['humans'] += len(rh)
                        tot['dhumans'] += len(set(gh) ^ set(rh))
                tot['range_fallbacks'] = model.fallbacks
                print('  openpose %s %-5s vs oracle: %s' % (case, mode, tot))
        fr = runtime.get_context(0).upload(synth.frames(4000, 4, 184, 327))
        print('  openpose f16x3 amax:', amax_report('openpose', sd_p, 'f16x3', lambda m: m.forward_frames(fr)))
        fr.free()
        # ---- embeddings
        from oracle import arcface_pre, nets
        rng = np.random.default_rng(5)
        crops = rng.integers(0, 256, (64, 3, 112, 112), dtype=np.uint8)
        crops[32:] = wild_weights._calib_frames(77, 32, 112, 112)[..., ::-1].transpose(0, 3, 1, 2)    # smooth, image-like crops
        ref = arcface_pre.l2_normalize(nets.arcface_forward(sd_a, torch.from_numpy(crops.astype(np.float32))).numpy())
        for mode in MODES:
            a = ArcFace(device=0, state=sd_a, precision=mode)
            e = a.embed_crops(crops)
            print('  arcface 64 crops %-5s vs oracle: max |d| of unit embeddings %.3g (noise crops %.3g, smooth crops %.3g), max cosine distance %.3g, '
                  'range fallbacks %d' % (mode, np.abs(e - ref).max(), np.abs(e - ref)[:32].max(), np.abs(e - ref)[32:].max(),
                                          1.0 - (e * ref).sum(1).min(), a.fallbacks))
        for mode in ('f16x3', 'f16'):
            print('  arcface %s amax:' % mode, amax_report('arcface', sd_a, mode, lambda m: m.forward_crops(crops)))
        runtime.clear_pack_memo()


if __name__ == '__main__':
    main()

"""Randomised post-processing parity: device (HIP) vs oracle, exact comparisons (run on the GPU box).

    python tests/fuzz_post.py [rounds] [seed]
* RetinaFace decode + threshold + sort + NMS on random head tensors (dw = dh = 0 so exp() is exact and every IoU
  comparison bit-identical; thresholds hit exactly; score ties; 0 .. thousands of candidates).
* OpenPose grouping on synthetic pose maps: random people counts, map sizes, scales, noise levels, dropped parts.
Any difference in counts, order, integer outputs or scores is printed with the generating parameters."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import openpose_post, retinaface_post           # noqa: E402
from terran_amd import lib, openpose, retinaface, synth     # noqa: E402


def rf_case(ctx, rng):
    H, W, N = int(rng.integers(8, 200)), int(rng.integers(8, 260)), int(rng.integers(1, 5))
    dens = float(rng.choice([0.0, 0.02, 0.2, 0.6, 1.0]))
    heads = []
    for s in (32, 16, 8):
        fh, fw = -(-H // s), -(-W // s)
        prob = rng.uniform(0, dens, (N, 4, fh, fw)).astype(np.float32) if dens else np.zeros((N, 4, fh, fw), np.float32)
        prob[:, 2:][rng.uniform(size=(N, 2, fh, fw)) < 0.1] = 0.5
        prob[:, 2:][rng.uniform(size=(N, 2, fh, fw)) < 0.1 * dens] = 0.75
        bbox = rng.normal(0, float(rng.choice([0.1, 0.35, 1.0])), (N, 8, fh, fw)).astype(np.float32)
        bbox[:, [2, 3, 6, 7]] = 0.0
        lmk = rng.normal(0, 0.3, (N, 20, fh, fw)).astype(np.float32)
        heads += [prob, bbox, lmk]
    thr, nms = float(rng.choice([0.5, 0.3, 0.75])), float(rng.choice([0.4, 0.2, 0.6]))
    ref = retinaface_post.postprocess(heads, H, W, thr, nms)
    got = retinaface.postprocess(ctx, heads, H, W, thr, nms)
    ok = [len(g) for g in got] == [len(r) for r in ref]
    if ok:
        for g, r in zip(got, ref):
            for a, b in zip(g, r):
                ok &= bool(np.array_equal(a['bbox'], b['bbox']) and a['score'] == b['score'] and
                           np.array_equal(a['landmarks'], b['landmarks']))
    return ok, dict(H=H, W=W, N=N, dens=dens, thr=thr, nms=nms, kept=sum(len(r) for r in ref))


def op_case(ctx, rng):
    seed = int(rng.integers(0, 1 << 30))
    P, n = int(rng.integers(0, 13)), int(rng.integers(1, 4))
    h, w = int(rng.integers(6, 40)), int(rng.integers(6, 56))
    scale = float(rng.choice([1.0, 0.37, 1.7, 0.17]))
    kw = dict(noise=float(rng.choice([0.0, 0.01, 0.05])), drop_prob=float(rng.choice([0.0, 0.1, 0.4])))
    hm, paf = synth.pose_maps_batch(seed, n, P, h, w, **kw)
    if rng.random() < 0.08 and h >= 10 and w >= 12:      # a plateau: > 1024 peaks of one part -> the global-memory re-run
        i, part = int(rng.integers(0, n)), int(rng.integers(0, 18))
        y0, x0 = int(rng.integers(1, h - 8)), int(rng.integers(1, w - 10))
        hm[i, part, y0:y0 + 7, x0:x0 + 9] = float(rng.choice([0.3, 0.5, 0.9]))
        kw = dict(kw, plateau=(i, part))
    ref = openpose_post.postprocess(paf, hm, scale)
    try:
        got = openpose.group(ctx, paf, hm, scale)
    except openpose.PoseOverflow as e:              # an image over a device cap: the others must still match
        ok = all(g is None or [(a['keypoints'].tolist(), a['score']) for a in g] == [(b['keypoints'].tolist(), b['score']) for b in r]
                 for g, r in zip(e.results, ref))
        return ok, dict(seed=seed, P=P, n=n, h=h, w=w, scale=scale, overflow=e.images, **kw)
    ok = [len(p) for p in got] == [len(p) for p in ref]
    if ok:
        for gp, rp in zip(got, ref):
            for a, b in zip(gp, rp):
                ok &= bool(np.array_equal(a['keypoints'], b['keypoints']) and a['score'] == b['score'])
    return ok, dict(seed=seed, P=P, n=n, h=h, w=w, scale=scale, humans=sum(len(p) for p in ref), **kw)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ctx = lib.Context(0)
    bad = 0
    tot = {'rf_kept': 0, 'op_humans': 0}
    for _ in range(rounds):
        ok, info = rf_case(ctx, rng)
        tot['rf_kept'] += info['kept']
        if not ok:
            bad += 1
            print('FAIL retinaface', info)
        ok, info = op_case(ctx, rng)
        tot['op_humans'] += info.get('humans', 0)
        if not ok:
            bad += 1
            print('FAIL openpose', info)
    print('%d rounds, %d failures, %s' % (rounds, bad, tot))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())

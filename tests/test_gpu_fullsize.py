"""-m gpu: BASELINE.json's full-size configurations through size-independent properties
(the oracle is far too slow at these sizes): determinism, batch-split (shard) equivalence,
sortedness, NMS idempotence, unit norms, symmetric cosine matrix."""
import numpy as np
import pytest

from terran_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', params=['f32', 'bf16x3'])
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
    pose = OpenPose(device=0, short_side=368, state=states('openpose'), precision=precision)
    frames = synth.frames(3, 16, 368, 656)
    out = pose.call(frames)
    assert len(out) == 16
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
    frames = synth.frames(4, 4, 1080, 1920)
    det = Detection(device=0, state=states('retinaface'), precision=precision)
    rec = Recognition(device=0, state=states('arcface'), precision=precision)
    est = Estimation(device=0, state=states('openpose'), precision=precision)

    def run(fs):
        dets = det(fs)
        faces = [d[:2] for d in dets]
        return dets, rec(list(fs), faces), est(fs)
    d1, f1, p1 = run(frames)
    da, fa, pa = run(frames[:2])
    db, fb, pb = run(frames[2:])
    _same(d1, da + db)
    _same(p1, pa + pb)
    for x, y in zip(f1, fa + fb):
        assert np.array_equal(x, y)
    assert all(b['bbox'].dtype == np.int32 for d in d1 for b in d)

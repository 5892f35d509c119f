"""-m gpu: terran_amd.pipeline.StreamPipeline (one process, a device list, several batches in flight) against the
one-device facades: same detections, embeddings and poses, bit for bit, in batch and frame order ("8-way
output must equal 1-way output exactly"; two replicas on the one card here)."""
import numpy as np
import pytest

from terran_amd import synth

pytestmark = pytest.mark.gpu


def _pick(dets):
    return [[{'landmarks': x['landmarks']} for x in d[:2]] for d in dets]


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert len(x) == len(y)
        for p, q in zip(x, y):
            if isinstance(p, dict):
                for k in p:
                    assert np.array_equal(p[k], q[k]), k
            else:
                assert np.array_equal(p, q)


@pytest.mark.parametrize('devices,inflight,embed,shared', [([0, 0], 2, None, True), ([0], 1, None, True), ([0, 0, 0], 1, None, True),
                                                          ([0, 0], 2, 'f16', True), ([0, 0], 2, None, False), ([0], 3, None, True)])
def test_stream_pipeline_equals_one_device_facades(states, devices, inflight, embed, shared):
    """embed='f16': the embedder in its single-half mode (batch-invariant like every other mode: the comparison stays exact).
    shared: ONE embed worker per device launches on the faces of several batches at once (the default) / an embed thread per
    lane -- either way a face's embedding carries the bits the one-device facade gives it."""
    from terran_amd import Detection, Estimation, Recognition
    from terran_amd.pipeline import StreamPipeline
    sd_r, sd_a, sd_p = states('retinaface'), states('arcface'), states('openpose_decoder')
    kw = dict(detection_kw=dict(short_side=96, state=sd_r), recognition_kw=dict(state=sd_a, precision=embed),
              estimation_kw=dict(short_side=96, state=sd_p))
    sizes = [5, 4, 1, 2, 7, 3]                               # odd sizes, fewer frames than replicas
    batches = [synth.pose_code_frames(900 + 10 * i, n, 96, 128, 3) for i, n in enumerate(sizes)]
    pipe = StreamPipeline(devices, inflight=inflight, pick_faces=_pick, shared_embedder=shared, **kw)
    try:
        got = list(pipe.run(iter(batches)))
        if shared:                                           # the worker did merge shards: fewer launches than shards with faces
            launches, crops = pipe.embed_stats()
            assert crops == sum(len(f) for t in got for f in t[1]) and launches >= 1
            if len(devices) * inflight > 1:
                assert launches < sum(min(len(devices), n) for n in sizes), (launches, crops)
        # a second run on the same pipeline, with resident shards
        res = pipe.scatter(batches[0])
        again = list(pipe.run([res, res]))
        for fr in res:
            if fr is not None:
                fr.free()
    finally:
        pipe.close()
    det, rec, est = (Detection(short_side=96, device=0, state=sd_r), Recognition(device=0, state=sd_a, precision=embed),
                     Estimation(short_side=96, device=0, state=sd_p))
    assert len(got) == len(batches)
    n_det = n_pose = 0
    for (d, f, p), batch in zip(got, batches):
        rd = det(batch)
        rf = rec(batch, _pick(rd))
        rp = est(batch)
        _same(d, rd)
        assert len(f) == len(rf) == len(batch) and all(np.array_equal(x, y) for x, y in zip(f, rf))
        _same(p, rp)
        n_det += sum(len(x) for x in rd)
        n_pose += sum(len(x) for x in rp)
    assert n_det > 50 and n_pose >= 20
    for t in again:
        _same(t[0], got[0][0])
        _same(t[2], got[0][2])


def test_stream_pipeline_propagates_errors(states):
    from terran_amd.pipeline import StreamPipeline
    kw = dict(detection_kw=dict(short_side=96, state=states('retinaface')), recognition_kw=dict(state=states('arcface')),
              estimation_kw=dict(short_side=96, state=states('openpose_decoder')))
    pipe = StreamPipeline([0], inflight=1, pick_faces=_pick, **kw)
    try:
        with pytest.raises(Exception):
            list(pipe.run([np.zeros((2, 64, 64, 4), np.uint8)]))        # 4 channels: the upload asserts
    finally:
        pipe.close()


def test_stream_pipeline_without_faces(states):
    """Frames on which nothing is detected (threshold above every score) still yield their (empty) embeddings and poses."""
    from terran_amd.pipeline import StreamPipeline
    kw = dict(detection_kw=dict(short_side=64, state=states('retinaface')), recognition_kw=dict(state=states('arcface')),
              estimation_kw=dict(short_side=64, state=states('openpose')))
    pipe = StreamPipeline([0], inflight=2, pick_faces=lambda dets: [[] for _ in dets], **kw)
    try:
        out = list(pipe.run([synth.frames(5, 3, 64, 96), synth.frames(6, 2, 64, 96)]))
    finally:
        pipe.close()
    assert [len(t[0]) for t in out] == [3, 2]
    for dets, feats, poses in out:
        assert all(f.shape == (0, 512) for f in feats) and len(poses) == len(dets)


def test_stream_pipeline_abandoned_run_does_not_leak_into_the_next(states):
    """A video loop breaks out of the generator while batches are in flight; the next run() of the same pipeline must hand
    out ITS batches' results (keys restart at 0: without generation ids the stale results of the abandoned run would be
    attributed to the new batches)."""
    from terran_amd import Detection, Estimation, Recognition
    from terran_amd.pipeline import StreamPipeline
    sd_r, sd_a, sd_p = states('retinaface'), states('arcface'), states('openpose_decoder')
    kw = dict(detection_kw=dict(short_side=96, state=sd_r), recognition_kw=dict(state=sd_a),
              estimation_kw=dict(short_side=96, state=sd_p))
    first = [synth.pose_code_frames(700 + i, 3, 96, 128, 3) for i in range(8)]
    second = [synth.pose_code_frames(800 + i, 2, 96, 128, 2) for i in range(3)]
    pipe = StreamPipeline([0, 0], inflight=2, pick_faces=_pick, **kw)
    try:
        for k, _ in enumerate(pipe.run(iter(first))):
            if k == 1:
                break                                        # six batches still queued / in flight
        with pytest.raises(RuntimeError):                    # the generator above is closed: a new run is allowed, two at once are not
            g1, g2 = pipe.run(iter(second)), pipe.run(iter(second))
            next(g1)
            try:
                next(g2)
            finally:
                g1.close()
        got = list(pipe.run(iter(second)))
    finally:
        pipe.close()
    det, rec, est = (Detection(short_side=96, device=0, state=sd_r), Recognition(device=0, state=sd_a),
                     Estimation(short_side=96, device=0, state=sd_p))
    assert len(got) == len(second)
    for (d, f, p), batch in zip(got, second):
        rd = det(batch)
        _same(d, rd)
        rf = rec(batch, _pick(rd))
        assert all(np.array_equal(x, y) for x, y in zip(f, rf))
        _same(p, est(batch))


def test_stream_pipeline_is_dead_after_an_error(states):
    """After a lane error run() raises at once (it used to wait 600 s for results that could not come)."""
    import time
    from terran_amd.pipeline import StreamPipeline
    kw = dict(detection_kw=dict(short_side=96, state=states('retinaface')), recognition_kw=dict(state=states('arcface')),
              estimation_kw=dict(short_side=96, state=states('openpose_decoder')))
    pipe = StreamPipeline([0], inflight=2, pick_faces=_pick, **kw)
    try:
        with pytest.raises(Exception):
            list(pipe.run([np.zeros((2, 64, 64, 4), np.uint8)] * 6))    # more bad batches than the lane queues hold
        t0 = time.perf_counter()
        with pytest.raises(RuntimeError, match='dead'):
            list(pipe.run([synth.frames(5, 2, 96, 128)]))
        assert time.perf_counter() - t0 < 5.0
    finally:
        pipe.close()

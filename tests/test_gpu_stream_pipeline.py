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


@pytest.mark.parametrize('devices,inflight,embed', [([0, 0], 2, None), ([0], 1, None), ([0, 0, 0], 1, None), ([0, 0], 2, 'f16')])
def test_stream_pipeline_equals_one_device_facades(states, devices, inflight, embed):
    """embed='f16': the embedder in its single-half mode (batch-invariant like every other mode: the comparison stays exact)."""
    from terran_amd import Detection, Estimation, Recognition
    from terran_amd.pipeline import StreamPipeline
    sd_r, sd_a, sd_p = states('retinaface'), states('arcface'), states('openpose_decoder')
    kw = dict(detection_kw=dict(short_side=96, state=sd_r), recognition_kw=dict(state=sd_a, precision=embed),
              estimation_kw=dict(short_side=96, state=sd_p))
    sizes = [5, 4, 1, 2, 7, 3]                               # odd sizes, fewer frames than replicas
    batches = [synth.pose_code_frames(900 + 10 * i, n, 96, 128, 3) for i, n in enumerate(sizes)]
    pipe = StreamPipeline(devices, inflight=inflight, pick_faces=_pick, **kw)
    try:
        got = list(pipe.run(iter(batches)))
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

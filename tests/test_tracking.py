"""SORT tracking: oracle and host implementation against vectors produced by the reference's own
tracker (tests/golden/make_golden_tracking.py).  Host-only: no GPU involved."""
import numpy as np
import pytest

from tests.util import golden, tracking_scenario

N_SCEN = 4


def _run(sort, frames):
    counts, face_idx, track = [], [], []
    with np.errstate(all='ignore'):
        for faces in frames:
            res = sort.update(faces)
            counts.append(len(res))
            face_idx += [r['_i'] for r in res]
            track += [-1 if r['track'] is None else r['track'] for r in res]
    return np.array(counts), np.array(face_idx, np.int64), np.array(track, np.int64)


@pytest.mark.parametrize('si', range(N_SCEN))
def test_scenario_inputs_reproducible(si):
    g = golden('tracking.npz')
    frames = tracking_scenario(int(g['s%d_seed' % si]))
    assert np.array_equal(g['s%d_in_counts' % si], [len(f) for f in frames])
    assert np.array_equal(g['s%d_in_bbox' % si].reshape(-1, 4),
                          np.array([f['bbox'] for fr in frames for f in fr]).reshape(-1, 4))


@pytest.mark.parametrize('si', range(N_SCEN))
def test_oracle_matches_reference(si):
    from oracle import tracking as OT
    g = golden('tracking.npz')
    pre = 's%d_' % si
    ma, mh, ru = g[pre + 'cfg'].tolist()
    OT.reset_ids()
    sort = OT.Sort(max_age=ma, min_hits=mh, return_unmatched=bool(ru))
    c, f, t = _run(sort, tracking_scenario(int(g[pre + 'seed'])))
    assert np.array_equal(c, g[pre + 'out_counts'])
    assert np.array_equal(f, g[pre + 'out_face'])
    assert np.array_equal(t, g[pre + 'out_track'])
    assert np.array_equal([tr['id'] for tr in sort.tracks], g[pre + 'final_ids'])
    assert np.array_equal(np.array([tr['kf'].x[:, 0] for tr in sort.tracks]).reshape(-1, 7), g[pre + 'final_x'])
    assert np.array_equal(np.array([tr['kf'].P for tr in sort.tracks]).reshape(-1, 7, 7), g[pre + 'final_P'])


@pytest.mark.parametrize('si', range(N_SCEN))
def test_host_sort_matches_reference(si):
    from terran_amd import tracking as T
    g = golden('tracking.npz')
    pre = 's%d_' % si
    ma, mh, ru = g[pre + 'cfg'].tolist()
    T.reset_track_ids()
    sort = T.Sort(max_age=ma, min_hits=mh, return_unmatched=bool(ru))
    c, f, t = _run(sort, tracking_scenario(int(g[pre + 'seed'])))
    assert np.array_equal(c, g[pre + 'out_counts'])           # which faces come back, in which order ...
    assert np.array_equal(f, g[pre + 'out_face'])
    assert np.array_equal(t, g[pre + 'out_track'])            # ... and under which identity: exact
    assert np.array_equal(sort.ids, g[pre + 'final_ids'])
    assert np.array_equal(sort.hits, g[pre + 'final_hits'])
    np.testing.assert_allclose(sort.x, g[pre + 'final_x'], rtol=1e-9, atol=1e-9)       # float64 filter state
    np.testing.assert_allclose(sort.P, g[pre + 'final_P'], rtol=1e-9, atol=1e-9)


def test_ids_ascend_across_instances_and_empty_frames():
    from terran_amd import tracking as T
    T.reset_track_ids()
    face = {'bbox': np.array([10, 10, 60, 70], np.int32), 'score': np.float32(1)}
    a, b = T.Sort(min_hits=0), T.Sort(min_hits=0)
    assert [f['track'] for f in a.update([face])] == [0]
    assert [f['track'] for f in b.update([face])] == [1]      # KalmanTracker.count is process-wide (face.py:114)
    assert a.update([]) == [] and len(a) == 1                  # survives one missed frame (max_age=1)
    assert a.update([]) == [] and len(a) == 0


def test_facetracking_wrapper_and_factory():
    from terran_amd import tracking as T
    from terran_amd.facade import Detection
    T.reset_track_ids()
    face = {'bbox': np.array([10, 10, 60, 70], np.int32), 'score': np.float32(1)}
    calls = []

    def detector(frames):
        calls.append(len(frames))
        return [[dict(face)] for _ in frames]
    ft = T.FaceTracking(detector=detector, tracker=T.Sort(min_hits=0, max_age=3))
    batch = np.zeros((3, 8, 8, 3), np.uint8)
    out = ft(batch)
    assert [len(o) for o in out] == [1, 1, 1] and {o[0]['track'] for o in out} == {0}
    single = ft(batch[0])
    assert isinstance(single, list) and single[0]['track'] == 0 and calls == [3, 1]
    with pytest.raises(ValueError):
        T.face_tracking(detector=object())

    import io
    from terran_amd.video import RawVideoReader
    video = RawVideoReader(io.BytesIO(b''), 8, 8, batch_size=2, upload=lambda a: a, framerate=25)   # empty stream
    video.close()
    tr = T.face_tracking(video=video, detector=Detection(lazy=True))
    assert (tr.tracker.max_age, tr.tracker.min_hits) == (25, 5)
    tr = T.face_tracking(detector=Detection(lazy=True), min_hits=2)
    assert (tr.tracker.max_age, tr.tracker.min_hits) == (30, 2)

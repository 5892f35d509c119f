"""Generate tests/golden/tracking.npz FROM THE REFERENCE's tracker (terran/tracking/face.py).

CONTAINER-ONLY (needs /root/reference).  The reference file is imported unmodified; `filterpy` is absent from the
image, so `filterpy.kalman.KalmanFilter` is shimmed by oracle.tracking.KalmanFilter (filterpy's published linear
filter restated: PARITY UNPINNED for that one class).  Everything else -- IoU, state conversion, KalmanTracker
setup, Hungarian association through scipy, Sort's bookkeeping and id policy -- is the reference's own code.

    python tests/golden/make_golden_tracking.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from tests.golden import ref_import as R   # noqa: E402
from tests.util import tracking_scenario    # noqa: E402
from oracle import tracking as OT           # noqa: E402


def load_reference_tracking():
    R.install()
    det = sys.modules['terran.face.detection']
    det.Detection = type('Detection', (), {})
    det.face_detection = None
    fp = types.ModuleType('filterpy')
    fpk = types.ModuleType('filterpy.kalman')
    fpk.KalmanFilter = OT.KalmanFilter
    fp.kalman = fpk
    sys.modules['filterpy'] = fp
    sys.modules['filterpy.kalman'] = fpk
    spec = importlib.util.spec_from_file_location('terran_ref_tracking_face',
                                                  os.path.join(R.REF_ROOT, 'terran', 'tracking', 'face.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    T = load_reference_tracking()
    out = {}
    scen = [(11, dict(max_age=1, min_hits=3, return_unmatched=False)),
            (12, dict(max_age=5, min_hits=2, return_unmatched=True)),
            (13, dict(max_age=30, min_hits=6, return_unmatched=False)),
            (14, dict(max_age=2, min_hits=0, return_unmatched=False))]
    for si, (seed, kw) in enumerate(scen):
        frames = tracking_scenario(seed)
        T.KalmanTracker.count = 0
        sort = T.Sort(**kw)
        counts, face_idx, track = [], [], []
        with np.errstate(all='ignore'):
            for faces in frames:
                res = sort.update(faces)
                counts.append(len(res))
                for r in res:
                    face_idx.append(r['_i'])
                    track.append(-1 if r['track'] is None else r['track'])
        pre = 's%d_' % si
        out[pre + 'seed'] = seed
        out[pre + 'cfg'] = np.array([kw['max_age'], kw['min_hits'], int(kw['return_unmatched'])])
        out[pre + 'out_counts'] = np.array(counts, np.int64)
        out[pre + 'out_face'] = np.array(face_idx, np.int64)
        out[pre + 'out_track'] = np.array(track, np.int64)
        out[pre + 'in_counts'] = np.array([len(f) for f in frames], np.int64)
        out[pre + 'in_bbox'] = np.array([f['bbox'] for fr in frames for f in fr], np.int32).reshape(-1, 4)
        out[pre + 'final_ids'] = np.array([t.id for t in sort.trackers], np.int64)
        out[pre + 'final_x'] = np.array([t.kf.x[:, 0] for t in sort.trackers]).reshape(-1, 7)
        out[pre + 'final_P'] = np.array([t.kf.P for t in sort.trackers]).reshape(-1, 7, 7)
        out[pre + 'final_hits'] = np.array([t.hits for t in sort.trackers], np.int64)
        print('scenario %d: %d frames, %d faces in, %d out, %d ids, %d live trackers' % (
            si, len(frames), sum(len(f) for f in frames), sum(counts), T.KalmanTracker.count, len(sort.trackers)))
    out['note'] = ('reference terran/tracking/face.py Sort.update on tests.util.tracking_scenario(seed); '
                   'filterpy.kalman.KalmanFilter shimmed by oracle.tracking.KalmanFilter (parity unpinned for it)')
    path = os.path.join(HERE, 'tracking.npz')
    np.savez_compressed(path, **out)
    print('wrote tracking.npz %.1f KB' % (os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()

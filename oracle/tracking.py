"""ORACLE (test infrastructure only) -- CPU restatement of Terran's SORT face tracker.

What is restated (terran/tracking/face.py): box <-> filter-state conversion :47-95, the per-face constant-velocity
Kalman model :119-151 with its negative-area guard :196-197, IoU association through the Hungarian method with the
0.3 gate :206-272, and the bookkeeping of `Sort.update` :333-411 (prediction, removal of non-finite tracks, id policy
`hits >= min_hits or frame_count <= min_hits`, births, pruning by `max_age`, output order = surviving tracks in track
order followed by births).  Tracks are plain dicts here; the arithmetic and its order are the reference's.

Third-party arithmetic not under /root/reference: `filterpy.kalman.KalmanFilter` (setup.py:14, UNPINNED version, not
importable here).  `KalmanFilter` below restates filterpy's published linear filter (defaults x = 0 (n,1), P = Q = I,
R = I, F = I; predict x = F x, P = F P F' + Q; update y = z - H x, S = H P H' + R, K = P H' S^-1, x += K y,
P = (I - K H) P (I - K H)' + K R K').  PARITY UNPINNED for that class: the golden vectors (tests/golden/tracking.npz)
come from the reference's own tracker code imported in the build container with THIS class standing in for
filterpy's.  `scipy.optimize.linear_sum_assignment` is importable and used as the reference uses it.
"""
import numpy as np
from scipy.optimize import linear_sum_assignment

_next_id = [0]          # ids ascend over every tracker of the process (face.py:114,149-150)


def reset_ids(start=0):
    _next_id[0] = start


class KalmanFilter:
    """filterpy.kalman.KalmanFilter, the subset the tracker touches (also the filterpy stand-in of
    tests/golden/make_golden_tracking.py)."""

    def __init__(self, dim_x, dim_z):
        self.dim_x, self.dim_z = dim_x, dim_z
        self.x = np.zeros((dim_x, 1))
        self.P = np.eye(dim_x)
        self.Q = np.eye(dim_x)
        self.F = np.eye(dim_x)
        self.H = np.zeros((dim_z, dim_x))
        self.R = np.eye(dim_z)
        self._I = np.eye(dim_x)

    def predict(self):
        self.x = np.dot(self.F, self.x)
        self.P = np.dot(np.dot(self.F, self.P), self.F.T) + self.Q

    def update(self, z):
        z = np.asarray(z, dtype=float).reshape(self.dim_z, 1)
        y = z - np.dot(self.H, self.x)
        PHT = np.dot(self.P, self.H.T)
        S = np.dot(self.H, PHT) + self.R
        K = np.dot(PHT, np.linalg.inv(S))
        self.x = self.x + np.dot(K, y)
        I_KH = self._I - np.dot(K, self.H)
        self.P = np.dot(np.dot(I_KH, self.P), I_KH.T) + np.dot(np.dot(K, self.R), K.T)


def measurement(box):
    """corner box -> column (cx, cy, area, aspect); integer boxes divide as floats."""
    w, h = box[2] - box[0], box[3] - box[1]
    return np.array([box[0] + w / 2.0, box[1] + h / 2.0, w * h, w / h]).reshape(4, 1)


def state_box(x):
    """filter state (7,1) -> flat corner box (4,)."""
    w = np.sqrt(x[2] * x[3])
    h = x[2] / w
    return np.concatenate([x[0] - w / 2.0, x[1] - h / 2.0, x[0] + w / 2.0, x[1] + h / 2.0])


def new_track(face):
    kf = KalmanFilter(dim_x=7, dim_z=4)
    kf.F[:3, 4:] = np.eye(3)                       # position / area integrate their velocities; aspect has none
    kf.H[:, :4] = np.eye(4)
    kf.R[2:, 2:] *= 10.0
    kf.P[4:, 4:] *= 1000.0
    kf.P *= 10.0
    kf.Q[-1, -1] *= 0.01
    kf.Q[4:, 4:] *= 0.01
    kf.x[:4] = measurement(face['bbox'])
    track = {'id': _next_id[0], 'kf': kf, 'hits': 0, 'misses': 0}
    _next_id[0] += 1
    return track


def advance(track):
    """One prediction step; returns the predicted corner box."""
    kf = track['kf']
    if (kf.x[6] + kf.x[2]) <= 0:
        kf.x[6] *= 0.0
    kf.predict()
    track['misses'] += 1
    return state_box(kf.x)


def overlap(a, b):
    iw = np.maximum(0.0, np.minimum(a[2], b[2]) - np.maximum(a[0], b[0]))
    ih = np.maximum(0.0, np.minimum(a[3], b[3]) - np.maximum(a[1], b[1]))
    inter = iw * ih
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def assign(faces, boxes, gate=0.3):
    """-> {track index: face index} of accepted pairs, and the face indices left over (reference order: never
    assigned first, then gated-out pairs in assignment order)."""
    if len(boxes) == 0:
        return {}, list(range(len(faces)))
    cost = np.zeros((len(faces), len(boxes)), dtype=np.float32)
    for i, f in enumerate(faces):
        for j, b in enumerate(boxes):
            cost[i, j] = overlap(f['bbox'], b)
    rows, cols = linear_sum_assignment(-cost)
    pairs, spare = {}, [i for i in range(len(faces)) if i not in set(rows.tolist())]
    for i, j in zip(rows.tolist(), cols.tolist()):
        if cost[i, j] < gate:
            spare.append(i)
        else:
            pairs[j] = i
    return pairs, spare


class Sort:
    def __init__(self, max_age=1, min_hits=3, return_unmatched=False):
        self.max_age, self.min_hits, self.return_unmatched = max_age, min_hits, return_unmatched
        self.tracks = []
        self.frame_count = 0

    def update(self, faces):
        self.frame_count += 1
        boxes = [advance(t) for t in self.tracks]
        alive = [k for k, b in enumerate(boxes) if not np.any(np.isnan(b))]
        self.tracks = [self.tracks[k] for k in alive]
        boxes = [boxes[k] for k in alive]
        pairs, spare = assign(faces, boxes)
        out = []
        for j, track in enumerate(self.tracks):
            if j not in pairs:
                continue
            face = faces[pairs[j]]
            track['misses'] = 0
            track['hits'] += 1
            track['kf'].update(measurement(face['bbox']))
            confirmed = track['hits'] >= self.min_hits or self.frame_count <= self.min_hits
            out.append({'track': track['id'] if confirmed else None, **face})
        for i in spare:
            track = new_track(faces[i])
            self.tracks.append(track)
            out.append({'track': track['id'] if self.min_hits == 0 else None, **faces[i]})
        if not self.return_unmatched:
            out = [f for f in out if f['track'] is not None]
        self.tracks = [t for t in self.tracks if t['misses'] <= self.max_age]
        return out

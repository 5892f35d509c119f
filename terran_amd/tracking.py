"""SORT face tracking over `Detection` output (the reference's terran/tracking/face.py), host side.

Tracking is sequential per stream and tiny next to the networks (a 7-state constant-velocity Kalman filter per face,
one Hungarian assignment per frame), so it stays on the host like the reference's -- but laid out as struct-of-arrays
over all live tracks (states (T,7), covariances (T,7,7)) with the filter written for this model's F and H instead of
one `filterpy.KalmanFilter` object per face:

  F = I + shift(4)   : x <- x F',  P <- F P F' + Q   (every sum has at most two non-zero terms: order-independent)
  H = [I4 | 0]       : S = P[:4,:4] + R,  K = P[:, :4] S^-1,  Joseph-form covariance update

Contract kept from the reference (file:line = terran/tracking/face.py):
  * `Sort(max_age, min_hits, return_unmatched).update(faces)` returns the same face dicts with a 'track' key, matched
    tracks first (in track order), then new tracks; id policy of :389-404; pruning by `max_age` :409-411.
  * IoU matrix in float32 from float64 arithmetic, assignment on its negation, matches below 0.3 IoU split back into
    unmatched face + unmatched track (:229-266).  `scipy.optimize.linear_sum_assignment`, as the reference (:4).
  * track ids ascend across ALL Sort instances of the process (`KalmanTracker.count`, :114,149-150).
  * a track whose predicted box is not finite is dropped before association (:372-381).
Deviations (the reference code cannot run these paths): `FaceTracking.__call__` on a single image wraps it into a
batch (the reference indexes `frames[0]`, :459-461, which drops the image); `face_tracking()` builds `Sort` from the
resolved `max_age` / `min_hits` (the reference reads `video.framerate` unconditionally, :548-551, and so raises
without a video).
"""
import itertools

import numpy as np

_ids = itertools.count()

_R = np.diag([1.0, 1.0, 10.0, 10.0])                                  # :139
_P0 = np.diag([10.0] * 4 + [10000.0] * 3)                             # :140-141
_Q = np.diag([1.0, 1.0, 1.0, 1.0, 0.01, 0.01, 0.0001])                # :142-143
_I7 = np.eye(7)
_H = np.eye(4, 7)
_F = np.eye(7)
_F[0, 4] = _F[1, 5] = _F[2, 6] = 1.0                                  # :126-134
_FT = _F.T.copy()


def reset_track_ids(start=0):
    """Restart the process-wide id counter (tests / new video)."""
    global _ids
    _ids = itertools.count(start)


def boxes_to_state(bbox):
    """(n,4) corner boxes -> (n,4) [cx, cy, area, ratio] (:47-71); integer inputs divide as floats."""
    b = np.asarray(bbox)
    w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    with np.errstate(all='ignore'):
        return np.stack([b[:, 0] + w / 2.0, b[:, 1] + h / 2.0, (w * h).astype(np.float64), w / h], axis=1)


def state_to_boxes(x):
    """(T,>=4) states -> (T,4) corner boxes (:74-95)."""
    with np.errstate(all='ignore'):
        w = np.sqrt(x[:, 2] * x[:, 3])
        h = x[:, 2] / w
    return np.stack([x[:, 0] - w / 2.0, x[:, 1] - h / 2.0, x[:, 0] + w / 2.0, x[:, 1] + h / 2.0], axis=1)


def iou_matrix(face_boxes, track_boxes):
    """float32 (n_faces, n_tracks) IoU, computed like :14-44 (face area in the faces' own dtype)."""
    f = np.asarray(face_boxes)
    t = np.asarray(track_boxes, dtype=np.float64)
    with np.errstate(all='ignore'):
        iw = np.maximum(0.0, np.minimum(f[:, None, 2], t[None, :, 2]) - np.maximum(f[:, None, 0], t[None, :, 0]))
        ih = np.maximum(0.0, np.minimum(f[:, None, 3], t[None, :, 3]) - np.maximum(f[:, None, 1], t[None, :, 1]))
        inter = iw * ih
        fa = (f[:, 2] - f[:, 0]) * (f[:, 3] - f[:, 1])
        ta = (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
        return (inter / (fa[:, None] + ta[None, :] - inter)).astype(np.float32)


def associate(face_boxes, track_boxes, iou_threshold=0.3):
    """-> matches (K,2) [face, track], unmatched face indices (reference order), unmatched track indices (:206-272)."""
    from scipy.optimize import linear_sum_assignment
    n_f, n_t = len(face_boxes), len(track_boxes)
    if n_t == 0:
        return np.empty((0, 2), dtype=int), list(range(n_f)), []
    m = iou_matrix(face_boxes, track_boxes) if n_f else np.zeros((0, n_t), np.float32)
    rows, cols = linear_sum_assignment(-m)
    unmatched_f = sorted(set(range(n_f)) - set(rows.tolist()))
    unmatched_t = sorted(set(range(n_t)) - set(cols.tolist()))
    matches = []
    for fi, ti in zip(rows.tolist(), cols.tolist()):
        if m[fi, ti] < iou_threshold:
            unmatched_f.append(fi)
            unmatched_t.append(ti)
        else:
            matches.append((fi, ti))
    return np.array(matches, dtype=int).reshape(-1, 2), unmatched_f, unmatched_t


class Sort:
    """Appearance-agnostic tracking-by-detection (:275-411): attaches an identity to each detection or filters it."""

    def __init__(self, max_age=1, min_hits=3, return_unmatched=False):
        self.max_age, self.min_hits, self.return_unmatched = max_age, min_hits, return_unmatched
        self.frame_count = 0
        self.x = np.zeros((0, 7))
        self.P = np.zeros((0, 7, 7))
        self.hits = np.zeros(0, np.int64)
        self.time_since_update = np.zeros(0, np.int64)
        self.ids = np.zeros(0, np.int64)

    def __len__(self):
        return len(self.ids)

    def _keep(self, mask):
        self.x, self.P = self.x[mask], self.P[mask]
        self.hits, self.time_since_update, self.ids = self.hits[mask], self.time_since_update[mask], self.ids[mask]

    def _predict(self):
        x, P = self.x, self.P
        with np.errstate(all='ignore'):
            x[(x[:, 6] + x[:, 2]) <= 0, 6] = 0.0                                      # :196-197
            x = x @ _FT                                                                # keeps 0*inf -> nan of F.x
            P = _F @ P @ _FT + _Q
        self.x, self.P = x, P
        self.time_since_update += 1
        return state_to_boxes(x)

    def _correct(self, rows, z):
        """Kalman update of tracks `rows` with measurements z (k,4)."""
        x, P = self.x[rows], self.P[rows]
        y = z - x[:, :4]
        PHT = P[:, :, :4]
        S = P[:, :4, :4] + _R
        K = PHT @ np.linalg.inv(S)
        x = x + (K @ y[:, :, None])[:, :, 0]
        I_KH = _I7 - K @ _H
        P = I_KH @ P @ np.swapaxes(I_KH, 1, 2) + K @ _R @ np.swapaxes(K, 1, 2)
        self.x[rows], self.P[rows] = x, P
        self.time_since_update[rows] = 0
        self.hits[rows] += 1

    def update(self, faces):
        """Call once per frame (also with no faces).  Returns the face dicts with 'track' (:333-411)."""
        self.frame_count += 1
        boxes = self._predict() if len(self) else np.zeros((0, 4))
        ok = np.isfinite(boxes).all(axis=1)
        if not ok.all():
            self._keep(ok)
            boxes = boxes[ok]
        face_boxes = np.stack([np.asarray(f['bbox']) for f in faces]) if len(faces) else np.zeros((0, 4))
        matches, unmatched_f, _ = associate(face_boxes, boxes)
        out = []
        if len(matches):
            order = np.argsort(matches[:, 1], kind='stable')                           # matched tracks in track order
            matches = matches[order]
            with np.errstate(all='ignore'):
                self._correct(matches[:, 1], boxes_to_state(face_boxes[matches[:, 0]]))
            for fi, ti in matches.tolist():
                confirmed = self.hits[ti] >= self.min_hits or self.frame_count <= self.min_hits
                out.append({'track': int(self.ids[ti]) if confirmed else None, **faces[fi]})
        if unmatched_f:
            n = len(unmatched_f)
            x0 = np.zeros((n, 7))
            x0[:, :4] = boxes_to_state(face_boxes[unmatched_f])
            new_ids = np.array([next(_ids) for _ in range(n)], np.int64)
            self.x = np.concatenate([self.x, x0])
            self.P = np.concatenate([self.P, np.broadcast_to(_P0, (n, 7, 7))])
            self.hits = np.concatenate([self.hits, np.zeros(n, np.int64)])
            self.time_since_update = np.concatenate([self.time_since_update, np.zeros(n, np.int64)])
            self.ids = np.concatenate([self.ids, new_ids])
            for fi, tid in zip(unmatched_f, new_ids.tolist()):
                out.append({'track': tid if self.min_hits == 0 else None, **faces[fi]})
        if not self.return_unmatched:
            out = [f for f in out if f['track'] is not None]
        self._keep(self.time_since_update <= self.max_age)
        return out


class FaceTracking:
    """Drop-in for a `Detection` object that adds a 'track' field to every face (:414-473)."""

    def __init__(self, detector=None, tracker=None):
        self.detector = detector
        self.tracker = tracker

    def __call__(self, frames):
        single = not isinstance(frames, (list, tuple)) and getattr(frames, 'ndim', 4) == 3
        batch = [frames] if single else frames
        per_frame = [self.tracker.update(d) for d in self.detector(batch)]
        return per_frame[0] if single else per_frame


def face_tracking(*, video=None, max_age=None, min_hits=None, detector=None, return_unmatched=False):
    """Factory with the reference's defaults (:476-554): one second of frames for `max_age`, a fifth of a second for
    `min_hits` when a video (anything with `.framerate`) is given, else 30 / 6."""
    import terran_amd
    from . import facade
    max_age_, min_hits_ = 30, 6
    if video is not None:
        max_age_, min_hits_ = video.framerate, video.framerate // 5
    max_age = max_age_ if max_age is None else max_age
    min_hits = min_hits_ if min_hits is None else min_hits
    if detector is None:
        detector = terran_amd.face_detection
    elif not isinstance(detector, facade.Detection):
        raise ValueError('`detector` must be an instance of `terran.face.Detection`.')
    return FaceTracking(detector=detector, tracker=Sort(max_age=max_age, min_hits=min_hits,
                                                        return_unmatched=return_unmatched))

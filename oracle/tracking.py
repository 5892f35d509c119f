"""ORACLE (test infrastructure only) -- CPU restatement of Terran's SORT face tracker.

Follows terran/tracking/face.py: `iou` :14-44, `corners_to_center` :47-71, `center_to_corners` :74-95,
`KalmanTracker` :98-203, `associate_detections_to_trackers` :206-272, `Sort` :275-411,
`FaceTracking.__call__` :429-473.

Third-party arithmetic not under /root/reference: `filterpy.kalman.KalmanFilter` (setup.py:14, UNPINNED version,
not importable here).  `KalmanFilter` below restates filterpy's published linear filter (defaults x = 0 (n,1),
P = I, Q = I, R = I, F = I; predict x = F x, P = F P F' + Q; update y = z - H x, S = H P H' + R, K = P H' S^-1,
x += K y, P = (I - K H) P (I - K H)' + K R K').  PARITY UNPINNED for that class: the golden vectors
(tests/golden/tracking.npz) come from the reference's own Sort / KalmanTracker / association code imported in
the build container with THIS class standing in for filterpy's.  `scipy.optimize.linear_sum_assignment` is
importable and used as the reference uses it.
"""
import numpy as np
from scipy.optimize import linear_sum_assignment


class KalmanFilter:
    """filterpy.kalman.KalmanFilter, the subset the tracker touches (face.py:125-151,166,185)."""

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


def iou(a, b):                                                   # face.py:14-44
    x_min, y_min = np.maximum(a[0], b[0]), np.maximum(a[1], b[1])
    x_max, y_max = np.minimum(a[2], b[2]), np.minimum(a[3], b[3])
    inter = np.maximum(0.0, x_max - x_min) * np.maximum(0.0, y_max - y_min)
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def corners_to_center(bbox):                                     # face.py:47-71
    w, h = bbox[2] - bbox[0], bbox[3] - bbox[1]
    return np.array([bbox[0] + w / 2.0, bbox[1] + h / 2.0, w * h, w / h]).reshape((4, 1))


def center_to_corners(s):                                        # face.py:74-95
    w = np.sqrt(s[2] * s[3])
    h = s[2] / w
    return np.concatenate([s[0] - w / 2.0, s[1] - h / 2.0, s[0] + w / 2.0, s[1] + h / 2.0])


class KalmanTracker:                                             # face.py:98-203
    count = 0

    def __init__(self, face):
        kf = self.kf = KalmanFilter(dim_x=7, dim_z=4)
        kf.F = np.eye(7)
        kf.F[0, 4] = kf.F[1, 5] = kf.F[2, 6] = 1.0
        kf.H = np.eye(4, 7)
        kf.R[2:, 2:] *= 10.0
        kf.P[4:, 4:] *= 1000.0
        kf.P *= 10.0
        kf.Q[-1, -1] *= 0.01
        kf.Q[4:, 4:] *= 0.01
        kf.x[:4] = corners_to_center(face['bbox'])
        self.hits = 0
        self.time_since_update = 0
        self.id = KalmanTracker.count
        KalmanTracker.count += 1

    def update(self, face):
        self.time_since_update = 0
        self.hits += 1
        self.kf.update(corners_to_center(face['bbox']))

    def predict(self):
        if (self.kf.x[6] + self.kf.x[2]) <= 0:
            self.kf.x[6] *= 0.0
        self.kf.predict()
        self.time_since_update += 1
        return center_to_corners(self.kf.x)


def associate(faces, trackers, iou_threshold=0.3):               # face.py:206-272
    if not len(trackers):
        return np.empty((0, 2), dtype=int), np.arange(len(faces)), np.empty((0, 5), dtype=int)
    m = np.zeros((len(faces), len(trackers)), dtype=np.float32)
    for fi, face in enumerate(faces):
        for ti, track in enumerate(trackers):
            m[fi, ti] = iou(face['bbox'], track)
    pairs = np.transpose(np.asarray(linear_sum_assignment(-m)))
    unmatched_faces = [fi for fi in range(len(faces)) if fi not in pairs[:, 0]]
    unmatched_trackers = [ti for ti in range(len(trackers)) if ti not in pairs[:, 1]]
    matches = []
    for fi, ti in pairs:
        if m[fi, ti] < iou_threshold:
            unmatched_faces.append(fi)
            unmatched_trackers.append(ti)
        else:
            matches.append(np.array([fi, ti], dtype=int))
    matches = np.stack(matches) if matches else np.empty((0, 2), dtype=int)
    return matches, np.array(unmatched_faces), np.array(unmatched_trackers)


class Sort:                                                      # face.py:275-411
    def __init__(self, max_age=1, min_hits=3, return_unmatched=False):
        self.max_age, self.min_hits, self.return_unmatched = max_age, min_hits, return_unmatched
        self.trackers = []
        self.frame_count = 0

    def update(self, faces):
        self.frame_count += 1
        to_delete = []
        tracks = np.zeros((len(self.trackers), 4))
        for ti, track in enumerate(tracks):
            pos = self.trackers[ti].predict()
            track[:] = pos
            if np.any(np.isnan(pos)):
                to_delete.append(ti)
        tracks = np.ma.compress_rows(np.ma.masked_invalid(tracks))
        for t in reversed(to_delete):
            self.trackers.pop(t)
        matched, unmatched_faces, unmatched_tracks = associate(faces, tracks)
        out = []
        for ti, track in enumerate(self.trackers):
            if ti not in unmatched_tracks:
                fi = int(matched[np.where(matched[:, 1] == ti)[0], 0][0])
                track.update(faces[fi])
                tid = track.id if (track.hits >= self.min_hits or self.frame_count <= self.min_hits) else None
                out.append({'track': tid, **faces[fi]})
        for fi in unmatched_faces:
            track = KalmanTracker(faces[fi])
            self.trackers.append(track)
            out.append({'track': track.id if self.min_hits == 0 else None, **faces[fi]})
        if not self.return_unmatched:
            out = [f for f in out if f['track'] is not None]
        self.trackers = [t for t in self.trackers if t.time_since_update <= self.max_age]
        return out

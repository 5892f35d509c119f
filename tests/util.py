import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def unflatten(counts, *arrays):
    out, o = [], 0
    for c in counts:
        out.append([tuple(a[o + i] for a in arrays) for i in range(int(c))])
        o += int(c)
    return out


def tracking_scenario(seed, n_frames=60):
    """Seeded detection sequences for the SORT tests: a handful of faces moving with constant velocity plus jitter,
    born and retired at random frames, randomly missed, occasionally crossing; a few degenerate boxes (zero height ->
    NaN state -> tracker deletion, face.py:374-381) and fast-shrinking boxes (negative area guard, face.py:196-197).
    Each face dict carries '_i' (its index within the frame) so outputs can be traced back to inputs."""
    rng = np.random.default_rng(seed)
    n_obj = int(rng.integers(3, 7))
    objs = []
    for _ in range(n_obj):
        birth = int(rng.integers(0, n_frames // 2))
        death = int(rng.integers(birth + 5, n_frames + 10))
        c = rng.uniform(100, 900, 2)
        v = rng.uniform(-12, 12, 2)
        wh = rng.uniform(40, 160, 2)
        shrink = rng.uniform(0.93, 1.03)
        objs.append((birth, death, c, v, wh, shrink))
    frames = []
    for t in range(n_frames):
        faces = []
        for (birth, death, c, v, wh, shrink) in objs:
            if not (birth <= t < death) or rng.random() < 0.12:
                continue
            k = t - birth
            cc = c + v * k + rng.normal(0, 1.5, 2)
            w, h = wh * (shrink ** k) + rng.normal(0, 1.0, 2)
            w, h = max(w, 2.0), max(h, 2.0)
            if rng.random() < 0.02:
                h = 0.0                                             # degenerate detection
            bbox = np.around([cc[0] - w / 2, cc[1] - h / 2, cc[0] + w / 2, cc[1] + h / 2]).astype(np.int32)
            faces.append({'bbox': bbox, 'landmarks': np.zeros((5, 2), np.int32), 'score': np.float32(0.9)})
        if rng.random() < 0.15:                                     # a spurious one-frame detection
            x, y = rng.uniform(0, 1000, 2)
            faces.append({'bbox': np.array([x, y, x + 50, y + 60]).astype(np.int32),
                          'landmarks': np.zeros((5, 2), np.int32), 'score': np.float32(0.6)})
        order = rng.permutation(len(faces))
        faces = [faces[i] for i in order]
        for i, f in enumerate(faces):
            f['_i'] = i
        frames.append(faces)
    return frames

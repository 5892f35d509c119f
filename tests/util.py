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

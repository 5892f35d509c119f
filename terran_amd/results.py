"""The reference's list-of-dicts return type over the packed result arrays of the C ABI.

`detections(counts, boxes, landmarks, scores)` -> list[N] of list[{'bbox', 'landmarks', 'score'}]: row VIEWS into the
arrays and numpy float32 scalars, exactly what the comprehension below yields (retinaface/wrapper.py:228-236).  The C
module `_pyresults` (csrc/pyresults.c, built by terran_amd.build) does the same ~1.5x faster with the GIL held for that
much less; without it the comprehension runs.  Host glue only: no arithmetic.
"""
try:
    from . import _pyresults
except ImportError:                                   # not built (gcc / Python headers missing): same objects, slower
    _pyresults = None


def detections_py(counts, boxes, landmarks, scores):
    out, o = [], 0
    for c in counts:
        c = int(c)
        out.append([{'bbox': b, 'landmarks': l, 'score': s}
                    for b, l, s in zip(boxes[o:o + c], landmarks[o:o + c], scores[o:o + c])])
        o += c
    return out


def detections(counts, boxes, landmarks, scores):
    if _pyresults is not None:
        return _pyresults.detections(counts, boxes, landmarks, scores)
    return detections_py(counts, boxes, landmarks, scores)

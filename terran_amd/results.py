"""The reference's list-of-dicts return type over the packed result arrays of the C ABI.

`detections(counts, boxes, landmarks, scores)` -> list[N] of list[{'bbox', 'landmarks', 'score'}]: row VIEWS into the
arrays and numpy float32 scalars, exactly what `detections_py` below yields (retinaface/wrapper.py:228-236).

By DEFAULT the per-image lists are plain `list`s of dicts, built at once (through the C module `_pyresults` when it is built):
the reference returns plain lists, and a drop-in must survive everything callers do to them -- including C code that reads a
list's storage directly (CPython's own `list + x` / `sum(dets, [])` with a plain list on the left, ujson, Cython `list`
arguments), which no subclass method can intercept.

OPT-IN (`detections(..., lazy=True)`, TERRAN_AMD_LAZY_RESULTS=1, the wrappers' `lazy_results=True`): the per-image lists are
`LazyFaces`: a `list` subclass that creates its dicts the first time anything looks at them.  A
detector call on a video batch returns thousands of detections (~11 000 per 32 1080p frames of noise through random
weights); one dict + two array views + one numpy scalar each is ~150 ns through the C API (`_pyresults`, csrc/pyresults.c)
and ~350 ns as a comprehension -- 0.7 - 2 ms per call with the GIL held, as long as the network itself takes on the device
-- and most callers read a few faces per image (the pipeline's `pick_faces` reads the top F).  Until then a LazyFaces
holds three array slices.  `len()` needs no dicts; indexing, slicing, iteration, comparison, mutation, pickling, `repr`
and anything that goes through the sequence protocol (`list(x)`, `sorted`, `json.dumps`, `numpy.array`) fill the list first
and then behave as the plain list they are; `plain_list + lazy` and `sum(lazy_lists, [])` go through `__radd__` (a subclass's
reflected slot is tried before list's own concatenation).  What stays out of reach is C code that takes the object for a
plain list and reads `ob_item` without calling anything: it sees an empty list -- hence opt-in, for callers that know
their consumers (the pipeline's `pick_faces` reads the top F faces of thousands).  `eager(x)` builds plain lists.
Host glue only: no arithmetic.
"""
import os
import threading

import numpy as _np

_fill_lock = threading.Lock()

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


def detections_eager(counts, boxes, landmarks, scores):
    if _pyresults is not None:
        return _pyresults.detections(counts, boxes, landmarks, scores)
    return detections_py(counts, boxes, landmarks, scores)


class LazyFaces(list):
    """One image's detections.  Empty storage + the three result slices until first use (see the module text)."""
    __slots__ = ('_src',)

    def __init__(self, boxes, landmarks, scores):
        list.__init__(self)
        self._src = (boxes, landmarks, scores)

    def _fill(self):
        if self._src is not None:
            with _fill_lock:                            # two threads looking at one unfilled list: one of them builds it
                src = self._src
                if src is not None:
                    b, l, s = src
                    if len(s):
                        list.extend(self, detections_eager(_np.array([len(s)], _np.int32), b, l, s)[0])
                    self._src = None                    # only now: a reader that sees None sees the items
        return self

    def __len__(self):
        src = self._src
        return len(src[2]) if src is not None else list.__len__(self)

    def __reduce_ex__(self, protocol):                  # pickles / copies as the plain list it stands for
        return (list, (list(self._fill()),))

    def __repr__(self):
        return list.__repr__(self._fill())

    __hash__ = None


def _filled(name):
    base = getattr(list, name)

    def method(self, *a, **k):
        # list's own C code reads another list's items directly (concatenation, comparison): fill those as well
        a = tuple(x._fill() if isinstance(x, LazyFaces) else x for x in a)
        return base(self._fill(), *a, **k)
    method.__name__ = name
    method.__doc__ = base.__doc__
    return method


def _radd(self, other):
    # `plain_list + lazy`, `sum(lazies, [])`: list's sq_concat would copy this object's (empty) storage
    return list.__add__(other._fill() if isinstance(other, LazyFaces) else list(other), self._fill())


LazyFaces.__radd__ = _radd

for _name in ('__getitem__', '__setitem__', '__delitem__', '__iter__', '__reversed__', '__contains__', '__eq__', '__ne__',
              '__lt__', '__le__', '__gt__', '__ge__', '__add__', '__iadd__', '__mul__', '__rmul__', '__imul__', 'append',
              'extend', 'insert', 'pop', 'remove', 'clear', 'index', 'count', 'sort', 'reverse', 'copy'):
    setattr(LazyFaces, _name, _filled(_name))
del _name


def eager(dets):
    """Plain lists of dicts out of whatever `detections` returned."""
    return [list(d) for d in dets]


def detections(counts, boxes, landmarks, scores, lazy=None):
    if lazy is None:
        lazy = bool(os.environ.get('TERRAN_AMD_LAZY_RESULTS')) and not os.environ.get('TERRAN_AMD_EAGER_RESULTS')
    if not lazy:
        return detections_eager(counts, boxes, landmarks, scores)
    out, o = [], 0
    for c in (counts.tolist() if hasattr(counts, 'tolist') else counts):
        c = int(c)
        out.append(LazyFaces(boxes[o:o + c], landmarks[o:o + c], scores[o:o + c]))
        o += c
    return out

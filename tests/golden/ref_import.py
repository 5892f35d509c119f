"""Import the *reference* (terran) leaf modules in the survey/build container.

CONTAINER-ONLY TOOLING.  `/root/reference` does not exist on the GPU box, so
nothing under `tests/` (other than the fixture generator `make_golden.py`) and
nothing in `bench.py` / `__graft_entry__.py` may import this module.

The reference package cannot be imported as-is (its `__init__` files import
cv2 / torchvision / skimage, which are absent here).  Following SURVEY.md
Appendix B we register empty parent packages whose `__path__` points into the
reference tree, plus three tiny third-party shims (the "parity unpinned"
surfaces of SURVEY.md §8c), and then import the real leaf files.

The shims delegate to *this repo's oracle restatements*:
  cv2.resize            -> oracle.facade.cv2_resize_linear
  torchvision.ops.nms   -> oracle.retinaface_post.nms
  skimage SimilarityTransform.estimate -> oracle.arcface_pre.umeyama
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = os.environ.get('TERRAN_REFERENCE', '/root/reference')
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if REPO_ROOT not in sys.path:
    sys.path.insert(0, REPO_ROOT)


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'terran'))


def _stub(name, subdir):
    mod = types.ModuleType(name)
    mod.__path__ = [os.path.join(REF_ROOT, 'terran', subdir) if subdir else os.path.join(REF_ROOT, 'terran')]
    sys.modules[name] = mod
    return mod


_installed = False


def install():
    """Install parent-package stubs + third-party shims (idempotent)."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError('reference tree not present at %s' % REF_ROOT)

    terran = _stub('terran', '')
    terran.default_device = torch.device('cpu')
    _stub('terran.face', 'face')
    _stub('terran.face.detection', 'face/detection')
    _stub('terran.face.detection.retinaface', 'face/detection/retinaface')
    _stub('terran.face.recognition', 'face/recognition')
    _stub('terran.face.recognition.arcface', 'face/recognition/arcface')
    _stub('terran.pose', 'pose')
    _stub('terran.pose.openpose', 'pose/openpose')

    # --- third-party shims (our own restatements; parity unpinned surfaces) ---
    from oracle import facade as o_facade
    from oracle import retinaface_post as o_rpost
    from oracle import arcface_pre as o_apre

    cv2 = types.ModuleType('cv2')
    cv2.INTER_LINEAR = 1

    def _resize(src, dsize, dst=None, interpolation=None):
        out = o_facade.cv2_resize_linear(np.asarray(src), dsize)
        if dst is not None:
            dst[...] = out
            return dst
        return out
    cv2.resize = _resize
    sys.modules['cv2'] = cv2

    tv = types.ModuleType('torchvision')
    tv_ops = types.ModuleType('torchvision.ops')

    def _nms(boxes, scores, thr):
        keep = o_rpost.nms(boxes.cpu().numpy(), float(thr))
        return torch.as_tensor(keep, dtype=torch.long)
    tv_ops.nms = _nms
    tv.ops = tv_ops
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.ops'] = tv_ops

    sk = types.ModuleType('skimage')
    sk_t = types.ModuleType('skimage.transform')

    class SimilarityTransform:
        def __init__(self):
            self.params = np.eye(3)

        def estimate(self, src, dst):
            self.params = o_apre.umeyama(np.asarray(src), np.asarray(dst))
            return True
    sk_t.SimilarityTransform = SimilarityTransform
    sk.transform = sk_t
    sys.modules['skimage'] = sk
    sys.modules['skimage.transform'] = sk_t
    _installed = True


def ref(modname):
    """Import a reference leaf module, e.g. ref('terran.pose.openpose.wrapper')."""
    install()
    return importlib.import_module(modname)

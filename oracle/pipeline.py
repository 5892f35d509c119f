"""ORACLE (test infrastructure only): the whole reference path on the CPU --
wrapper `.call` methods and the three task facades -- assembled from the
restatements in this package.  This is what `bench.py` times as `cpu_baseline`
(kind "port") and what the `-m gpu` parity tests compare the HIP path against.
Never imported by the product path (`terran_amd/`).

Follows:
  retinaface_call : terran/face/detection/retinaface/wrapper.py:133-238
  arcface_call    : terran/face/recognition/arcface/wrapper.py:109-184
  openpose_call   : terran/pose/openpose/wrapper.py:182-485
  detection       : terran/face/detection/__init__.py:234-287
  recognition     : terran/face/recognition/__init__.py:34-90
  estimation      : terran/pose/__init__.py:182-223
"""
import numpy as np
import torch

from . import nets, retinaface_post, arcface_pre, openpose_post, facade


def retinaface_call(sd, images, threshold=0.5, nms_threshold=0.4):
    """images (N,H,W,3) uint8 RGB -> list[N] of list[dict] (network-input pixels)."""
    H, W = images.shape[1:3]
    x = torch.from_numpy(np.ascontiguousarray(images)).to(torch.float32).permute(0, 3, 1, 2).flip(1).contiguous()
    outs = [o.numpy() for o in nets.retinaface_forward(sd, x)]
    return retinaface_post.postprocess(outs, H, W, threshold, nms_threshold)


def arcface_call(sd, images, faces_per_image=None):
    pre = []
    if faces_per_image is not None:
        for image, faces in zip(images, faces_per_image):
            for face in faces:
                pre.append(arcface_pre.preprocess_face(image, face['landmarks']))
        splits = np.cumsum([len(f) for f in faces_per_image])[:-1]
    else:
        for image in images:
            pre.append(arcface_pre.preprocess_face_no_landmarks(image))
        splits = []
    if not pre:
        return [np.empty((0, 512)) for _ in images]          # float64, wrapper.py:160-164
    x = torch.from_numpy(np.stack(pre, 0).astype(np.float32))
    feats = arcface_pre.l2_normalize(nets.arcface_forward(sd, x).numpy())
    out = np.split(feats, splits, axis=0)
    return out[0] if faces_per_image is None else out


def openpose_call(sd, images, short_side=184, bicubic_impl='numpy'):
    resized, scale = facade.pose_resize(images, short_side)
    x = torch.from_numpy(np.transpose(resized, (0, 3, 1, 2)).astype(np.float32) / 255.0 - 0.5)
    pafs, hms = nets.openpose_forward(sd, x)
    return openpose_post.postprocess(pafs.numpy(), hms.numpy(), scale, bicubic_impl)


# ---- facades ---------------------------------------------------------------------
def _is_single(images):
    return not isinstance(images, (list, tuple)) and images.ndim == 3


def detection(sd, images, short_side=416):
    expanded = _is_single(images)
    if expanded:
        images = np.expand_dims(images, 0)
    images, scales = facade.det_resize_in(images, short_side)
    images, mp = facade.merge_in(images)
    out = retinaface_call(sd, images)
    out = facade.det_merge_out(out, mp)
    out = facade.det_resize_out(out, scales)
    return out[0] if expanded else out


def recognition(sd, images, faces_per_image=None):
    expanded = False
    if _is_single(images):
        expanded = True
        images = [images]
        faces_per_image = [[faces_per_image]] if isinstance(faces_per_image, dict) else [faces_per_image]
    if faces_per_image is not None and len(faces_per_image) != len(images):
        raise ValueError('`images` and `faces_per_image` must be of the same size')
    out = arcface_call(sd, images, faces_per_image)
    # reference quirk (face/recognition/__init__.py:85): the dict test is dead code
    return out[0] if expanded else out


def estimation(sd, images, short_side=184, bicubic_impl='numpy'):
    expanded = _is_single(images)
    if expanded:
        images = np.expand_dims(images, 0)
    images, mp = facade.merge_in(images)
    out = openpose_call(sd, images, short_side, bicubic_impl)
    out = facade.pose_merge_out(out, mp)
    return out[0] if expanded else out

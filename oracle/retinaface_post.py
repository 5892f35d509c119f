"""ORACLE (test infrastructure only): numpy float32 restatement of RetinaFace
post-processing.  Never imported by the product path (`terran_amd/`).

Follows:
  anchor_reference / anchors_plane : terran/face/detection/retinaface/anchors.py:54-134, 7-51
  decode_bboxes / decode_landmarks : terran/face/detection/retinaface/wrapper.py:25-61, 64-89
  select (threshold, sort, NMS)    : terran/face/detection/retinaface/wrapper.py:207-236
  nms                              : torchvision.ops.nms (setup.py:22 `torchvision`, version
                                     unpinned, NOT vendored in /root/reference) - restated from its
                                     documented semantics: greedy over descending scores,
                                     IoU = inter/(area_a+area_b-inter) (no +1), suppress iff IoU > thr,
                                     kept indices returned in descending-score order.  PARITY UNPINNED
                                     for this one function (no reference test or golden vector exists).
Tie rule (reference leaves it unspecified): equal scores
keep ascending anchor order (stable sort).
"""
import math

import numpy as np

F32 = np.float32

# stride -> (base_size, scales); ratio is 1 everywhere (wrapper.py:101-117)
ANCHOR_SETTINGS = {32: (16, (32, 16)), 16: (16, (8, 4)), 8: (16, (2, 1))}
STRIDES = (32, 16, 8)


def anchor_reference(stride):
    """(A,4) float32 reference boxes centred on the 16x16 base box."""
    base, scales = ANCHOR_SETTINGS[stride]
    ctr = 0.5 * (base - 1)                       # base box [0,0,15,15], ratio 1 keeps w=h=16
    out = []
    for s in scales:
        side = base * s
        half = 0.5 * (side - 1)
        out.append([ctr - half, ctr - half, ctr + half, ctr + half])
    return np.asarray(out, dtype=F32)


def anchors_plane(stride, feat_h, feat_w):
    """(feat_h*feat_w*A, 4) float32, order (y, x, anchor)."""
    ref = anchor_reference(stride)
    sy = (np.arange(feat_h, dtype=F32) * F32(stride))[:, None, None]
    sx = (np.arange(feat_w, dtype=F32) * F32(stride))[None, :, None]
    plane = np.empty((feat_h, feat_w, ref.shape[0], 4), F32)
    plane[..., 0] = ref[None, None, :, 0] + sx
    plane[..., 1] = ref[None, None, :, 1] + sy
    plane[..., 2] = ref[None, None, :, 2] + sx
    plane[..., 3] = ref[None, None, :, 3] + sy
    return plane.reshape(-1, 4)


def _anchor_wh_ctr(anchors):
    w = anchors[:, 2] - anchors[:, 0] + F32(1.0)
    h = anchors[:, 3] - anchors[:, 1] + F32(1.0)
    cx = anchors[:, 0] + F32(0.5) * (w - F32(1.0))
    cy = anchors[:, 1] + F32(0.5) * (h - F32(1.0))
    return w, h, cx, cy


def decode_bboxes(anchors, deltas):
    """anchors (A,4), deltas (N,A,4) -> (N,A,4); every op rounded to float32."""
    w, h, cx, cy = _anchor_wh_ctr(anchors)
    d = deltas.astype(F32)
    pcx = d[..., 0] * w + cx
    pcy = d[..., 1] * h + cy
    pw = np.exp(d[..., 2]) * w
    ph = np.exp(d[..., 3]) * h
    out = np.empty_like(d)
    out[..., 0] = pcx - F32(0.5) * (pw - F32(1.0))
    out[..., 1] = pcy - F32(0.5) * (ph - F32(1.0))
    out[..., 2] = pcx + F32(0.5) * (pw - F32(1.0))
    out[..., 3] = pcy + F32(0.5) * (ph - F32(1.0))
    return out


def decode_landmarks(anchors, deltas):
    """anchors (A,4), deltas (N,A,5,2) -> (N,A,5,2)."""
    w, h, cx, cy = _anchor_wh_ctr(anchors)
    d = deltas.astype(F32)
    out = np.empty_like(d)
    out[..., 0] = d[..., 0] * w[None, :, None] + cx[None, :, None]
    out[..., 1] = d[..., 1] * h[None, :, None] + cy[None, :, None]
    return out


def decode_outputs(outputs, H, W):
    """Nine head tensors (numpy, NCHW, reference order s32,s16,s8 x cls/bbox/lmk)
    -> (scores (N,T), boxes (N,T,4), landmarks (N,T,5,2)); T concatenates strides
    32 -> 16 -> 8, within a stride (y, x, anchor).  wrapper.py:153-202."""
    A = 2
    sc_l, bb_l, lm_l = [], [], []
    for i, s in enumerate(STRIDES):
        fh, fw = math.ceil(H / s), math.ceil(W / s)
        anchors = anchors_plane(s, fh, fw)
        cls, bbox, lmk = (np.asarray(outputs[3 * i + k], dtype=F32) for k in range(3))
        N = cls.shape[0]
        scores = cls[:, A:].transpose(0, 2, 3, 1).reshape(N, -1)
        bd = bbox.transpose(0, 2, 3, 1).reshape(N, -1, 4)
        ld = lmk.transpose(0, 2, 3, 1).reshape(N, -1, 5, 2)
        sc_l.append(scores)
        bb_l.append(decode_bboxes(anchors, bd))
        lm_l.append(decode_landmarks(anchors, ld))
    return (np.concatenate(sc_l, 1), np.concatenate(bb_l, 1), np.concatenate(lm_l, 1))


def nms(boxes, thr):
    """Greedy NMS over boxes ALREADY in descending-score order.  Returns kept
    indices (ascending = descending score).  All arithmetic float32."""
    boxes = np.asarray(boxes, dtype=F32)
    n = boxes.shape[0]
    thr = F32(thr)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 == n:
            break
        xx1 = np.maximum(x1[i], x1[i + 1:])
        yy1 = np.maximum(y1[i], y1[i + 1:])
        xx2 = np.minimum(x2[i], x2[i + 1:])
        yy2 = np.minimum(y2[i], y2[i + 1:])
        w = np.maximum(F32(0), xx2 - xx1)
        h = np.maximum(F32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= ovr > thr
    return np.asarray(keep, dtype=np.int64)


def select(scores, boxes, landmarks, threshold=0.5, nms_threshold=0.4):
    """Per-image selection.  scores (T,), boxes (T,4), landmarks (T,5,2) ->
    (kept anchor indices (K,) int64 in descending-score order, list of dicts)."""
    scores = np.asarray(scores, dtype=F32)
    cand = np.nonzero(scores >= F32(threshold))[0]
    if cand.size == 0:
        return cand.astype(np.int64), []
    order = np.argsort(-scores[cand], kind='stable')
    cand = cand[order]
    keep = nms(boxes[cand], nms_threshold)
    idx = cand[keep]
    objs = [{'bbox': boxes[i].copy(), 'landmarks': landmarks[i].copy(), 'score': scores[i]}
            for i in idx]
    return idx.astype(np.int64), objs


def postprocess(outputs, H, W, threshold=0.5, nms_threshold=0.4):
    """Full wrapper tail: nine head tensors -> list[N] of list[dict]."""
    scores, boxes, lmks = decode_outputs(outputs, H, W)
    return [select(scores[n], boxes[n], lmks[n], threshold, nms_threshold)[1]
            for n in range(scores.shape[0])]

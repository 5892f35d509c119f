"""TEST INFRASTRUCTURE: how far was the ORACLE from deciding otherwise?

north_star asks for discrete results (detection indices / counts, keypoint assignments) bit-exact with the reference's fp32
path.  Two correct float32 evaluations of one network differ in the last bits (summation order), so a decision whose deciding
quantity sits within rounding distance of its threshold is not determined by "fp32" at all -- and everything else IS, and must
come out exactly.  This module computes, for every decision the device and the oracle disagree on, the MARGIN the oracle had at
that decision site, and sorts the disagreement into

    above   the oracle's margin exceeds MARGIN_TOL (1e-4; float32 evaluation noise of these networks' maps is ~1e-6 .. 1e-5):
            fp32 itself determines the answer, the device got it wrong.  The suites assert ZERO of these, in every parity mode.
    sub     the deciding quantity lies within MARGIN_TOL of its threshold (or the disagreement follows from one that does):
            a near-tie either float32 evaluation may break either way.  Counted and reported, bounded at the exact-f32 mode's rate.

Decision sites and their margins (all quantities are the oracle's own float32 values):
  pose (openpose/wrapper.py:241-247, 325-330, 335-359, 470-478)
    peak (part, y, x)        min(h - 0.1, h - max(4 neighbours))                                   [>= both: signed, >= 0 is a peak]
    candidate (limb, a, b)   min(9th largest of the ten samples - 0.05, regularised score - 0)     [count > 8 and reg > 0]
    greedy choice            |reg - reg'| between accepted candidates that compete for an endpoint (the ONE `seen` set is shared by
                             source and destination indices, Appendix C #15, so "compete" = share any index value)
    human filter             total / count - 0.4 (count is an integer)
  detector (retinaface/wrapper.py:212-228)
    threshold                score - 0.5
    suppression              |IoU(i, j) - 0.4| for the pairs that decide whether i survives, and |score_i - score_j| where the
                             order of the two decides who suppresses whom
A disagreement that is the CONSEQUENCE of another one (a connection one of whose endpoints is a flipped peak, a human built
from a flipped connection, a box suppressed by a box that itself flipped) takes the class of its cause.
"""
import numpy as np

from oracle import openpose_post as opp
from oracle import retinaface_post as rfp

MARGIN_TOL = 1e-4
F32 = np.float32


# ---- pose ------------------------------------------------------------------------------------------------------------
class PoseFrame:
    """The oracle's upsampled maps of ONE frame -> margins of its pose decisions."""

    def __init__(self, hm_up, paf_up):
        self.hm, self.paf = np.asarray(hm_up, F32), np.asarray(paf_up, F32)
        self.peaks = opp.find_peaks(self.hm)
        self.index = [{(int(y), int(x)): k for k, (y, x) in enumerate(locs)} for locs, _ in self.peaks]
        self._limb = {}

    def peak_margin(self, part, y, x):
        """Signed: >= 0 -> the oracle has a peak here by that much; < 0 -> it does not, by that much."""
        m = self.hm[part]
        H, W = m.shape
        if not (1 <= y < H - 1 and 1 <= x < W - 1):
            return -np.inf                                           # border pixels are never peaks: fully determined
        c = m[y, x]
        return float(min(c - F32(opp.KEYPOINT_THRESHOLD), c - max(m[y - 1, x], m[y + 1, x], m[y, x - 1], m[y, x + 1])))

    def limb(self, limb_id):
        """(reg (Ns,Nd), accept margin (Ns,Nd) signed, accept (Ns,Nd) bool) over the oracle's peaks; None if the limb is skipped."""
        if limb_id not in self._limb:
            ks, kd = opp.LIMBSEQ[limb_id][0] - 1, opp.LIMBSEQ[limb_id][1] - 1
            ls, ld = self.peaks[ks][0], self.peaks[kd][0]
            if ls.shape[0] == 0 or ld.shape[0] == 0:
                self._limb[limb_id] = None
            else:
                dbg = {}
                reg, acc = opp.score_limb(self.paf, limb_id, ls, ld, dbg)
                mid = np.where(np.isnan(dbg['mid']), -np.inf, dbg['mid'])
                s9 = np.sort(mid, axis=0)[1]                         # 9th largest of ten = 2nd smallest: count > 8  <=>  s9 > 0.05
                regc = np.where(np.isnan(reg), -np.inf, reg)         # coincident endpoints: NaN -> rejected, whatever the rounding
                m_acc = np.minimum(s9 - F32(opp.MIDPOINT_THRESHOLD), regc).astype(np.float64)
                self._limb[limb_id] = (reg, m_acc, acc)
        return self._limb[limb_id]

    def classify(self, dev_peaks, dev_conns, dev_humans, ref_peaks, ref_conns, ref_humans, tol=MARGIN_TOL):
        """Sets as tests/test_gpu_decisions_vs_oracle.py builds them: peaks {(part, y, x)}, conns {(limb, sy, sx, dy, dx)},
        humans [keypoint bytes].  -> {'peaks': (above, sub), 'conns': (above, sub), 'humans': (above, sub), 'worst': [...]}"""
        worst = []
        pk_class = {}
        for p in dev_peaks ^ ref_peaks:
            m = abs(self.peak_margin(*p))
            pk_class[p] = 'sub' if m <= tol else 'above'
            worst.append(('peak', p, m))
        above_parts = {p[0] for p, c in pk_class.items() if c == 'above'}
        sub_parts = {p[0] for p, c in pk_class.items() if c == 'sub'}
        cn_class = {}
        diff_conns = dev_conns ^ ref_conns
        by_limb = {}
        for c in diff_conns:
            by_limb.setdefault(c[0], []).append(c)
        for limb_id, cs in by_limb.items():
            ks, kd = opp.LIMBSEQ[limb_id][0] - 1, opp.LIMBSEQ[limb_id][1] - 1
            # a flipped peak of either endpoint part changes the candidate lists (and their indices) of this limb
            if ks in above_parts or kd in above_parts:
                cls, m = 'above', np.inf
            elif ks in sub_parts or kd in sub_parts:
                cls, m = 'sub', 0.0
            else:
                m = self._limb_margin(limb_id, cs)
                cls = 'sub' if m <= tol else 'above'
            for c in cs:
                cn_class[c] = cls
                worst.append(('conn', c, m))
        hd = set(dev_humans) ^ set(ref_humans)
        hm_class = {}
        if hd:
            causes = list(pk_class.values()) + list(cn_class.values())
            if causes:
                cls, m = ('above' if 'above' in causes else 'sub'), None
            else:                                                     # same peaks, same connections: only the final filter is left
                m = self._human_filter_margin()
                cls = 'sub' if m <= tol else 'above'
            for h in hd:
                hm_class[h] = cls
            worst.append(('humans', len(hd), m))

        def count(d):
            return (sum(1 for v in d.values() if v == 'above'), sum(1 for v in d.values() if v == 'sub'))
        return {'peaks': count(pk_class), 'conns': count(cn_class), 'humans': count(hm_class), 'worst': worst}

    def _limb_margin(self, limb_id, conns):
        """Smallest oracle margin among the decisions that can explain differing connections of one limb whose endpoint peak
        lists are the same on both sides: the acceptance of every candidate that touches an involved peak, and the order of
        every two accepted (or nearly accepted) candidates that compete for one."""
        t = self.limb(limb_id)
        if t is None:
            return np.inf
        reg, m_acc, _ = t
        ks, kd = opp.LIMBSEQ[limb_id][0] - 1, opp.LIMBSEQ[limb_id][1] - 1
        involved = set()
        for (_, sy, sx, dy, dx) in conns:
            a, b = self.index[ks].get((sy, sx)), self.index[kd].get((dy, dx))
            if a is None or b is None:
                return np.inf                                        # an endpoint the oracle does not have, yet no peak flip: not a near-tie
            involved.update((a, b))                                  # index VALUES: the `seen` set mixes source and destination indices
        ns, nd = reg.shape
        ii, jj = np.meshgrid(np.arange(ns), np.arange(nd), indexing='ij')
        touch = np.isin(ii, list(involved)) | np.isin(jj, list(involved))
        m = float(np.abs(m_acc[touch]).min()) if touch.any() else np.inf
        live = touch & (m_acc > -MARGIN_TOL)                         # accepted, or within the tolerance of being accepted
        r = np.sort(reg[live].astype(np.float64))
        if r.size > 1:
            m = min(m, float(np.diff(r).min()))
        return m

    def _human_filter_margin(self):
        conns = []
        for limb_id in range(19):
            t = self.limb(limb_id)
            conns.append(None if t is None else opp.greedy_match(t[0], t[2]))
        dbg = {}
        opp.assemble_humans(self.peaks, conns, dbg)
        # count < 4 is an integer test (fully determined); the average is compared with 0.4 only for count >= 4
        m = [abs(h[18] / h[19] - opp.HUMAN_THRESHOLD) for h in dbg['unfiltered'] if h[19] >= 4]
        return float(min(m)) if m else np.inf


# ---- detector --------------------------------------------------------------------------------------------------------
def _iou(b, others):
    """torchvision-convention IoU (no +1) of one box against (n,4) boxes, float32 like oracle.retinaface_post.nms."""
    b, o = np.asarray(b, F32), np.asarray(others, F32)
    xx1, yy1 = np.maximum(b[0], o[:, 0]), np.maximum(b[1], o[:, 1])
    xx2, yy2 = np.minimum(b[2], o[:, 2]), np.minimum(b[3], o[:, 3])
    inter = np.maximum(F32(0), xx2 - xx1) * np.maximum(F32(0), yy2 - yy1)
    a = (b[2] - b[0]) * (b[3] - b[1])
    ao = (o[:, 2] - o[:, 0]) * (o[:, 3] - o[:, 1])
    with np.errstate(divide='ignore', invalid='ignore'):
        return inter / (a + ao - inter)


class DetectorFrame:
    """The oracle's decoded anchors of ONE image (scores (T,), boxes (T,4)) -> margins of its selection decisions."""

    def __init__(self, scores, boxes, threshold=0.5, nms_threshold=0.4):
        self.s, self.b = np.asarray(scores, F32), np.asarray(boxes, F32)
        self.thr, self.nms_thr = F32(threshold), F32(nms_threshold)
        self.near = np.nonzero(self.s >= self.thr - F32(MARGIN_TOL))[0]      # every anchor that passes or nearly passes the threshold

    def find(self, bbox, coord_tol=2e-3):
        """Index of the oracle anchor whose decoded box is the device's `bbox` (to `coord_tol` per coordinate), or None."""
        d = np.abs(self.b[self.near] - np.asarray(bbox, F32)[None]).max(1)
        k = int(np.argmin(d)) if d.size else -1
        if k >= 0 and d[k] <= coord_tol:
            return int(self.near[k])
        d = np.abs(self.b - np.asarray(bbox, F32)[None]).max(1)             # an anchor the oracle scores clearly below the threshold
        k = int(np.argmin(d))
        return k if d[k] <= coord_tol else None

    def margin(self, i):
        """Smallest margin among the decisions that decide whether anchor i is in the output: its own threshold test, the IoU
        tests against every other (nearly) passing anchor, and the score order against every anchor that overlaps it enough to
        suppress / be suppressed."""
        m = abs(float(self.s[i] - self.thr))
        others = self.near[self.near != i]
        if others.size:
            iou = _iou(self.b[i], self.b[others])
            iou = np.where(np.isnan(iou), 0.0, iou)
            m = min(m, float(np.abs(iou - self.nms_thr).min()))
            ov = iou > self.nms_thr - F32(MARGIN_TOL)
            if ov.any():
                m = min(m, float(np.abs(self.s[others[ov]] - self.s[i]).min()))
        return m

    def classify(self, dev_dets, ref_keys, key, tol=MARGIN_TOL):
        """dev_dets: the device's dicts for this image; ref_keys: the oracle's keys in output order; key(d) -> hashable.
        -> {'dets': (above, sub), 'worst': [...]}.  A disagreement is `sub` when its own margin is, or when it overlaps (IoU
        above the suppression threshold) another disagreement that is: suppression chains take the class of their first link."""
        dev_keys = [key(d) for d in dev_dets]
        diff = set(dev_keys) ^ set(ref_keys)
        if not diff:
            return {'dets': (0, 0), 'rekeyed': 0, 'worst': []}
        # oracle anchors behind the differing keys
        idx = {}
        ref_sel, _ = rfp.select(self.s, self.b, np.zeros((len(self.s), 5, 2), F32), float(self.thr), float(self.nms_thr))
        ref_key_of = {int(i): key({'bbox': self.b[i]}) for i in ref_sel}
        for i, k in ref_key_of.items():
            if k in diff:
                idx[k] = i
        rekeyed = 0
        for d in dev_dets:
            k = key(d)
            if k in diff and k not in idx:
                i = self.find(d['bbox'])
                if i is not None and ref_key_of.get(i) in diff and ref_key_of[i] in idx:
                    # the SAME anchor was selected on both sides, its box agrees to 2e-3 per coordinate (north_star's 1e-3 bar is on
                    # the coordinates): only the test's own rounding of a coordinate near .5 differs -- an identity-key artefact
                    del idx[ref_key_of[i]]
                    rekeyed += 1
                    continue
                idx[k] = i
        margins = {k: (self.margin(i) if i is not None else np.inf) for k, i in idx.items()}
        cls = {k: ('sub' if m <= tol else 'above') for k, m in margins.items()}
        # chains: a disagreement that overlaps a `sub` disagreement inherits it
        changed = True
        while changed:
            changed = False
            for k, i in idx.items():
                if cls[k] == 'above' and i is not None:
                    for k2, i2 in idx.items():
                        if k2 != k and cls[k2] == 'sub' and i2 is not None:
                            v = _iou(self.b[i], self.b[[i2]])[0]
                            if v > self.nms_thr - F32(MARGIN_TOL):
                                cls[k] = 'sub'
                                changed = True
                                break
        worst = [('det', k, margins[k]) for k in idx]
        above = sum(1 for v in cls.values() if v == 'above')
        return {'dets': (above, len(cls) - above), 'rekeyed': rekeyed, 'worst': worst}

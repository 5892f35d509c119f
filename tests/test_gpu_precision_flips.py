"""-m gpu: how often do the split-operand arithmetic modes -- the default `f16x3` (split-half operands, 22 significant
bits) and `bf16x3` (split-bf16, 16 bits; 3 MFMAs per product each) -- take a DIFFERENT discrete decision than the exact-f32
mode ON THE DEVICE, over a few hundred frames per task?  (Each mode against the ORACLE: test_gpu_decisions_vs_oracle.py.)

Decisions counted: RetinaFace score threshold (>= 0.5) + greedy NMS (the kept anchors per image); OpenPose peaks
(>= neighbours, >= 0.1) and accepted limb connections on the random-weight net (hundreds of near-tie decisions per frame)
and assembled humans on frames that carry pose maps.  Both modes pass the same parity suite against the oracle; this
test bounds how far the two can drift apart on inputs nobody tuned, and prints the measured rates.
"""
import numpy as np
import pytest

from terran_amd import synth

pytestmark = pytest.mark.gpu
N_FRAMES = 208


def _det_keys(dets):
    """Identity of a detection independent of float noise: its box rounded to the nearest pixel."""
    return [tuple(np.rint(d['bbox']).astype(int).tolist()) for d in dets]


@pytest.mark.parametrize('mode', ['f16x3', 'bf16x3'])
def test_retinaface_decisions_f32_vs_split_modes(states, mode):
    """bf16x3: the detector runs on the exact-f32 MFMA (with bf16x3 convs 6 of 208 images came back with near-tied scores
    in a different order): same detections, same order, same bits as `f32`.  f16x3: its refiner and the deep half of its
    base run on the split-half MFMA (22-bit operands): the same detections; a near-tie may swap."""
    from terran_amd import RetinaFace
    a = RetinaFace(device=0, state=states('retinaface'), precision='f32')
    b = RetinaFace(device=0, state=states('retinaface'), precision=mode)
    n_img = n_det = diff_img = diff_det = 0
    worst = 0.0
    for k in range(0, N_FRAMES, 16):
        frames = synth.frames(1000 + k, 16, 208, 277)           # C1-sized network input: ~300 detections per image
        da, db = a.call(frames), b.call(frames)
        for x, y in zip(da, db):
            n_img += 1
            n_det += len(x)
            kx, ky = _det_keys(x), _det_keys(y)
            if kx != ky:
                diff_img += 1
                diff_det += len(set(kx) ^ set(ky))
            else:
                for p, q in zip(x, y):
                    worst = max(worst, float(np.abs(p['bbox'] - q['bbox']).max()), float(abs(p['score'] - q['score'])))
    print('retinaface f32 vs split modes: %d images, %d detections; %d images differ (%d detections); max |bbox/score| diff '
          'on identical sets %.2e' % (n_img, n_det, diff_img, diff_det, worst))
    assert n_det > 5000
    if mode == 'bf16x3':
        assert diff_img == 0 and diff_det == 0 and worst == 0.0
    else:                                              # measured: 0 images, 0 detections, 1.2e-4 px / 2e-7 score
        assert diff_img <= 2 and diff_det <= 2 and worst < 2e-3


# measured (MI355X, 208 frames): f16x3 differs from f32 in <= 2 of 16 468 peaks, 0 connections, 0 humans;
# bf16x3 in 6 peaks, 0 connections, 2 humans (a symmetric difference: one person)
BOUNDS = {'f16x3': lambda t: (4, 2, 0), 'bf16x3': lambda t: (max(4, t['peaks'] // 500), max(4, t['conns'] // 200), max(2, t['humans'] // 100))}


@pytest.mark.parametrize('mode', ['f16x3', 'bf16x3'])
def test_openpose_decisions_f32_vs_split_modes(states, mode):
    from terran_amd import OpenPose
    tot = dict(peaks=0, conns=0, dpeaks=0, dconns=0, humans=0, dhumans=0)
    for sd_name, frames_fn, short in (
            ('openpose', lambda k: synth.frames(2000 + k, 16, 96, 128), 96),
            ('openpose_decoder', lambda k: synth.pose_code_frames(3000 + k, 16, 96, 128, 3), 96)):
        a = OpenPose(device=0, short_side=short, state=states(sd_name), precision='f32')
        b = OpenPose(device=0, short_side=short, state=states(sd_name), precision=mode)
        for k in range(0, N_FRAMES // 2, 16):
            frames = frames_fn(k)
            ha = a.call(frames)
            pa, ca = a.ctx.pose_debug(16)
            hb = b.call(frames)
            pb, cb = b.ctx.pose_debug(16)
            for i in range(16):
                for part in range(18):
                    sa = set(map(tuple, pa[i][part][0].tolist()))
                    sb = set(map(tuple, pb[i][part][0].tolist()))
                    tot['peaks'] += len(sa)
                    tot['dpeaks'] += len(sa ^ sb)
                for limb in range(19):
                    ea = set() if ca[i][limb] is None else set(map(tuple, ca[i][limb][0].tolist()))
                    eb = set() if cb[i][limb] is None else set(map(tuple, cb[i][limb][0].tolist()))
                    tot['conns'] += len(ea)
                    tot['dconns'] += len(ea ^ eb)
                ka = [h['keypoints'].tobytes() for h in ha[i]]
                kb = [h['keypoints'].tobytes() for h in hb[i]]
                tot['humans'] += len(ka)
                tot['dhumans'] += len(set(ka) ^ set(kb))
    print('openpose f32 vs %s over %d frames: %s' % (mode, N_FRAMES // 16 * 16, tot))
    assert tot['peaks'] > 5000 and tot['conns'] > 1000 and tot['humans'] > 200
    bp, bc, bh = BOUNDS[mode](tot)
    assert tot['dpeaks'] <= bp and tot['dconns'] <= bc and tot['dhumans'] <= bh, tot


@pytest.mark.parametrize('mode', ['f16x3', 'bf16x3'])
def test_arcface_embeddings_f32_vs_split_modes(states, mode):
    from terran_amd import ArcFace
    a = ArcFace(device=0, state=states('arcface'), precision='f32')
    b = ArcFace(device=0, state=states('arcface'), precision=mode)
    crops = np.random.default_rng(5).integers(0, 256, (N_FRAMES, 3, 112, 112), dtype=np.uint8)
    ea, eb = a.embed_crops(crops), b.embed_crops(crops)
    d = float(np.abs(ea - eb).max())
    cos = 1.0 - float((ea * eb).sum(1).min())
    print('arcface f32 vs %s over %d crops: max |diff| of unit embeddings %.2e, max cosine distance %.2e' % (mode, N_FRAMES, d, cos))
    assert d < (5e-6 if mode == 'f16x3' else 2e-4) and cos < 1e-6

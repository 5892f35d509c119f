"""-m gpu: discrete decisions of every device arithmetic mode counted against the ORACLE (the CPU restatement of the
reference, torch-CPU float32 convolutions), over >= 200 frames per task at 96 x 128 / 208 x 277 and >= 16 frames at the
working sizes of BASELINE configs[4] (pose 184 x 327, detector 416 x 739).

north_star's bar is "detection indices / counts and keypoint assignments bit-exact"; two float32 implementations of
one network agree to ~1e-6, so on inputs nobody tuned a few of the thousands of near-tie decisions per batch
(peak >= neighbour, score >= threshold, argsort of near-equal scores) land on the other side.  This test counts them
per mode -- decisions are identified by COORDINATES (a peak by (part, y, x), a connection by its two peaks, a detection
by its rounded box), so one flipped peak does not shift every later index -- prints the table DESIGN.md section 4
carries, writes it to gpurun_out/decisions_vs_oracle.json, and asserts that the headline mode decides no worse than the
exact-f32 MFMA mode does.
"""
import json
import os

import numpy as np
import pytest
import torch

from terran_amd import pack, synth
from tests.util import REPO

pytestmark = pytest.mark.gpu
MODES = [m for m in ('f32', 'f16x3', 'bf16x3', 'f16') if m in pack.PRECISIONS]      # 'f16': the opt-in mode; its detector / pose programs are f16x3's
HEADLINE = 'f16x3' if 'f16x3' in pack.PRECISIONS else 'bf16x3'
N_SMALL, N_WORK, BATCH = 208, 16, 16
# a longer count for profiles/ (the suite keeps the sizes above): TA_DECISIONS_SCALE=5 -> 1040 + 80 frames per task
_SCALE = max(1, int(os.environ.get('TA_DECISIONS_SCALE', '1')))
N_SMALL, N_WORK = N_SMALL * _SCALE, N_WORK * _SCALE
_results = {}


def rare_bound(n):
    """Upper bound for a count of rare events that is claimed to occur at the rate of a reference count `n`: n + n / 2 plus
    two standard deviations of a Poisson count of that size (never less than 3).  For the ill-conditioned wild weights and
    for the long counts (TA_DECISIONS_SCALE > 1) only: the benign cases at the suite's own size are held to `tight_bound`."""
    return n + max(3, n // 2 + 2 * int(np.sqrt(n)))


def tight_bound(n):
    """Benign weights at the suite's default size: the default mode may flip ONE near-tie more than the exact-f32 MFMA mode
    does (measured: 0 where f32 has 0 - 2).  A doubling of the default mode's flip rate fails here."""
    return n + 1


def benign_bound(n):
    return tight_bound(n) if _SCALE == 1 else rare_bound(n)


def _record(task, mode, tot):
    _results.setdefault(task, {})[mode] = tot
    out = os.path.join(REPO, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'decisions_vs_oracle.json'), 'w') as fh:
            json.dump(_results, fh, indent=1, sort_keys=True)
    except OSError:
        pass


# ---- margins: which disagreements does fp32 itself determine? (tests/margins.py) -----------------------------------------
# Every disagreement between a device mode and the oracle is looked up in the ORACLE's own maps: `above` = the oracle decided
# with more than 1e-4 to spare (the device is wrong: asserted ZERO in every parity mode), `sub` = a near-tie within 1e-4 of its
# threshold, or the consequence of one (counted, reported, held to the exact-f32 mode's rate by the bounds below).
from tests import margins      # noqa: E402
ZERO_ABOVE_MODES = ('f32', 'f16x3', 'f16', 'f16x2')      # bf16x3 (16-bit operands) is reported beside them
_margin_cache = {}


def _cached(key, make):
    if key not in _margin_cache:
        while len(_margin_cache) >= 3:
            _margin_cache.pop(next(iter(_margin_cache)))
        _margin_cache[key] = make()
    return _margin_cache[key]


def _pose_margin_frames(sd, frames, short):
    """The oracle's upsampled maps of one batch -> [margins.PoseFrame]."""
    from oracle import facade, nets, openpose_post
    resized, _ = facade.pose_resize(frames, short)
    x = torch.from_numpy(np.transpose(resized, (0, 3, 1, 2)).astype(np.float32) / 255.0 - 0.5)
    pafs, hms = nets.openpose_forward(sd, x)
    paf_up = openpose_post.bicubic_x8(pafs.numpy(), 'torch')
    hm_up = openpose_post.bicubic_x8(hms.numpy(), 'torch')
    return [margins.PoseFrame(hm_up[i], paf_up[i]) for i in range(len(frames))]


def _det_margin_frames(sd, images):
    """The oracle's decoded anchors of one batch -> [margins.DetectorFrame]."""
    from oracle import nets, retinaface_post
    H, W = images.shape[1:3]
    x = torch.from_numpy(np.ascontiguousarray(images)).to(torch.float32).permute(0, 3, 1, 2).flip(1).contiguous()
    outs = [o.numpy() for o in nets.retinaface_forward(sd, x)]
    scores, boxes, _ = retinaface_post.decode_outputs(outs, H, W)
    return [margins.DetectorFrame(scores[i], boxes[i]) for i in range(len(images))]


# ---- pose ------------------------------------------------------------------------------------------------------------
def _pose_sets_oracle(sd, frames, short):
    """Per frame: (set of (part, y, x), set of (limb, sy, sx, dy, dx), list of keypoint bytes)."""
    from oracle import facade, nets, openpose_post
    resized, scale = facade.pose_resize(frames, short)
    x = torch.from_numpy(np.transpose(resized, (0, 3, 1, 2)).astype(np.float32) / 255.0 - 0.5)
    pafs, hms = nets.openpose_forward(sd, x)
    paf_up = openpose_post.bicubic_x8(pafs.numpy(), 'torch')
    hm_up = openpose_post.bicubic_x8(hms.numpy(), 'torch')
    out = []
    for i in range(len(frames)):
        dbg = {}
        humans = openpose_post.group_image(hm_up[i], paf_up[i], scale, dbg)
        peaks = {(p, int(y), int(x_)) for p in range(18) for y, x_ in dbg['peaks'][p][0]}
        conns = set()
        for limb, cl in enumerate(dbg['connections']):
            if cl is None:
                continue
            ks, kd = openpose_post.LIMBSEQ[limb][0] - 1, openpose_post.LIMBSEQ[limb][1] - 1
            ls, ld = dbg['peaks'][ks][0], dbg['peaks'][kd][0]
            for (a, b, _) in cl:
                conns.add((limb,) + tuple(int(v) for v in ls[a]) + tuple(int(v) for v in ld[b]))
        out.append((peaks, conns, [h['keypoints'].tobytes() for h in humans]))
    return out


def _pose_sets_device(model, frames):
    from oracle import openpose_post
    humans = model.call(frames)
    n = len(frames)
    pk, cn = model.ctx.pose_debug(n, cap_peaks=4096, cap_conn=1024)
    out = []
    for i in range(n):
        peaks = {(p, int(y), int(x_)) for p in range(18) for y, x_ in pk[i][p][0]}
        conns = set()
        for limb in range(19):
            if cn[i][limb] is None:
                continue
            ks, kd = openpose_post.LIMBSEQ[limb][0] - 1, openpose_post.LIMBSEQ[limb][1] - 1
            ls, ld = pk[i][ks][0], pk[i][kd][0]
            for a, b in cn[i][limb][0]:
                conns.add((limb,) + tuple(int(v) for v in ls[a]) + tuple(int(v) for v in ld[b]))
        out.append((peaks, conns, [h['keypoints'].tobytes() for h in humans[i]]))
    return out


N_WILD = 32
_POSE_CASES = {
    # name: (weights, frame generator(k) -> BATCH frames, short side, number of frames)
    'small_random': ('openpose', lambda k: synth.frames(2000 + k, BATCH, 96, 128), 96, N_SMALL // 2),
    'small_people': ('openpose_decoder', lambda k: synth.pose_code_frames(3000 + k, BATCH, 96, 128, 3), 96, N_SMALL // 2),
    'work_random': ('openpose', lambda k: synth.frames(4000 + k, BATCH, 184, 327), 184, N_WORK),
    'work_people': ('openpose_decoder', lambda k: synth.pose_code_frames(5000 + k, BATCH, 184, 327, 4), 184, N_WORK),
}
_pose_oracle_cache = {}


def _pose_oracle(states, case):
    if case not in _pose_oracle_cache:
        sd_name, gen, short, n = _POSE_CASES[case]
        res = []
        for k in range(0, n, BATCH):
            res += _pose_sets_oracle(states(sd_name), gen(k), short)
        _pose_oracle_cache[case] = res
    return _pose_oracle_cache[case]


@pytest.mark.parametrize('case', list(_POSE_CASES))
def test_openpose_decisions_vs_oracle(states, case):
    from terran_amd import OpenPose
    sd_name, gen, short, n = _POSE_CASES[case]
    ref = _pose_oracle(states, case)
    table = {}
    for mode in MODES:
        model = OpenPose(device=0, short_side=short, state=states(sd_name), precision=mode)
        tot = dict(frames=0, peaks=0, dpeaks=0, conns=0, dconns=0, humans=0, dhumans=0, range_fallbacks=0, above_margin=0, sub_margin=0)
        for k in range(0, n, BATCH):
            got = _pose_sets_device(model, gen(k))
            for j, ((gp, gc, gh), (rp, rc, rh)) in enumerate(zip(got, ref[k:k + BATCH])):
                tot['frames'] += 1
                if gp != rp or gc != rc or set(gh) != set(rh):
                    pf = _cached(('pose', case, k), lambda: _pose_margin_frames(states(sd_name), gen(k), short))[j]
                    c = pf.classify(gp, gc, gh, rp, rc, rh)
                    tot['above_margin'] += sum(c[x][0] for x in ('peaks', 'conns', 'humans'))
                    tot['sub_margin'] += sum(c[x][1] for x in ('peaks', 'conns', 'humans'))
                    if any(c[x][0] for x in ('peaks', 'conns', 'humans')):
                        print('  ABOVE-MARGIN disagreement, %s frame %d: %s' % (mode, k + j, c['worst'][:6]))
                tot['peaks'] += len(rp)
                tot['dpeaks'] += len(gp ^ rp)
                tot['conns'] += len(rc)
                tot['dconns'] += len(gc ^ rc)
                tot['humans'] += len(rh)
                tot['dhumans'] += len(set(gh) ^ set(rh))
        tot['range_fallbacks'] = model.fallbacks
        table[mode] = tot
        _record('openpose_' + case, mode, tot)
        print('openpose %s, device %s vs oracle: %s' % (case, mode, tot))
    f32, head = table['f32'], table[HEADLINE]
    # north_star's "bit-exact": wherever the oracle's fp32 decides with more than 1e-4 to spare, every parity mode decides the same
    assert all(table[m]['above_margin'] == 0 for m in ZERO_ABOVE_MODES if m in table), table
    assert all(t['above_margin'] + t['sub_margin'] == t['dpeaks'] + t['dconns'] + t['dhumans'] for t in table.values()), table
    assert f32['peaks'] > 500 and (f32['humans'] > 40 or 'random' in case)
    assert all(t['range_fallbacks'] == 0 for t in table.values()), table        # the activation scales keep every tensor inside the half-float range
    # the exact-f32 MFMA mode itself: a handful of near-ties per thousand decisions at most (per hundred connections on the
    # ill-conditioned wild weights, where two float32 evaluations of one network differ by more)
    assert f32['dpeaks'] <= max(3, f32['peaks'] // 1000) and f32['dhumans'] <= max(2, f32['humans'] // 100)
    # the default mode flips near-ties at the exact-f32 mode's RATE: at the suite's size the bound is that mode's own count
    # + 1 (`tight_bound`); the long counts (TA_DECISIONS_SCALE=5 measures 12 vs 15 and 21 vs 16 peaks of ~50 000 on the random
    # frames, 0 vs 0 on the people) are counts of rare events and get the statistical bound (`rare_bound`)
    slack = benign_bound
    assert head['dpeaks'] <= slack(f32['dpeaks']) and head['dconns'] <= slack(f32['dconns']) and head['dhumans'] <= slack(f32['dhumans']), table
    if 'f16' in table:                                   # the opt-in mode packs this network exactly as f16x3 does: same decisions
        assert {k: v for k, v in table['f16'].items()} == {k: v for k, v in head.items()}, table


# ---- detector --------------------------------------------------------------------------------------------------------
def _det_keys(dets):
    return [tuple(np.rint(d['bbox']).astype(int).tolist()) for d in dets]


_DET_CASES = {
    'small': ('retinaface', lambda k: synth.frames(1000 + k, BATCH, 208, 277), N_SMALL),
    'work': ('retinaface', lambda k: synth.frames(6000 + k, BATCH, 416, 739), N_WORK),
}


@pytest.mark.parametrize('case', list(_DET_CASES))
def test_retinaface_decisions_vs_oracle(states, case):
    from oracle import pipeline
    from terran_amd import RetinaFace
    sd_name, gen, n = _DET_CASES[case]
    sd = states(sd_name)
    wild = case.endswith('_wild')
    ref = []
    for k in range(0, n, BATCH):
        ref += [_det_keys(d) for d in pipeline.retinaface_call(sd, gen(k))]
    table = {}
    for mode in MODES:
        model = RetinaFace(device=0, state=sd, precision=mode)
        tot = dict(images=0, dets=0, ddets=0, images_reordered=0, positions_swapped=0, range_fallbacks=0, above_margin=0, sub_margin=0,
                   rekeyed=0, swaps_above_margin=0)
        for k in range(0, n, BATCH):
            for j, (gd, r) in enumerate(zip(model.call(gen(k)), ref[k:k + BATCH])):
                g = _det_keys(gd)
                tot['images'] += 1
                if set(g) != set(r):
                    df = _cached(('det', case, k), lambda: _det_margin_frames(sd, gen(k)))[j]
                    c = df.classify(gd, r, lambda d: tuple(np.rint(d['bbox']).astype(int).tolist()))
                    tot['above_margin'] += c['dets'][0]
                    tot['sub_margin'] += c['dets'][1]
                    tot['rekeyed'] += c['rekeyed']
                    if c['dets'][0]:
                        print('  ABOVE-MARGIN disagreement, %s image %d: %s' % (mode, k + j, c['worst'][:6]))
                elif g != r:                             # same set, other order: only scores within the margin may trade places
                    sc = {key: float(d['score']) for key, d in zip(g, gd)}
                    tot['swaps_above_margin'] += sum(1 for a, b in zip(g, r) if a != b and abs(sc[a] - sc[b]) > margins.MARGIN_TOL)
                tot['dets'] += len(r)
                tot['ddets'] += len(set(g) ^ set(r))
                if set(g) == set(r) and g != r:
                    tot['images_reordered'] += 1
                    tot['positions_swapped'] += sum(a != b for a, b in zip(g, r))
        tot['range_fallbacks'] = model.fallbacks
        table[mode] = tot
        _record('retinaface_' + case, mode, tot)
        print('retinaface %s, device %s vs oracle: %s' % (case, mode, tot))
    f32, head = table['f32'], table[HEADLINE]
    # north_star's "bit-exact": wherever the oracle's fp32 decides with more than 1e-4 to spare, every parity mode decides the same
    assert all(table[m]['above_margin'] == 0 and table[m]['swaps_above_margin'] == 0 for m in ZERO_ABOVE_MODES if m in table), table
    assert f32['dets'] > 1500
    assert all(t['range_fallbacks'] == 0 for t in table.values()), table
    # wild weights: the detector becomes ill-conditioned enough that the device's exact-f32 evaluation and the oracle's
    # disagree on ~0.5 % of the near-threshold anchors (tests/probe_wild_weights.py counts both against a float64 evaluation)
    assert f32['ddets'] <= (max(2, f32['dets'] // 1000) if not wild else max(4, f32['dets'] // 100))
    # f16x3: refiner + deep base on the split-half MFMA (bf16x3 keeps the whole detector exact f32): no worse than f32
    bound = rare_bound if wild else benign_bound
    assert head['ddets'] <= bound(f32['ddets']) and head['images_reordered'] <= bound(f32['images_reordered']), table
    assert table.get('bf16x3', f32) == f32
    assert table.get('f16', head) == head                # the opt-in mode's detector IS the f16x3 program


# ---- embeddings (no decisions: the distance to the oracle per mode) -------------------------------------------------
@pytest.mark.parametrize('stats', ['benign', 'wild'])
def test_arcface_embeddings_vs_oracle(states, stats):
    from oracle import arcface_pre, nets
    from terran_amd import ArcFace
    sd = states('arcface' if stats == 'benign' else 'wild_arcface')
    crops = np.random.default_rng(5).integers(0, 256, (64, 3, 112, 112), dtype=np.uint8)
    if stats == 'wild':                                  # half noise, half smooth image-like crops
        from tests import wild_weights
        crops[32:] = wild_weights._calib_frames(77, 32, 112, 112)[..., ::-1].transpose(0, 3, 1, 2)
    ref = arcface_pre.l2_normalize(nets.arcface_forward(sd, torch.from_numpy(crops.astype(np.float32))).numpy())
    table = {}
    for mode in MODES + ['f16x2']:
        a = ArcFace(device=0, state=sd, precision=mode, guard=False)      # the mode AS ASKED FOR: this test measures it (the guard has its own)
        e = a.embed_crops(crops)
        table[mode] = dict(max_abs=float(np.abs(e - ref).max()), max_cosine_distance=float(1.0 - (e * ref).sum(1).min()),
                           range_fallbacks=a.fallbacks)
        _record('arcface_' + stats, mode, table[mode])
        print('arcface 64 crops (%s weights), device %s vs oracle: %s' % (stats, mode, table[mode]))
    assert all(t['range_fallbacks'] == 0 for t in table.values()), table
    assert table['f32']['max_abs'] < (5e-6 if stats == 'benign' else 2e-5)
    assert table[HEADLINE]['max_abs'] <= max(2 * table['f32']['max_abs'], 2e-6)
    # the opt-in two-product embedder (weights and trunk at 22 bits, activations enter as their hi half): inside north_star's
    # 1e-3 on both weight sets (measured 1.8e-4 / 8.2e-4), cosine distances ~1e-6 -- with 1.2 x of headroom on the wild ones,
    # which is why the library default is f16x3 and the mode is guarded at load (test_f16x2_guard_* below)
    assert table['f16x2']['max_abs'] <= (3e-4 if stats == 'benign' else 1e-3) and table['f16x2']['max_cosine_distance'] <= 5e-6, table
    if 'f16' in table:
        # the opt-in single-half embedder: inside north_star's 1e-3 with a 3 x margin on the benign statistics; on weights with
        # trained-looking statistics 11-bit operands measure 1.8e-3 -- OUTSIDE the bar: that mode is for checkpoints it has
        # been validated on, never the default, never the bench headline
        if stats == 'benign':
            assert table['f16']['max_abs'] <= 5e-4 < 1e-3 and table['f16']['max_cosine_distance'] <= 1e-5
        else:
            assert table['f16']['max_abs'] <= 4e-3 and table['f16']['max_cosine_distance'] <= 5e-5


def test_f16x2_guard_keeps_the_mode_on_seeded_weights_and_falls_back_on_wild_ones(states, tmp_path, monkeypatch):
    """precision='f16x2' is taken on measurement, not on trust (arcface.guard_f16x2; the reference's contract is fp32,
    arcface/wrapper.py:166-176): 32 fixed calibration crops are embedded in f16x2 and in f16x3 when the model is loaded and the
    two-product program is kept only if no unit-embedding component moves by more than 5e-4.  Seeded weights stay f16x2; the
    wild-statistics weights (8e-4 against the oracle) run in f16x3 with ONE warning; a tolerance below the seeded weights' own
    figure turns them away as well; the decision travels with the repack cache of a checkpoint file."""
    import warnings
    from terran_amd import ArcFace, arcface, runtime
    ctx = runtime.get_context(0)
    sd = states('arcface')
    with warnings.catch_warnings():
        warnings.simplefilter('error')                                   # seeded weights: no warning
        a = ArcFace(device=0, state=sd, precision='f16x2')
    g = a.guard
    print('f16x2 guard, seeded weights:', g)
    assert a.precision == 'f16x2' and g['selected'] == 'f16x2' and 0 < g['max_abs_diff'] <= arcface.GUARD_TOL and g['crops'] == 32
    assert arcface.guard_f16x2(ctx, sd) is g                             # memoised per state dict
    crops = np.random.default_rng(9).integers(0, 256, (8, 3, 112, 112), dtype=np.uint8)
    e2 = a.embed_crops(crops)
    e3 = ArcFace(device=0, state=sd, precision='f16x3').embed_crops(crops)
    assert 0 < np.abs(e2 - e3).max() <= 5e-4                           # it IS the two-product program (differs from f16x3), inside the guard's bar
    # a bar below what these weights measure: the same weights are turned away
    tight = arcface.guard_f16x2(ctx, dict(sd), tol=g['max_abs_diff'] / 2)
    assert tight['selected'] == 'f16x3' and abs(tight['max_abs_diff'] - g['max_abs_diff']) < 1e-7
    # wild statistics: outside the guard's bar -> the embedder IS the f16x3 one, bit for bit, with a warning
    sw = states('wild_arcface')
    with pytest.warns(RuntimeWarning, match='f16x3'):
        w = ArcFace(device=0, state=sw, precision='f16x2')
    print('f16x2 guard, wild weights:', w.guard)
    assert w.precision == 'f16x3' and w.guard['selected'] == 'f16x3' and w.guard['max_abs_diff'] > arcface.GUARD_TOL
    assert np.array_equal(w.embed_crops(crops), ArcFace(device=0, state=sw, precision='f16x3').embed_crops(crops))
    unguarded = ArcFace(device=0, state=sw, precision='f16x2', guard=False)
    assert unguarded.precision == 'f16x2' and not np.array_equal(unguarded.embed_crops(crops), w.embed_crops(crops))
    # a checkpoint FILE: the decision is written into the f16x2 repack cache and read back (no second calibration)
    monkeypatch.setenv('TERRAN_HOME', str(tmp_path))
    (tmp_path / 'checkpoints').mkdir()
    from terran_amd import checkpoint
    pth = tmp_path / 'checkpoints' / ('%s.pth' % next(c['id'] for c in checkpoint.CHECKPOINTS if c['kind'] == 'arcface'))
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sw.items()}, pth)
    with pytest.warns(RuntimeWarning):
        f1 = ArcFace(device=0, precision='f16x2')
    assert f1.precision == 'f16x3'
    cached = runtime.packed_program('arcface', None, 'f16x2')
    assert cached.extra.get('f16x2_guard', {}).get('selected') == 'f16x3'
    calls = []
    monkeypatch.setattr(arcface, 'calibration_crops', lambda *a, **k: calls.append(1) or (_ for _ in ()).throw(AssertionError('recalibrated')))
    with pytest.warns(RuntimeWarning):
        f2 = ArcFace(device=0, precision='f16x2')
    assert f2.precision == 'f16x3' and not calls


@pytest.mark.parametrize('threshold', [0.02, 0.3, 0.45, 0.55, 0.6, 0.9, 0.0, 1.0])
def test_retinaface_score_thresholds_vs_oracle(states, threshold):
    """rf_select_kernel skips the softmax of an anchor whose logit margin lies clearly below logit(threshold) (1e-2 of slack)
    and scores the rest exactly: at thresholds far from 0.5 -- where the margin test works on large logit differences --
    and at the ends (0 and 1: no shortcut; everything / nothing passes) the kept sets must still be the oracle's."""
    from oracle import pipeline
    from terran_amd import RetinaFace
    sd = states('retinaface')
    model = RetinaFace(device=0, state=sd, precision='f32')
    dets = ddets = 0
    for k in range(2):
        frames = synth.frames(7000 + k, 4, 208, 277)
        ref = pipeline.retinaface_call(sd, frames, threshold=threshold)
        got = model.call(frames, threshold=threshold)
        for g, r in zip(got, ref):
            g, r = _det_keys(g), _det_keys(r)
            dets += len(r)
            ddets += len(set(g) ^ set(r))
    print('threshold %g: %d detections, %d differ' % (threshold, dets, ddets))
    if threshold <= 0.55:                      # the seeded random detector scores 0.35 .. 0.65
        assert dets > 100
    assert ddets <= max(2, dets // 500), (threshold, dets, ddets)


# ---- weights with trained-looking statistics, refereed by a FLOAT64 evaluation ---------------------------------------
# tests/wild_weights.py: BatchNorm gains / variances and per-channel magnitudes spread over orders of magnitude.  Such
# networks are ill-conditioned enough that two float32 evaluations of one network (the oracle's torch-CPU convs, the
# device's exact-f32 MFMA: different summation orders, BatchNorm applied vs folded) decide a fraction of a percent of the
# near-ties differently, so "flips against the oracle" stops measuring arithmetic quality.  The referee here is the same
# graph evaluated in float64 (post-processing unchanged): every mode -- and the float32 oracle itself -- is counted against it.
def f64_state(sd):
    return {k: (np.asarray(v).astype(np.float64) if np.asarray(v).dtype.kind == 'f' else np.asarray(v)) for k, v in sd.items()}


def det_keys_f64(sd64, frames):
    """RetinaFace.call with the network evaluated in float64 (heads cast to float32 for the unchanged post-processing)."""
    from oracle import nets, retinaface_post
    H, W = frames.shape[1:3]
    x = torch.from_numpy(np.ascontiguousarray(frames)).to(torch.float64).permute(0, 3, 1, 2).flip(1).contiguous()
    outs = [o.numpy().astype(np.float32) for o in nets.retinaface_forward(sd64, x)]
    return [_det_keys(d) for d in retinaface_post.postprocess(outs, H, W, 0.5, 0.4)]


def pose_sets_f64(sd64, frames, short):
    from oracle import facade, nets, openpose_post
    resized, scale = facade.pose_resize(frames, short)
    x = torch.from_numpy(np.transpose(resized, (0, 3, 1, 2)).astype(np.float64) / 255.0 - 0.5)
    pafs, hms = nets.openpose_forward(sd64, x)
    paf_up = openpose_post.bicubic_x8(pafs.numpy().astype(np.float32), 'torch')
    hm_up = openpose_post.bicubic_x8(hms.numpy().astype(np.float32), 'torch')
    out = []
    for i in range(len(frames)):
        dbg = {}
        humans = openpose_post.group_image(hm_up[i], paf_up[i], scale, dbg)
        peaks = {(p, int(y), int(x_)) for p in range(18) for y, x_ in dbg['peaks'][p][0]}
        conns = set()
        for limb, cl in enumerate(dbg['connections']):
            if cl is None:
                continue
            ks, kd = openpose_post.LIMBSEQ[limb][0] - 1, openpose_post.LIMBSEQ[limb][1] - 1
            ls, ld = dbg['peaks'][ks][0], dbg['peaks'][kd][0]
            for (a, b, _) in cl:
                conns.add((limb,) + tuple(int(v) for v in ls[a]) + tuple(int(v) for v in ld[b]))
        out.append((peaks, conns, [h['keypoints'].tobytes() for h in humans]))
    return out


def det_count(got, ref):
    tot = dict(dets=0, ddets=0, images_reordered=0)
    for g, r in zip(got, ref):
        tot['dets'] += len(r)
        tot['ddets'] += len(set(g) ^ set(r))
        tot['images_reordered'] += int(set(g) == set(r) and g != r)
    return tot


def pose_count(got, ref):
    tot = dict(peaks=0, dpeaks=0, conns=0, dconns=0, humans=0, dhumans=0)
    for (gp, gc, gh), (rp, rc, rh) in zip(got, ref):
        tot['peaks'] += len(rp)
        tot['dpeaks'] += len(gp ^ rp)
        tot['conns'] += len(rc)
        tot['dconns'] += len(gc ^ rc)
        tot['humans'] += len(rh)
        tot['dhumans'] += len(set(gh) ^ set(rh))
    return tot


WILD_DET = {'208x277': (lambda k: synth.frames(1000 + k, BATCH, 208, 277), N_WILD),
            '416x739': (lambda k: synth.frames(6000 + k, BATCH, 416, 739), N_WORK)}
WILD_POSE = {'random 96x128': ('wild_openpose', lambda k: synth.frames(2000 + k, BATCH, 96, 128), 96, N_WILD),
             'people 96x128': ('wild_openpose_decoder', lambda k: synth.pose_code_frames(3000 + k, BATCH, 96, 128, 3), 96, N_WILD),
             'random 184x327': ('wild_openpose', lambda k: synth.frames(4000 + k, BATCH, 184, 327), 184, N_WORK),
             'people 184x327': ('wild_openpose_decoder', lambda k: synth.pose_code_frames(5000 + k, BATCH, 184, 327, 4), 184, N_WORK)}


def wild_table(states, modes=('f32', 'f16x3'), log=print):
    """-> {'detector' / 'pose': {who: flips against the float64 evaluation}}, who = 'oracle_f32' or a device mode; every row also
    carries the device's TA_E_RANGE fallbacks.  Shared with tests/probe_wild_weights.py (which prints it per case)."""
    from oracle import pipeline
    from terran_amd import OpenPose, RetinaFace
    tot = {'detector': {}, 'pose': {}}

    def add(task, who, cnt, fallbacks=0):
        t = tot[task].setdefault(who, dict(decisions=0, flips=0, range_fallbacks=0))
        t['decisions'] += cnt.get('dets', 0) + cnt.get('peaks', 0) + cnt.get('conns', 0) + cnt.get('humans', 0)
        t['flips'] += cnt.get('ddets', 0) + cnt.get('dpeaks', 0) + cnt.get('dconns', 0) + cnt.get('dhumans', 0)
        t['range_fallbacks'] += fallbacks
    sd = states('wild_retinaface')
    sd64 = f64_state(sd)
    for case, (gen, n) in WILD_DET.items():
        ref, truth = [], []
        for k in range(0, n, BATCH):
            ref += [_det_keys(d) for d in pipeline.retinaface_call(sd, gen(k))]
            truth += det_keys_f64(sd64, gen(k))
        c = det_count(ref, truth)
        add('detector', 'oracle_f32', c)
        log('  retinaface %s oracle (f32)  vs f64: %s' % (case, c))
        for mode in modes:
            model = RetinaFace(device=0, state=sd, precision=mode)
            got = []
            for k in range(0, n, BATCH):
                got += [_det_keys(g) for g in model.call(gen(k))]
            c = det_count(got, truth)
            add('detector', mode, c, model.fallbacks)
            log('  retinaface %s device %-5s vs f64: %s   vs oracle: %s   range fallbacks %d' % (case, mode, c, det_count(got, ref), model.fallbacks))
    for case, (sd_name, gen, short, n) in WILD_POSE.items():
        sd = states(sd_name)
        sd64 = f64_state(sd)
        ref, truth = [], []
        for k in range(0, n, BATCH):
            ref += _pose_sets_oracle(sd, gen(k), short)
            truth += pose_sets_f64(sd64, gen(k), short)
        c = pose_count(ref, truth)
        add('pose', 'oracle_f32', c)
        log('  openpose %s oracle (f32)  vs f64: %s' % (case, c))
        for mode in modes:
            model = OpenPose(device=0, short_side=short, state=sd, precision=mode)
            got = []
            for k in range(0, n, BATCH):
                got += _pose_sets_device(model, gen(k))
            c = pose_count(got, truth)
            add('pose', mode, c, model.fallbacks)
            log('  openpose %s device %-5s vs f64: %s   vs oracle: %s   range fallbacks %d' % (case, mode, c, pose_count(got, ref), model.fallbacks))
    return tot


def test_wild_weights_decisions_vs_float64(states):
    """On weights with trained-looking statistics the default mode (f16x3, per-channel activation scales, per-output-channel
    weight exponents) decides as close to the FLOAT64 evaluation of the network as the exact-f32 MFMA mode does, never needs
    the exact-f32 fallback, and both stay within a percent of all decisions."""
    tot = wild_table(states)
    for task, rows in tot.items():
        for who, t in rows.items():
            _record('wild_' + task, who, t)
        print('wild weights, %s, flips against the float64 evaluation: %s' % (task, rows))
        f32, head, orc = rows['f32'], rows[HEADLINE], rows['oracle_f32']
        assert f32['decisions'] > 5000
        assert head['range_fallbacks'] == 0 and f32['range_fallbacks'] == 0, rows
        assert f32['flips'] <= f32['decisions'] // 100 and orc['flips'] <= orc['decisions'] // 100, rows
        # counts of rare events on near-ties: the exact-f32 mode's own count with half of it and two standard deviations of
        # slack (5 x the frames: detector 178 vs 194, pose 329 vs 256 -- connection flips come in clusters; tests/probe_map_error.py: the maps
        # themselves are as close to float64 in f16x3 as in f32)
        assert head['flips'] <= rare_bound(f32['flips']), rows


def test_pooled_benign_flip_count():
    """All benign-weight cases of this file pooled: the default mode's near-tie flips against the oracle do not exceed the
    exact-f32 MFMA mode's by more than two over ~100 000 decisions (per-case slack does not add up)."""
    keys = ('dpeaks', 'dconns', 'dhumans', 'ddets', 'images_reordered')
    rows = {k: v for k, v in _results.items() if k.startswith(('openpose_', 'retinaface_')) and 'f32' in v and HEADLINE in v}
    if len(rows) < 4:
        pytest.skip('the per-case tests of this file did not run in this session')
    tot = {m: sum(v[m].get(k, 0) for v in rows.values() for k in keys) for m in ('f32', HEADLINE)}
    print('pooled flips vs oracle over %d cases: %s' % (len(rows), tot))
    assert tot[HEADLINE] <= (tot['f32'] + 2 if _SCALE == 1 else rare_bound(tot['f32'])), (tot, rows)

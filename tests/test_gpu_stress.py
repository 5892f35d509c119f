"""-m gpu: long-running behaviour of the path -- steady device memory over many calls with changing batch shapes
and face counts (scratch, pinned staging, frame cache, plan buckets must all reach a bound), and concurrent use of
one GPU from several host threads, each with its own context (the contract of include/terran_amd.h)."""
import threading

import numpy as np
import pytest

from terran_amd import synth

pytestmark = pytest.mark.gpu


def _round(det, rec, est, sizes, seed):
    out = []
    for k, (n, h, w) in enumerate(sizes):
        frames = synth.upscale_for_resize(synth.pose_code_frames(seed + k, n, 48, 64, 2), h, w)
        dets = det(frames)
        faces = [d[:1 + (k + i) % 3] for i, d in enumerate(dets)]            # 1..3 faces per frame: the face count moves
        out.append((dets, rec(list(frames), faces), est(frames)))
    return out


def test_device_memory_reaches_a_bound(states):
    import torch
    from terran_amd import Detection, Recognition, Estimation
    det = Detection(short_side=96, device=0, state=states('retinaface'))
    rec = Recognition(device=0, state=states('arcface'))
    est = Estimation(short_side=48, device=0, state=states('openpose_decoder'))
    sizes = [(2, 240, 320), (3, 192, 256), (1, 288, 384), (4, 240, 320), (2, 144, 192)]
    for i in range(3):                                   # warm-up: plans, scratch, caches grow to their steady size
        _round(det, rec, est, sizes, 10)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for i in range(12):
        _round(det, rec, est, sizes, 10 + i)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    grown = (free0 - free1) / 2 ** 20
    print('device memory after 12 more rounds of 5 shapes: %+.1f MiB' % grown)
    assert grown < 64.0


def test_three_host_threads_three_contexts_one_gpu(states):
    """Every thread owns a context (stream + scratch) and its own models; results equal the single-thread results."""
    from terran_amd import Detection, Recognition, Estimation, runtime
    sizes = [(2, 240, 320), (3, 192, 256)]

    def build(ctx_dev):
        return (Detection(short_side=96, device=ctx_dev, state=states('retinaface')),
                Recognition(device=ctx_dev, state=states('arcface')),
                Estimation(short_side=48, device=ctx_dev, state=states('openpose_decoder')))
    want = _round(*build(0), sizes, 40)
    assert sum(len(p) for _, _, poses in want for p in poses) > 0
    results, errors = [None] * 3, []

    def work(k):
        try:
            # device=[0]: the device-list form gives this thread a private replica (own context, weights, stream)
            models = build([0])
            for _ in range(4):
                results[k] = _round(*models, sizes, 40)
        except Exception as e:                                   # surfaced below: a thread must not die silently
            errors.append(e)
    threads = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for got in results:
        assert got is not None
        for (d0, f0, p0), (d1, f1, p1) in zip(want, got):
            assert [[tuple(x['bbox']) for x in d] for d in d0] == [[tuple(x['bbox']) for x in d] for d in d1]
            assert all(np.array_equal(a, b) for a, b in zip(f0, f1))
            assert [[x['keypoints'].tolist() for x in p] for p in p0] == [[x['keypoints'].tolist() for x in p] for p in p1]

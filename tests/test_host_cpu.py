"""CPU tests (no GPU): C-ABI surface, host-side logic of the wrappers, sharding over gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from tests.util import REPO


def test_abi_exports_every_declared_symbol():
    """libterran_amd.so must export every function include/terran_amd.h declares, and the ctypes
    table must bind exactly that set (no compute call: there is no GPU here)."""
    from terran_amd import build, lib
    path = build.build()
    hdr = open(os.path.join(REPO, 'include', 'terran_amd.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(ta_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 30
    so = ctypes.CDLL(path)
    for name in declared:
        assert hasattr(so, name), 'missing ABI symbol %s' % name
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    lib.load()
    so.ta_version.restype = ctypes.c_char_p
    assert b'terran_amd' in so.ta_version()


def test_stale_library_is_refused(monkeypatch):
    """The .so travels prebuilt to the GPU box: loading one that was not built from the sources beside it must fail
    loudly (no silent run of an old kernel), and there is no CPU fallback behind it."""
    from terran_amd import build, lib
    build.build()
    assert open(build.STAMP).read().strip() == build.source_hash()
    monkeypatch.setattr(lib, '_lib', None)
    monkeypatch.setattr(build, 'source_hash', lambda: 'edited-after-the-build')
    with pytest.raises(lib.TerranAmdError, match='not built from the sources'):
        lib.load()


def test_no_gpu_fails_loudly():
    """No CPU fallback: without a gfx950 device the product path raises (unless a GPU is present)."""
    from terran_amd import lib
    l = lib.load()
    if l.ta_device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(lib.TerranAmdError):
        lib.Context(0)
    from terran_amd import RetinaFace
    with pytest.raises(lib.TerranAmdError):
        RetinaFace(device=0, state={})


def test_product_never_imports_oracle():
    pat = r'^\s*(from|import)\s+oracle'
    for sub in ('terran_amd', 'tools', 'profiles'):
        for root, _, files in os.walk(os.path.join(REPO, sub)):
            for f in files:
                if f.endswith('.py'):
                    src = open(os.path.join(root, f)).read()
                    assert not re.search(pat, src, flags=re.M), f
    # bench.py: only the cpu_baseline leg; __graft_entry__.py: only smoke()
    bench = open(os.path.join(REPO, 'bench.py')).read()
    head, tail = bench.split('def cpu_baseline(', 1)
    assert not re.search(pat, head, flags=re.M) and re.search(pat, tail, flags=re.M)
    entry = open(os.path.join(REPO, '__graft_entry__.py')).read()
    head, tail = entry.split('def smoke(', 1)
    assert not re.search(pat, head, flags=re.M) and re.search(pat, tail, flags=re.M)


def test_align_matrix_matches_umeyama():
    from oracle import arcface_pre
    from terran_amd import arcface, synth
    for lm in synth.landmarks(3, 20, 480, 640):
        np.testing.assert_allclose(arcface.align_matrix(lm), arcface_pre.align_matrix(lm), rtol=1e-9, atol=1e-9)
    # reflected landmarks (det < 0 branch of Umeyama): still a proper rotation
    lm = synth.landmarks(5, 1, 200, 200)[0]
    lm[:, 0] = 200 - lm[:, 0]
    np.testing.assert_allclose(arcface.align_matrix(lm), arcface_pre.align_matrix(lm), rtol=1e-8, atol=1e-8)


def test_pads_match_reference_rule():
    from oracle import facade as of
    from terran_amd import facade
    shapes = [(5, 7), (8, 4), (8, 7), (3, 3)]
    mh, mw, pads = facade._pads(shapes)
    imgs = [np.full(s + (3,), i + 1, np.uint8) for i, s in enumerate(shapes)]
    padded, params = of.merge_in(imgs)
    assert (mh, mw) == padded.shape[1:3]
    for p, q in zip(pads, params['pads_per_image']):
        assert (p[0], p[1]) == (q[0], q[1])


def test_pack_programs_are_well_formed(states):
    from terran_amd import pack
    for kind, fn in (('openpose', pack.pack_openpose), ('arcface', pack.pack_arcface),
                     ('retinaface', pack.pack_retinaface)):
        P = fn(states(kind))
        blob = P.blob()
        hdr = np.frombuffer(blob[:128], pack.HEADER_DT)[0]
        assert hdr['magic'] == pack.MAGIC and hdr['n_ops'] == len(P.ops)
        assert hdr['weights_off'] % 256 == 0 and hdr['weights_off'] + hdr['weights_bytes'] == len(blob)
        ops = np.frombuffer(blob[hdr['ops_off']:hdr['ops_off'] + pack.OP_DT.itemsize * len(P.ops)], pack.OP_DT)
        conv = ops[(ops['type'] == pack.OP_CONV) | (ops['type'] == pack.OP_DWPW)]
        assert np.all(conv['coutp'] % 32 == 0) and np.all(conv['cin'] % 4 == 0) and np.all(conv['n_slabs'] > 0)
    # algorithmic MACs match SURVEY.md Appendix A (conv + linear, per image / crop)
    def macs(P, h, w):
        # walk shapes like the C++ planner does
        dims = {P.input_tensor: (h, w)}
        total = 0.0
        for op in P.ops:
            ih, iw = dims.get(op['in'], (1, 1)) if P.tensors[op['in']][2] < 0 else (1, 1)
            if op['type'] in (pack.OP_CONV, pack.OP_DWCONV):
                oh = (ih + 2 * op['pad'] - op['kh']) // op['stride'] + 1
                ow = (iw + 2 * op['pad'] - op['kw']) // op['stride'] + 1
                total += op['macs_per_pixel'] * oh * ow
                if op['pool']:                                      # 2x2 max-pool fused into the conv's epilogue
                    oh, ow = oh // 2, ow // 2
            elif op['type'] == pack.OP_RFSTEM:                     # conv3x3 s2 + depthwise (8 ch) + 1x1, one op
                oh, ow = (ih + 1) // 2, (iw + 1) // 2
                total += (8 * 27 + 16 * 8 + 72) * oh * ow
                if op['cout'] == 32:                                # ... fused with the next block: depthwise s2 (16 ch) + 1x1 16 -> 32
                    oh, ow = (oh + 1) // 2, (ow + 1) // 2
                    total += (32 * 16 + 144) * oh * ow
            elif op['type'] == pack.OP_DWPW:                       # depthwise 3x3 (stride) + 1x1, one op
                oh, ow = (ih - 1) // op['stride'] + 1, (iw - 1) // op['stride'] + 1
                total += (op['macs_per_pixel'] + 9 * op['cin']) * oh * ow
            elif op['type'] == pack.OP_MAXPOOL:
                oh, ow = ih // 2, iw // 2
            else:
                oh, ow = ih, iw
            dims[op['out']] = (oh, ow)
            if op['out2'] >= 0:
                dims[op['out2']] = (oh, ow)
        return total
    assert abs(macs(pack.pack_openpose(states('openpose')), 368, 656) / 242.32e9 - 1) < 2e-3
    assert abs(macs(pack.pack_arcface(states('arcface')), 112, 112) / 12.090e9 - 1) < 2e-3
    assert abs(macs(pack.pack_retinaface(states('retinaface')), 640, 640) / 981.1e6 - 1) < 5e-3             # fused program
    assert abs(macs(pack.pack_retinaface(states('retinaface'), fused=False), 640, 640) / 981.1e6 - 1) < 5e-3


def test_shard_bounds():
    from terran_amd import shard
    for n in (0, 1, 7, 32, 33):
        for world in (1, 2, 4, 8):
            spans = [shard.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
from terran_amd import shard
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
frames = [np.full((4, 4, 3), i, np.uint8) for i in range(11)]
def fn(fs):   # variable-length per-frame results, like detections
    return [[{'id': int(f[0, 0, 0]), 'k': k} for k in range(int(f[0, 0, 0]) %% 3)] for f in fs]
res = shard.run_sharded(frames, fn, dist)
if dist.get_rank() == 0:
    assert res == fn(frames), res          # 2-way result == 1-way result, order preserved
    print('SHARD_OK')
else:
    assert res is None
dist.barrier()
dist.destroy_process_group()
'''


def test_sharded_gather_gloo_world2(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % REPO)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'SHARD_OK' in outs[0]


_STREAM_WORKER = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
from terran_amd import shard
rank = int(os.environ['RANK'])
dist.init_process_group('gloo', rank=rank, world_size=int(os.environ['WORLD_SIZE']))
K = 7                                          # steps per rank; 3 per message -> messages of 3, 3, 1 steps
sg = shard.StreamedGather(dist, steps_per_message=3)
for i in range(K):
    time.sleep(0.01 * (1 + 3 * rank))          # the ranks run at different speeds: the collectives still pair up
    sg.put((rank, i, np.full(2 + i, 10 * rank + i, np.int32)))
res = sg.finish()
if rank == 0:
    assert sg.messages == 3 and sg.steps == 2 * K, (sg.messages, sg.steps)
    flat = [(r, i, a.tolist()) for msg in res for (r, i, a) in msg]
    want = []
    for lo, hi in ((0, 3), (3, 6), (6, 7)):    # per message: rank 0's steps, then rank 1's, each in step order
        for r in (0, 1):
            want += [(r, i, [10 * r + i] * (2 + i)) for i in range(lo, hi)]
    assert flat == want, flat
    print('STREAM_OK')
else:
    assert res == [] and sg.steps == 0 and sg.messages == 3
single = shard.StreamedGather(None, steps_per_message=2)          # no process group: everything stays local
for i in range(3):
    single.put(i)
assert single.finish() == [[0, 1], [2]] and single.steps == 3
dist.barrier()
dist.destroy_process_group()
'''


def test_streamed_gather_gloo_world2(tmp_path):
    """shard.StreamedGather (bench.py's ingest leg: per-step results travel to rank 0 while the run goes on): two gloo ranks of
    different speed, 7 steps each in messages of 3 -- rank 0 ends up with every step of both ranks, rank-ordered inside a message."""
    script = tmp_path / 'worker.py'
    script.write_text(_STREAM_WORKER % REPO)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29534', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'STREAM_OK' in outs[0]


def test_raw_video_reader_framing():
    """Batch framing of the rawvideo rgb24 contract (terran/io/video/reader.py:88-117), no GPU: whole batches,
    a short final batch, trailing partial frame dropped, end of stream, prefetch thread shutdown."""
    import io
    from terran_amd import video
    w, h, bs = 8, 6, 4
    frames = np.random.default_rng(0).integers(0, 256, (10, h, w, 3), dtype=np.uint8)
    raw = frames.tobytes() + b'\x01\x02\x03'                      # 10 frames + 3 stray bytes
    got = list(video.RawVideoReader(io.BytesIO(raw), w, h, batch_size=bs, upload=lambda a: a.copy()))
    assert [g.shape[0] for g in got] == [4, 4, 2]
    assert np.array_equal(np.concatenate(got), frames)
    r = video.RawVideoReader(io.BytesIO(frames[:4].tobytes()), w, h, batch_size=bs, upload=lambda a: a.copy())
    assert r.read().shape == (4, h, w, 3)
    with pytest.raises(video.EndOfVideo):
        r.read()
    with pytest.raises(video.VideoClosed):
        next(r)
    assert list(video.RawVideoReader(io.BytesIO(b''), w, h, batch_size=bs, upload=lambda a: a)) == []

    class Boom(io.RawIOBase):
        def readinto(self, b):
            raise OSError('pipe broke')
    with pytest.raises(OSError):
        list(video.RawVideoReader(Boom(), w, h, batch_size=bs, upload=lambda a: a))
    with video.RawVideoReader(io.BytesIO(frames.tobytes() * 50), w, h, batch_size=bs, upload=lambda a: a.copy()) as rr:
        next(rr)                                                   # close mid-stream: the worker must stop
    rr._thread.join(timeout=5)
    assert not rr._thread.is_alive()


def test_pack_cache_roundtrip(tmp_path, states, monkeypatch):
    """Checkpoint file -> packed program -> cache file -> identical blob (the registry/CLI integration row)."""
    import torch
    from terran_amd import runtime, pack, checkpoint
    monkeypatch.setenv('TERRAN_HOME', str(tmp_path))
    (tmp_path / 'checkpoints').mkdir()
    sd = states('retinaface')
    pth = tmp_path / 'checkpoints' / 'b5d77fff.pth'               # the reference's id for RetinaFace
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, pth)
    assert checkpoint.find_checkpoint_file('retinaface') == pth
    p1 = runtime.packed_program('retinaface', None, 'bf16x3')
    cached = list((tmp_path / 'checkpoints').glob('*.tam'))
    assert len(cached) == 1
    p2 = runtime.packed_program('retinaface', None, 'bf16x3')     # served from the cache
    assert p2.blob() == p1.blob() and p2.names == p1.names and p2.kind == pack.MODEL_RETINAFACE
    assert p1.blob() == pack.pack_retinaface(sd, 'bf16x3').blob()
    p3 = runtime.packed_program('retinaface', None, 'f32')        # other precision: its own cache entry ...
    assert len(list((tmp_path / 'checkpoints').glob('*.tam'))) == 2
    assert p3.blob() == p1.blob()                                 # ... but the detector runs exact-f32 in both parity modes
    assert pack.pack_retinaface(sd, 'bf16').blob() != p1.blob()
    sa = states('arcface')
    assert pack.pack_arcface(sa, 'f32').blob() != pack.pack_arcface(sa, 'bf16x3').blob()
    with pytest.raises(ValueError):
        runtime.resolve_state('openpose', None)                   # no file, no synthetic fallback


def test_fanout_shards_contiguously_and_keeps_order(monkeypatch):
    """facade._Fanout (single-process multi-device) with stand-in replicas: contiguous shards in device
    order, sizes differing by at most one, aligned per-item arguments sharded alike, keyword arguments passed on,
    devices beyond the batch size left idle, one private context per replica."""
    from terran_amd import facade, runtime
    made = []
    monkeypatch.setattr(runtime, 'get_context', lambda d=None: ('shared', d))
    monkeypatch.setattr(runtime, 'new_context', lambda d=None: ('new', d))
    calls = []

    def make(dev, ctx):
        made.append((dev, ctx))

        def rep(items, *per, **kw):
            calls.append((dev, ctx, list(items), [list(p) for p in per], kw))
            return [(x, ctx) for x in items]
        return rep
    fo = facade._Fanout([0, 1, 0], make)
    assert made == [(0, ('new', 0)), (1, ('new', 1)), (0, ('new', 0))]            # every replica owns its context
    out = fo(list(range(8)), list('abcdefgh'), _canvas=(3, 4))
    assert [x for x, _ in out] == list(range(8))                                       # frame order kept
    got = sorted(calls, key=lambda c: c[2][0])
    assert [c[2] for c in got] == [[0, 1, 2], [3, 4, 5], [6, 7]]                       # contiguous, sizes differ by <= 1
    assert [c[3] for c in got] == [[list('abc')], [list('def')], [list('gh')]]
    assert all(c[4] == {'_canvas': (3, 4)} for c in got)
    calls.clear()
    assert [x for x, _ in fo(np.arange(2))] == [0, 1] and len(calls) == 2              # third device idle
    with pytest.raises(ValueError):
        facade._Fanout([], make)


def test_fold_input_affine_equals_affine_then_zero_padded_conv():
    """pack.fold_input_affine (ArcFace's BatchNorm in front of a zero-padded conv, arcface/model.py:12-14, folded into
    weights + border-class biases) against the reference order -- affine first, THEN zero padding -- in float64, on maps
    with every border class: ordinary, one row, one column, a single pixel."""
    import torch
    import torch.nn.functional as F
    from terran_amd import pack
    rng = np.random.default_rng(3)
    W, b = rng.normal(0, 0.3, (5, 4, 3, 3)), rng.normal(0, 0.2, 5)
    sc, sh = rng.uniform(0.5, 1.5, 4), rng.normal(0, 0.7, 4)
    Wf, b16 = pack.fold_input_affine(W, b, sc, sh)
    assert b16.shape == (16, 5)
    for h, w in ((6, 7), (1, 5), (4, 1), (1, 1), (2, 2), (3, 3)):
        x = rng.normal(0, 1, (2, 4, h, w))
        ref = F.conv2d(torch.from_numpy(x * sc[None, :, None, None] + sh[None, :, None, None]), torch.from_numpy(W),
                       torch.from_numpy(b), padding=1).numpy()
        raw = F.conv2d(torch.from_numpy(x), torch.from_numpy(Wf), None, padding=1).numpy()
        cy = np.array([3 if h == 1 else (0 if y == 0 else (2 if y == h - 1 else 1)) for y in range(h)])
        cx = np.array([3 if w == 1 else (0 if v == 0 else (2 if v == w - 1 else 1)) for v in range(w)])
        cls = 4 * cy[:, None] + cx[None, :]
        got = raw + np.transpose(b16[cls], (2, 0, 1))[None]
        assert np.abs(got - ref).max() < 1e-12, (h, w)


def test_split_f16_rows_carry_22_bits_and_a_power_of_two_scale():
    """pack.split_f16_rows: every OUTPUT CHANNEL's weights x 2^s[co] as [hi | lo] half floats; (hi + lo) 2^-s[co] reproduces
    every weight within 2^-16 of its ROW's largest to 2^-21 relative whatever the other rows' magnitudes are (a BatchNorm with
    a small gamma / sigma folded in gives a row whose weights are ALL far below the layer's maximum: with one exponent per
    layer it kept ~17 bits), nothing overflows; and the detector's mixed-mode program gives the refiner's 64-channel
    tensors the half-split format while everything the float32 base touches stays float32."""
    from terran_amd import pack, weights
    rng = np.random.default_rng(5)
    w = (rng.normal(0, 0.04, (3, 64, 32)) * np.exp(rng.normal(0, 4.0, (1, 64, 1)))).astype(np.float32)   # rows 2^+-12 apart
    assert np.abs(w).max((0, 2)).max() / np.abs(w).max((0, 2)).min() > 2.0 ** 16
    rows, s = pack.split_f16_rows(w)
    assert s.shape == (64,)
    h16 = rows.view(np.float16).reshape(3, 64, 64)
    hi, lo = h16[..., :32].astype(np.float64), h16[..., 32:].astype(np.float64)
    rowmax = np.abs(w).max((0, 2))
    assert np.isfinite(hi).all() and np.abs(hi).max() < 2.0 ** 15
    assert ((2.0 ** 13 <= rowmax * 2.0 ** s) & (rowmax * 2.0 ** s < 2.0 ** 14)).all()
    back = (hi + lo) * 2.0 ** -s[None, :, None]
    big = np.abs(w) >= rowmax[None, :, None] * 2.0 ** -16
    assert big.mean() > 0.99
    assert (np.abs(back - w)[big] <= np.abs(w)[big] * 2.0 ** -21).all()
    assert (np.abs(back - w).max((0, 2)) <= rowmax * 2.0 ** -23).all()      # absolute error per row: one rounding of hi + lo at the top of ITS range
    P = pack.pack_retinaface(weights.make_retinaface_state(), 'f16x3')
    fmt = P.tensor_formats()
    precs = {}
    for op in P.ops:
        precs.setdefault(op['type'], set()).add(op['prec'])
    assert precs[pack.OP_CONV] == {3} and precs[pack.OP_DWPW] == {0, 3} and precs[pack.OP_RFSTEM] == {0}
    split = [t for t, f in enumerate(fmt) if f == pack.FMT_SPLIT16]
    assert sorted(P.tensors[t][0] for t in split) == [64] * 5 + [256]        # p32, s16, p16, s8, p8 + the stride-32 feature (read by its lateral only)
    assert set(pack.pack_retinaface(weights.make_retinaface_state(), 'bf16x3').tensor_formats()) == {pack.FMT_F32}


def test_c_result_builder_yields_the_comprehensions_objects():
    """terran_amd.results.detections (csrc/pyresults.c when built, else the comprehension): the reference's list[N] of
    list[{'bbox', 'landmarks', 'score'}] (retinaface/wrapper.py:228-236) as row VIEWS into the packed arrays + numpy
    float32 scalars -- same keys in the same order, same dtypes, same aliasing as the Python comprehension."""
    from terran_amd import results
    rng = np.random.default_rng(2)
    counts = np.array([3, 0, 5, 1], np.int32)
    T = int(counts.sum())
    boxes = rng.random((T + 2, 4)).astype(np.float32)
    lmks = rng.random((T + 2, 5, 2)).astype(np.float32)
    scores = rng.random(T + 2).astype(np.float32)
    for b, l in ((boxes, lmks), (np.around(boxes * 50).astype(np.int32), np.around(lmks * 50).astype(np.int32))):
        a = results.detections_py(counts, b[:T], l[:T], scores[:T])
        c = results.detections(counts, b[:T], l[:T], scores[:T])
        assert [len(x) for x in c] == [3, 0, 5, 1] == [len(x) for x in a]
        for p, q in zip(a, c):
            for x, y in zip(p, q):
                assert list(y) == ['bbox', 'landmarks', 'score'] == list(x)
                assert y['bbox'].shape == (4,) and y['landmarks'].shape == (5, 2) and y['bbox'].dtype == b.dtype
                assert np.array_equal(x['bbox'], y['bbox']) and np.array_equal(x['landmarks'], y['landmarks'])
                assert type(y['score']) is np.float32 and y['score'] == x['score']
        c[2][1]['bbox'][0] = 77                                   # a view into the packed array, like the comprehension's rows
        assert b[4, 0] == 77
    if results._pyresults is not None:
        with pytest.raises(ValueError):
            results.detections_eager(counts.astype(np.int64), boxes, lmks, scores)
        with pytest.raises(ValueError):
            results.detections_eager(np.array([T + 3], np.int32), boxes, lmks, scores)


def test_detector_lanes_are_closed_branches(monkeypatch):
    """pack.pack_retinaface marks the context module + heads of the stride-32 / 16 levels as lanes 1 / 2 (side streams,
    ta_op_desc.variant bits 17..18).  What net.hip's loader enforces is checked here on the packed program too: a lane's ops
    follow the op that finishes their input, write tensors nobody outside the lane touches, and read nothing that a later
    op outside the lane writes; TERRAN_AMD_NO_DETECTOR_LANES packs the same ops without lanes."""
    from terran_amd import pack, weights
    sd = weights.make_retinaface_state()
    P = pack.pack_retinaface(sd, 'f16x3')
    lanes = [(op['variant'] >> 17) & 3 for op in P.ops]
    assert set(lanes) == {0, 1, 2} and lanes.count(1) == 4 and lanes.count(2) == 4
    written_by = {}
    for i, op in enumerate(P.ops):
        for t in (op['out'], op.get('out2', -1)):
            if t >= 0:
                written_by.setdefault(t, set()).add(lanes[i])
    for i, op in enumerate(P.ops):
        L = lanes[i]
        for t in (op['in'], op.get('res', -1)):
            if t < 0 or t not in written_by:
                continue
            assert written_by[t] <= {0, L}, (i, t)                 # reads its own lane's tensors or the main stream's
            if L and 0 in written_by[t]:                           # ... and those were finished BEFORE the lane's first op
                first = lanes.index(L)
                assert all(j < first for j, o in enumerate(P.ops) if o['out'] == t and lanes[j] == 0), (i, t)
    for L in (1, 2):                                               # a lane's ops are consecutive
        idx = [i for i, x in enumerate(lanes) if x == L]
        assert idx == list(range(idx[0], idx[0] + 4))
    monkeypatch.setenv('TERRAN_AMD_NO_DETECTOR_LANES', '1')
    Q = pack.pack_retinaface(sd, 'f16x3')
    assert {(op['variant'] >> 17) & 3 for op in Q.ops} == {0} and len(Q.ops) == len(P.ops)
    assert sorted(op['macs_per_pixel'] for op in Q.ops) == sorted(op['macs_per_pixel'] for op in P.ops)


def test_f16_mode_packs_half_float_tensors_and_64_channel_slabs():
    """precision='f16' (single-half embedder): ArcFace's tensors become TA_FMT_F16 (2 bytes per element), every conv that
    reads one walks K in slabs of 64 channels with weights re-packed as [slab][cout][64 halfs] (x 2^wscale), the stem conv
    (float32 crops in) keeps 32-wide slabs, the FC asks for its 32 K ranges itself; the detector / pose packers read the
    mode as 'f16x3'."""
    from terran_amd import pack, weights
    sd = weights.make_arcface_state()
    P = pack.pack_arcface(sd, 'f16')
    blob = P.blob()
    fmt = P.tensor_formats()
    assert fmt[P.input_tensor] == pack.FMT_F32 and fmt[P.outputs[0]] == pack.FMT_F32
    assert sum(f == pack.FMT_F16 for f in fmt) == len(fmt) - 2          # everything but the float32 crops and the embedding
    hdr = np.frombuffer(blob[:pack.HEADER_DT.itemsize], pack.HEADER_DT)[0]
    wreg = blob[int(hdr['weights_off']):]
    for i, op in enumerate(P.ops):
        assert op['prec'] == 4
        taps = op['kh'] * op['kw']
        if fmt[op['in']] == pack.FMT_F16:
            assert op['cin'] % 64 == 0 and op['n_slabs'] == taps * op['cin'] // 64
        else:
            assert i == 0 and op['n_slabs'] == 2                            # the stem: 27 -> 36 -> 2 slabs of 32
    op = P.ops[3]                                                           # a 64 -> 64 3x3 conv on a half-float tensor
    coutp = op['coutp']
    rows = np.frombuffer(wreg[op['w_off']:op['w_off'] + op['n_slabs'] * coutp * 128], np.float16).reshape(op['n_slabs'], coutp, 64)
    rmax = np.abs(rows.astype(np.float32)).max((0, 2))[:op['cout']]
    assert np.array_equal(rows, P._fold[3]['rows64']) and ((2.0 ** 13 <= rmax) & (rmax < 2.0 ** 14)).all()      # per output channel
    # the rows are the folded weights times 2^(s[co] - a_in[c]): the input channels' activation exponents live in the columns
    W = np.asarray(sd['stages.0.0.body.4.weight'], np.float64)
    from terran_amd import arch
    g, b_, m_, v_ = (np.asarray(sd['stages.0.0.body.5.' + k], np.float64) for k in ('weight', 'bias', 'running_mean', 'running_var'))
    Wf = W * (g / np.sqrt(v_ + arch.ARC_BN_EPS))[:, None, None, None]
    a_in, wexp = P.scales[op['in']], P._fold[3]['wexp']
    back = rows.astype(np.float64).reshape(9, 1, coutp, 64)[:, 0].transpose(1, 0, 2)              # [cout][tap][cin]
    want = Wf.transpose(0, 2, 3, 1).reshape(64, 9, 64) * 2.0 ** (wexp[:64, None, None] - a_in[None, None, :])
    assert P.ops[3]['kh'] == 3 and P.ops[3]['stride'] == 2
    assert np.abs(back[:64] - want).max() <= np.abs(want).max() * 2.0 ** -10
    assert (P.ops[-1]['variant'] >> 8) & 255 == 32 and P.ops[-1]['n_slabs'] == 392
    for packer, state in ((pack.pack_retinaface, weights.make_retinaface_state()), (pack.pack_openpose, weights.make_openpose_state())):
        precs = {o['prec'] for o in packer(state, 'f16').ops if o['type'] == pack.OP_CONV}
        assert 4 not in precs and 3 in precs
        assert packer(state, 'f16').blob() == packer(state, 'f16x3').blob()    # the SAME programs: every decision as in f16x3


def test_f16x2_mode_is_the_f16x3_image_with_another_opcode():
    """precision='f16x2' (opt-in, guarded at load: arcface.guard_f16x2): tensors, formats, scales and every weight byte of ArcFace
    are f16x3's -- the kernels just skip the w_hi * x_lo product -- only the ops' `prec` field differs; the detector / pose
    packers read the mode as 'f16x3' (the same programs bit for bit).  The library default is the float32-grade 'f16x3'."""
    from terran_amd import pack, runtime, weights
    sd = weights.make_arcface_state()
    P2, P3 = pack.pack_arcface(sd, 'f16x2'), pack.pack_arcface(sd, 'f16x3')
    assert {op['prec'] for op in P2.ops} == {5} and {op['prec'] for op in P3.ops} == {3}
    assert P2.tensor_formats() == P3.tensor_formats()
    b2, b3 = np.frombuffer(P2.blob(), np.uint8), np.frombuffer(P3.blob(), np.uint8)
    assert len(b2) == len(b3)
    hdr = np.frombuffer(P2.blob()[:pack.HEADER_DT.itemsize], pack.HEADER_DT)[0]
    o0, n = int(hdr['ops_off']), len(P2.ops) * pack.OP_DT.itemsize
    diff = np.nonzero(b2 != b3)[0]
    assert len(diff) == len(P2.ops) and ((diff >= o0) & (diff < o0 + n)).all()      # one byte per op: its arithmetic mode
    ops2 = np.frombuffer(P2.blob()[o0:o0 + n], pack.OP_DT)
    assert set(ops2['prec'].tolist()) == {5}
    for packer, state in ((pack.pack_retinaface, weights.make_retinaface_state()), (pack.pack_openpose, weights.make_openpose_state())):
        assert packer(state, 'f16x2').blob() == packer(state, 'f16x3').blob()
    assert runtime.DEFAULT_PRECISION == 'f16x3' and runtime.resolve_precision('f16x2') == 'f16x2'
    import os
    if not os.environ.get('TERRAN_AMD_PRECISION'):
        assert runtime.resolve_precision() == 'f16x3'


def test_activation_scales_follow_the_expected_magnitudes():
    """pack.Program: per-channel (mean, variance) are propagated through the folded weights (Gaussian moments through ReLU /
    PReLU) and every tensor of a program with half-float convs is stored times 2^a with its expected max |x| 2^a in
    (2^9, 2^10]; consumers / producers get the powers of two folded into their epilogue vectors; programs without half-float
    convs keep every exponent at 0; input, outputs and raw copies obey the loader's rules."""
    from terran_amd import pack, weights
    # (1) moments of a rectified Gaussian against sampling
    rng = np.random.default_rng(0)
    z = rng.normal(0.7, 1.3, 2_000_000)
    m, v = pack.act_moments(np.array([0.7]), np.array([1.3 ** 2]), pack.ACT_RELU)
    assert abs(m[0] - np.maximum(z, 0).mean()) < 3e-3 and abs(v[0] - np.maximum(z, 0).var()) < 6e-3
    y = np.where(z > 0, z, 0.3 * z)
    m, v = pack.act_moments(np.array([0.7]), np.array([1.3 ** 2]), pack.ACT_PRELU, np.array([0.3]))
    assert abs(m[0] - y.mean()) < 3e-3 and abs(v[0] - y.var()) < 6e-3
    # (2) the three networks: scales only where half-float convs exist, rules of the loader
    for kind, sd in (('retinaface', weights.make_retinaface_state()), ('arcface', weights.make_arcface_state()),
                     ('openpose', weights.make_openpose_state())):
        packer = getattr(pack, 'pack_' + kind)
        for prec in ('f32', 'bf16x3'):
            P = packer(sd, prec)
            P.blob()
            assert not any(x.any() for x in P.scales), (kind, prec)
        P = packer(sd, 'f16x3')
        P.blob()
        assert not P.scales[P.input_tensor].any() and all(not P.scales[t].any() for t in P.f32_only)
        used = [t for t in range(len(P.tensors)) if P.expected_amax(t) > 0 and t != P.input_tensor and t not in P.f32_only]
        nz = [t for t in used if P.scales[t].any()]
        assert len(nz) > len(used) // 2, (kind, len(nz), len(used))
        for op in P.ops:
            if op['type'] == pack.OP_MAXPOOL:
                assert np.array_equal(P.scales[op['in']], P.scales[op['out']])
            if op['type'] == pack.OP_COPYCH:
                assert np.array_equal(P.scales[op['in']][op['in_ch_off']:op['in_ch_off'] + op['cin']],
                                      P.scales[op['out']][op['out_ch_off']:op['out_ch_off'] + op['cin']])
            if op['type'] == pack.OP_CONV and op['res'] >= 0:      # a shortcut and the sum it joins: same exponents, channel by channel
                assert np.array_equal(P.scales[op['res']][op['res_ch_off']:op['res_ch_off'] + op['cout']],
                                      P.scales[op['out']][op['out_ch_off']:op['out_ch_off'] + op['cout']])
        for t in nz:
            if P.tensors[t][2] >= 0:
                continue
            a = P.expected_amax(t, per_channel=True) * 2.0 ** P.scales[t]
            assert a.max() <= 2.0 ** 10 * 1.0001, (kind, t, a.max())   # grouped with channels of larger magnitude: may sit lower, never higher
    # (3) gain invariance: scaling a layer's weights by 2^k moves its output tensor's exponent by -k and nothing else
    rng = np.random.default_rng(1)

    def prog(gain):
        P = pack.Program(pack.MODEL_OPENPOSE, 'f16x3')
        t0 = P.tensor(32, 1)
        P.input_tensor = t0
        t1, t2 = P.tensor(64, 1), P.tensor(64, 0, f32=True)
        r = np.random.default_rng(3)
        P.conv(t0, t1, r.normal(0, 0.1, (64, 32, 3, 3)) * gain, r.normal(0, 0.1, 64) * gain, act=pack.ACT_RELU)
        P.conv(t1, t2, r.normal(0, 0.1, (64, 64, 3, 3)) / gain, r.normal(0, 0.1, 64))
        P.outputs = [t2]
        P.blob()
        return P
    a, b = prog(1.0), prog(2.0 ** 7)
    assert np.array_equal(b.scales[1], a.scales[1] - 7) and not b.scales[2].any() and not a.scales[2].any()
    # (4) a wild per-channel spread is absorbed channel by channel: one channel 2^6 above the rest moves ITS exponent only
    def prog2(boost):
        P = pack.Program(pack.MODEL_OPENPOSE, 'f16x3')
        t0 = P.tensor(32, 1)
        P.input_tensor = t0
        t1, t2 = P.tensor(64, 1), P.tensor(64, 0, f32=True)
        r = np.random.default_rng(3)
        W1, b1 = r.normal(0, 0.1, (64, 32, 3, 3)), r.normal(0, 0.1, 64)
        W1[5] *= boost
        b1[5] *= boost
        P.conv(t0, t1, W1, b1, act=pack.ACT_RELU)
        P.conv(t1, t2, r.normal(0, 0.1, (64, 64, 3, 3)), r.normal(0, 0.1, 64))
        P.outputs = [t2]
        P.blob()
        return P
    a, b, c = prog2(1.0), prog2(2.0 ** 6), prog2(2.0 ** 12)
    d = b.scales[1] - a.scales[1]
    assert d[5] == -6 and not np.delete(d, 5).any()
    # ... up to 2^8 (pack._CH_SPREAD): beyond that the rest of the tensor follows (a channel's own bound is a noisy number)
    d = c.scales[1] - a.scales[1]
    assert d[5] == -12 and (np.delete(d, 5) == -(12 - pack._CH_SPREAD)).all()


def test_power_sampler_reads_hwmon_files_and_is_a_noop_without_them(tmp_path):
    """terran_amd.telemetry: bench.py's `power` object (socket power / shader clock over the timed region) comes from the amdgpu
    hwmon files of the GPU's PCI function; a box without them (this container) yields None and never raises."""
    import time
    from terran_amd import telemetry
    assert telemetry.hwmon_of_pci(None) is None and telemetry.hwmon_of_pci('0000:00:00.0') is None
    idle = telemetry.PowerSampler(None).start()
    assert idle.stop() == [] and idle.summary() is None
    for name, v in (('power1_input', 1300000000), ('power1_cap', 1400000000), ('freq1_input', 1900000000),
                    ('temp2_input', 55000), ('temp3_input', 61000)):
        (tmp_path / name).write_text('%d\n' % v)
    row = telemetry.sample(str(tmp_path))
    assert row == {'power_w': 1300.0, 'cap_w': 1400.0, 'sclk_mhz': 1900.0, 'temp_c': 61.0}
    s = telemetry.PowerSampler(str(tmp_path), 0.005).start()
    time.sleep(0.06)
    rows = s.stop()
    assert len(rows) >= 3 and rows[0]['t'] <= rows[-1]['t']
    summ = s.summary(skip_seconds=0.02)
    assert summ['power_w_mean'] == 1300.0 and summ['cap_w'] == 1400.0 and summ['sclk_mhz_min'] == 1900.0
    assert summ['samples'] < len(rows)


def test_lazy_faces_behave_as_the_lists_they_stand_for():
    """results.detections(lazy=True) (opt-in; the default is plain lists) returns per-image `LazyFaces` (a list subclass that
    creates its dicts on first use): every way of looking at one must give what the eager list of dicts gives."""
    import copy
    import json
    import pickle
    from terran_amd import results
    rng = np.random.default_rng(0)
    counts = np.array([3, 0, 5, 1], np.int32)
    T = int(counts.sum())
    boxes, lms, scores = (rng.normal(size=(T, 4)).astype(np.float32), rng.normal(size=(T, 5, 2)).astype(np.float32),
                          rng.uniform(size=T).astype(np.float32))
    want = results.detections_py(counts, boxes, lms, scores)

    def same(a, b):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert set(x) == {'bbox', 'landmarks', 'score'} and x['score'] == y['score'] and isinstance(x['score'], np.float32)
            assert np.array_equal(x['bbox'], y['bbox']) and np.array_equal(x['landmarks'], y['landmarks'])
            assert x['landmarks'].shape == (5, 2)

    def fresh():
        return results.detections(counts, boxes, lms, scores, lazy=True)
    plain = results.detections(counts, boxes, lms, scores)                     # the default: the reference's plain lists
    assert all(type(g) is list for g in plain) and [len(g) for g in plain] == counts.tolist()
    for g, w in zip(plain, want):
        same(g, w)
    # a plain list on the LEFT (CPython's list_concat reads the right operand's storage directly): the reflected slot fills first
    for g, w in zip(fresh(), want):
        same([] + g, w)
        same([w[0]] + g if w else [] + g, ([w[0]] + w) if w else w)
    flat = sum(fresh(), [])                                                    # the flattening idiom over a batch's results
    same(flat, [d for w in want for d in w])
    assert len(sum(fresh(), [])) == T
    got = fresh()
    assert isinstance(got, list) and all(isinstance(g, list) for g in got)
    assert [len(g) for g in got] == counts.tolist() and [bool(g) for g in got] == [True, False, True, True]     # no dict built yet
    assert all(g._src is not None for g in got)
    for g, w in zip(got, want):
        same(g, w)                                                     # iteration
        assert all(d['bbox'].base is not None for d in g)               # views into the packed arrays, not copies
    for mk in (lambda g: g[:2], lambda g: list(g), lambda g: sorted(g, key=lambda d: d['score']), lambda g: g + [],
               lambda g: [] + list(g), lambda g: g.copy(), lambda g: copy.copy(g), lambda g: copy.deepcopy(g),
               lambda g: pickle.loads(pickle.dumps(g)), lambda g: [d for d in reversed(g)][::-1], lambda g: g * 1):
        for g, w in zip(fresh(), want):
            ref = mk(w)
            same(mk(g), ref)
    g = fresh()[2]
    assert g[0]['score'] == want[2][0]['score'] and g[-1]['score'] == want[2][-1]['score']
    a, b = fresh()[0], fresh()[0]
    assert len(a + b) == 6 and len(b) == 3                             # concatenation of two unfilled lists
    a += fresh()[2]
    assert len(a) == 8
    g = fresh()[0]
    g.append({'bbox': None})
    assert len(g) == 4 and g.pop()['bbox'] is None and len(g) == 3
    g.sort(key=lambda d: -d['score'])
    assert [d['score'] for d in g] == sorted((d['score'] for d in want[0]), reverse=True)
    assert repr(fresh()[3]) == repr(want[3]) and fresh()[1] == [] and not (fresh()[1] != [])
    assert json.dumps([[float(d['score']) for d in g] for g in fresh()]) == json.dumps([[float(d['score']) for d in w] for w in want])
    assert results.eager(fresh())[0][1]['score'] == want[0][1]['score'] and type(results.eager(fresh())[0]) is list
    lazy_empty = results.detections(np.zeros(2, np.int32), boxes[:0], lms[:0], scores[:0], lazy=True)
    assert lazy_empty == [[], []] and [len(x) for x in lazy_empty] == [0, 0]


def test_embed_worker_launch_rule(monkeypatch):
    """pipeline._Embedder without a GPU (contexts and the model stubbed): it launches on >= min_crops, it does NOT sit out
    `max_wait` when no shard of its device is between upload and detect (a consumer that feeds one batch at a time), it DOES wait
    for a shard that was announced, and a worker that dies lets go of the frames still queued for it."""
    import queue
    import time
    from terran_amd import affinity, pipeline, runtime

    class Ctx:
        def close(self):
            pass
    monkeypatch.setattr(runtime, 'new_context', lambda d=None: Ctx())
    monkeypatch.setattr(affinity, 'bind', lambda d: {})
    launches = []

    class Model:
        def call_multi(self, items):
            launches.append((time.perf_counter(), sum(sum(len(f) for f in faces) for _, faces in items)))
            if any(fr == 'boom' for fr, _ in items):
                raise RuntimeError('boom')
            return [[np.zeros((len(f), 512)) for f in faces] for _, faces in items]

        def free(self):
            pass

    class Rec:
        model = Model()

    class Refs:
        def __init__(self):
            self.released = 0

        def release(self):
            self.released += 1
    out, errs, live = queue.Queue(), [], {1}
    emb = pipeline._Embedder(0, lambda dev, ctx: Rec(), out, errs.append, live, min_crops=8, max_crops=16, max_wait=0.5)

    def shard(key, n_faces, frames='fr'):
        return (1, key, frames, [[{}] * n_faces], Refs(), n_faces)
    # (1) one small shard, nothing upstream: launched at once, not after max_wait
    t0 = time.perf_counter()
    emb.announce()
    emb.deliver(shard('a', 2))
    assert out.get(timeout=5)[1] == 'a' and time.perf_counter() - t0 < 0.25 and launches[-1][1] == 2
    # (2) a second shard is announced before the first arrives: the worker waits for it and embeds both in ONE launch
    emb.announce()
    emb.announce()
    emb.deliver(shard('b', 3))
    time.sleep(0.1)
    assert out.empty() and len(launches) == 1                    # still waiting: upstream > 0, min_crops not reached
    emb.deliver(shard('c', 6))
    got = {out.get(timeout=5)[1], out.get(timeout=5)[1]}
    assert got == {'b', 'c'} and len(launches) == 2 and launches[-1][1] == 9
    # (3) a shard that would overflow max_crops is carried to the next launch
    for k, n in (('d', 7), ('e', 12)):
        emb.announce()
        emb.deliver(shard(k, n))
    assert {out.get(timeout=5)[1], out.get(timeout=5)[1]} == {'d', 'e'} and [l[1] for l in launches[-2:]] == [7, 12]
    # (4) the worker dies: what is still queued is released, the failure is reported
    emb.announce()
    emb.announce()
    bad, waiting = shard('f', 9, frames='boom'), shard('g', 1)
    with emb.cv:                                                  # both in the inbox before the worker looks
        emb.items.extend([bad, waiting])
        emb.cv.notify()
    deadline = time.perf_counter() + 5
    while not errs and time.perf_counter() < deadline:
        time.sleep(0.01)
    assert errs and bad[4].released == 1
    emb.thread.join(timeout=5)
    assert waiting[4].released == 1
    # (5) an announced shard whose lane failed takes its announcement back (skip): a fresh worker with one shard waiting and one
    # announced launches as soon as the skip arrives, not after max_wait
    out2 = queue.Queue()
    emb2 = pipeline._Embedder(0, lambda dev, ctx: Rec(), out2, errs.append, live, min_crops=8, max_crops=16, max_wait=2.0)
    emb2.announce()
    emb2.announce()
    t0 = time.perf_counter()
    emb2.deliver(shard('h', 2))
    time.sleep(0.05)
    assert out2.empty()
    emb2.skip()
    assert out2.get(timeout=5)[1] == 'h' and time.perf_counter() - t0 < 1.0 and emb2.upstream == 0
    emb2.close()


def test_bench_line_is_compact():
    """bench.py's stdout line is what the driver parses, and the driver keeps the LAST 8 KB of stdout: the line is built by
    `compact_line` from the detail object and must stay under 8192 bytes whatever prose and side legs the detail carries
    (round 5's ~25 KB line came back as `parsed: null`)."""
    import json
    import bench
    prose = 'x' * 3000
    roof = {'kernel': 'conv_igemm_split<2,4,4,3,3>', 'bound': 'mfma', 'achieved': 439.1, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': 0.1756,
            'traffic': 149800000, 'mfma_issue_frac': 0.527, 'launches_per_step': 38, 'avg_launch_ms': 0.2023,
            'algorithmic_gflop_per_launch': 87.19, 'share_of_conv_time': 0.467, 'source': prose, 'traffic_source': prose,
            'all_conv_kernels': {'note': prose, 'pmc_dominant_layers': {'a': prose}}}
    power = {'samples': 40, 'power_w_mean': 1301.2, 'sclk_mhz_mean': 1859.0, 'cap_w': 1400.0, 'what': prose}
    per_model = {}
    for prec in ('f16x3', 'f32'):
        per_model['C2 RetinaFace 32x640x640 ' + prec] = {'images_per_s': 17000.1, 'images_per_s_packed': 27000.5, 'roofline': {'bound': 'hbm', 'unit': 'GB/s', 'peak': 8000.0,
                                                         'achieved': 1551.0, 'frac': 0.194, 'note': prose}}
        per_model['C3 ArcFace 256x3x112x112 ' + prec] = {'images_per_s': 16000.0, 'roofline': {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': 2500.0,
                                                         'achieved': 394.0, 'frac': 0.158}}
        per_model['C4 OpenPose 16x368x656 ' + prec] = {'images_per_s': 800.0, 'roofline': {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': 2500.0,
                                                       'achieved': 430.0, 'frac': 0.172}}
    per_model['C3 ArcFace 256x3x112x112 (embedder in the f16x2 mode)'] = {'images_per_s': 1.0, 'roofline': {'note': prose}}
    detail = {
        'metric': 'frames/sec 1080p detect+embed+pose', 'value': 2267.123, 'unit': 'frames/s', 'n_gpus': 1, 'steps': 144, 'warmup': 6,
        'ms_per_step': 14.115, 'timed_region_s': 2.54, 'timed_steps': 180, 'value_k_steps': 2301.5, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': bench.DTYPES['f16x3'], 'data': 'synthetic',
        'config': {'workload': prose, 'workload_short': 'BASELINE configs[4]: 1080p detect+embed+pose', 'precision': 'f16x3',
                   'frames_per_gpu_step': 32, 'faces_per_frame': 2, 'resident_batch_reused': True, 'detections_per_frame': 358.2,
                   'humans_per_frame': 4.0, 'host_placement_per_rank': [prose] * 8,
                   'host_per_rank': [{'cpu_s_per_step': 0.085, 'threads': 15, 'cores_allowed': 64}] * 8,
                   'batches_in_flight_per_gpu': 4, 'step_overlap': prose},
        'roofline': roof, 'roofline_f32': dict(roof, kernel='conv_igemm_pipe<0>', peak=157.3, frac=0.8),
        'roofline_f16x3': roof, 'stage_ms_per_step': {'conv_igemm': 14.0}, 'power': power, 'power_f32': power,
        'decision_drift': {'vs': prose, 'detections': 11461, 'detections_differ': 0, 'people': 128, 'people_differ': 0,
                           'embedding_max_abs_diff': 1.2e-6},
        'value_f32': 775.2, 'value_f16x2': 2377.0, 'value_ingest': 2512.3,
        'ingest': {'value': 2512.3, 'steps': 110, 'ms_per_step': 12.7, 'gather_tail_s': 0.001, 'gather_messages': 14,
                   'steps_gathered_on_rank0': 110, 'what': prose},
        'other_precisions': {p: {'roofline': roof, 'power': power, 'value': 1.0} for p in ('f32', 'f16x2', 'bf16', 'f16', 'bf16x3')},
        'other_faces_per_frame': {'1': {'value': 1.0}, '4': {'value': 2.0}}, 'per_model': per_model,
        'cpu_baseline': {'value': 1.82, 'unit': 'frames/s', 'cores': 128, 'cpus_allowed': 128, 'kind': 'port', 'sample': prose,
                         'stage_ms_per_frame': {'detection': 1.0}},
    }
    assert len(json.dumps(detail)) > 60000
    line = bench.compact_line(detail, 'gpurun_out/bench_detail.json')
    assert '\n' not in line and len(line) < 8192 and len(line) < 4500, len(line)
    d = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'value_f32', 'value_f16x2', 'value_ingest', 'decision_drift'):
        assert k in d, k
    assert d['value'] == 2267.123 and len(d['dtype']) <= 200 and d['config']['resident_batch_reused'] is True
    assert set(d['roofline']) == {'kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms', 'launches_per_step',
                                  'algorithmic_gflop_per_launch', 'mfma_issue_frac', 'share_of_conv_time'}
    assert set(d['cpu_baseline']) == {'value', 'unit', 'cores', 'kind', 'sample'} and len(d['cpu_baseline']['sample']) <= 160
    assert len(d['per_model']) == 6 and all(set(v) - {'images_per_s_packed'} == {'images_per_s', 'bound', 'achieved', 'unit', 'frac'} for v in d['per_model'].values())
    assert prose[:300] not in line
    # a failed secondary leg (its error string) must not break the line either
    detail['per_model'] = {'error': 'RuntimeError: ' + prose}
    detail['cpu_baseline'] = {'error': 'OSError: ' + prose}
    assert len(bench.compact_line(detail)) < 8192


def test_smi_counter_deltas():
    """telemetry.SmiCounters.delta: joules, mean power and per-throttler residency (% of the region) from two readings of the SMI
    accumulators; missing fields stay absent, a wrapped energy counter yields no energy figure."""
    from terran_amd import telemetry
    a = {'t': 10.0, 'energy_uj': 5.0e9, 'accumulation_counter': 1000, 'ppt_residency_acc': 100, 'socket_thm_residency_acc': 7, 'hbm_thm_residency_acc': 0}
    b = {'t': 12.5, 'energy_uj': 8.25e9, 'accumulation_counter': 3000, 'ppt_residency_acc': 1700, 'socket_thm_residency_acc': 7, 'hbm_thm_residency_acc': 0,
         'throttle_status': 4}
    d = telemetry.SmiCounters.delta(a, b)
    assert d['seconds'] == 2.5 and d['energy_j'] == 3250.0 and d['power_w_from_energy'] == 1300.0
    assert d['throttle_residency_pct'] == {'ppt': 80.0, 'socket_thm': 0.0, 'hbm_thm': 0.0} and d['throttle_status'] == 4
    assert telemetry.SmiCounters.delta(None, b) is None and telemetry.SmiCounters.delta(a, None) is None
    wrapped = telemetry.SmiCounters.delta(dict(a, energy_uj=9e9), b)
    assert 'energy_j' not in wrapped and wrapped['throttle_residency_pct']['ppt'] == 80.0
    assert telemetry.SmiCounters.delta({'t': 0.0}, {'t': 1.0}) is None                 # nothing but time stamps: no reading
    assert telemetry.SmiCounters(None).read() is None                                  # no PCI address: the probe is a no-op
    s = telemetry.PowerSampler(None, bdf=None).start()
    assert s.stop() == [] and s.summary() is None

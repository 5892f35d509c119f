import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import RetinaFace, synth, weights, runtime
from oracle import pipeline
sd = weights.make_retinaface_state()
def keys(dets): return [tuple(np.rint(d['bbox']).astype(int).tolist()) for d in dets]
det = RetinaFace(device=0, state=sd, precision='f16x3')
print('program ops', det.model.kind)
tot = dict(images=0, dets=0, ddets=0, reordered=0, swapped=0)
worst = 0.0
for case, n in (((208, 277), 208), ((416, 739), 16)):
    for k in range(0, n, 16):
        fr = synth.frames((1000 if case[0] == 208 else 6000) + k, 16, *case)
        ref = pipeline.retinaface_call(sd, fr)
        got = det.call(fr)
        for g, r in zip(got, ref):
            kg, kr = keys(g), keys(r)
            tot['images'] += 1; tot['dets'] += len(kr); tot['ddets'] += len(set(kg) ^ set(kr))
            if set(kg) == set(kr) and kg != kr:
                tot['reordered'] += 1; tot['swapped'] += sum(a != b for a, b in zip(kg, kr))
            if kg == kr:
                for a, b in zip(g, r):
                    worst = max(worst, float(np.abs(a['bbox'] - b['bbox']).max()), float(abs(a['score'] - b['score'])))
print(tot, 'worst abs diff on identical lists %.2e' % worst)
ctx = det.ctx
fr = ctx.upload(synth.frames(1, 32, 640, 640))
for _ in range(3): det.detect_arrays(fr)
ctx.sync(); t0 = time.perf_counter()
for _ in range(20): det.detect_arrays(fr)
ctx.sync(); dt = (time.perf_counter() - t0) / 20
ctx.profile_reset(); ctx.profile(True); det.detect_arrays(fr); ctx.sync()
print('C2 packed: %.3f ms per batch, kernels %s' % (dt * 1e3, [tuple(round(x, 3) if isinstance(x, float) else x for x in ctx.profile_read(k)[:2]) for k in range(4)]))

"""rf_select_kernel A/B (run on the GPU box): the detections of 8 frames at 416 x 739 with the NMS on the greedy loop
(TA_RF_FAST_MAX=0) and on the suppression matrix (default) must be identical; TA_RF_DEBUG=1 prints the kernel's phase times
(compaction / sort / decode / matrix / walk) per image on stderr.   python tools/rf_select_probe.py"""
import os, sys, subprocess, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from terran_amd import retinaface, synth, weights, lib
    ctx = lib.Context(0)
    det = retinaface.RetinaFace(device=0, state=weights.make_retinaface_state(), precision='f16x3', ctx=ctx)
    fr = ctx.upload(synth.frames(4, 8, 416, 739))
    counts, boxes, lm, sc = det.detect_arrays(fr)
    pickle.dump((counts, boxes, sc), open(sys.argv[1], 'wb'))
else:
    import numpy as np
    subprocess.run([sys.executable, __file__, '/tmp/a.pkl'], env=dict(os.environ, TA_RF_FAST_MAX='0'))
    subprocess.run([sys.executable, __file__, '/tmp/b.pkl'], env=dict(os.environ, TA_RF_FAST_MAX=os.environ.get('FM', '1024')))
    ca, ba, sa = pickle.load(open('/tmp/a.pkl', 'rb')); cb, bb, sb = pickle.load(open('/tmp/b.pkl', 'rb'))
    print('counts slow', ca, 'fast', cb)
    oa = ob = 0
    for i in range(len(ca)):
        A = sa[oa:oa + ca[i]]; B = sb[ob:ob + cb[i]]
        k = 0
        while k < min(len(A), len(B)) and A[k] == B[k]:
            k += 1
        print('img', i, 'first difference at kept index', k, 'of', len(A), len(B), A[k:k+3], B[k:k+3])
        oa += ca[i]; ob += cb[i]

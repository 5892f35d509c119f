"""Single-frame latency of the three facades (1080p, one frame per call, serial): the webcam / live-video use of the
reference (examples), as opposed to bench.py's batched throughput.  Run on the GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import Detection, Recognition, Estimation, synth, weights   # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
    sd = (weights.make_retinaface_state(), weights.make_arcface_state(), weights.make_openpose_decoder_state())
    det = Detection(device=0, state=sd[0], precision=prec)
    rec = Recognition(device=0, state=sd[1], precision=prec)
    est = Estimation(device=0, state=sd[2], precision=prec)
    frame = synth.upscale_for_resize(synth.pose_code_frames(4, 1, 184, 327, 4), 1080, 1920)[0]

    def step():
        t = [time.perf_counter()]
        faces = det(frame)
        t.append(time.perf_counter())
        feats = rec(frame, faces[:2])
        t.append(time.perf_counter())
        poses = est(frame)
        t.append(time.perf_counter())
        return np.diff(t) * 1e3, len(faces), len(feats), len(poses)
    for _ in range(5):
        step()
    rows = np.array([step()[0] for _ in range(30)])
    _, nf, ne, npose = step()
    med = np.median(rows, axis=0)
    print('%s: one 1080p frame (host ndarray in, results out), median of 30: detect %.2f ms (%d faces), embed(2) %.2f ms, '
          'pose %.2f ms (%d people); total %.2f ms = %.0f frames/s unbatched' %
          (prec, med[0], nf, med[1], med[2], npose, med.sum(), 1e3 / med.sum()))


if __name__ == '__main__':
    main()

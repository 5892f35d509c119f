"""Per-model throughput at SURVEY.md 8's shapes C2 / C3 / C4 (run on the GPU box).

    python tools/model_bench.py [--out gpurun_out/per_model.json] [--precisions f16x3 f32 f16]
('f16' = the single-half embedder; the detector and the pose network run as f16x3 in that mode)

C2  RetinaFace   32 x 640 x 640   frames resident in HBM -> network + decode + sort + NMS + result download
C3  ArcFace      256 x 3 x 112 x 112 BGR crops (host)    -> upload + network + L2 norm + download (0.7 MB/crop: PCIe-light)
C4  OpenPose     16 x 368 x 656   frames resident in HBM -> network + x8 bicubic + peaks + limbs + assembly + download
Each line: images/s over `reps` calls after warm-up, and the conv kernels' share from one HIP-event profiled call
(algorithmic conv TFLOP/s, time in conv / other kernels).  Seeds follow SURVEY.md 8(d)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import arcface, openpose, retinaface, runtime, synth, weights   # noqa: E402


def timed(ctx, fn, reps, warm=2):
    for _ in range(warm):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    ctx.profile_reset()
    ctx.profile(True)
    fn()
    ctx.sync()
    conv_ms, conv_n, conv_flops = ctx.profile_read(0)
    other_ms = other_n = 0
    for klass in (1, 2, 3):                 # layer kernels, ingest (resize / preprocess / warp), post-processing
        ms, n, _ = ctx.profile_read(klass)
        other_ms += ms
        other_n += n
    ctx.profile(False)
    return dt, {'conv_ms': round(conv_ms, 3), 'conv_launches': conv_n,
                'conv_tflops': round(conv_flops / max(conv_ms, 1e-9) / 1e9, 1),
                'other_kernels_ms': round(other_ms, 3), 'other_launches': other_n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=None)
    ap.add_argument('--precisions', nargs='+', default=['f16x3', 'f32'])
    ap.add_argument('--reps', type=int, default=5)
    args = ap.parse_args()
    ctx = runtime.get_context(0)
    sd_r, sd_a, sd_p = weights.make_retinaface_state(), weights.make_arcface_state(), weights.make_openpose_state()
    c2 = ctx.upload(synth.frames(1, 32, 640, 640))
    c3 = np.random.default_rng(2).integers(0, 256, (256, 3, 112, 112), dtype=np.uint8)
    c4 = ctx.upload(synth.frames(3, 16, 368, 656))
    rows = []
    for prec in args.precisions:
        det = retinaface.RetinaFace(device=0, state=sd_r, precision=prec)
        dt, prof = timed(ctx, lambda: det.call_frames(c2), args.reps)
        rows.append({'config': 'C2 RetinaFace 32x640x640', 'precision': prec, 'images_per_s': round(32 / dt, 1),
                     'ms_per_batch': round(dt * 1e3, 3), **prof})
        arc = arcface.ArcFace(device=0, state=sd_a, precision=prec)
        dt, prof = timed(ctx, lambda: arc.embed_crops(c3), args.reps)
        rows.append({'config': 'C3 ArcFace 256x3x112x112', 'precision': prec, 'images_per_s': round(256 / dt, 1),
                     'ms_per_batch': round(dt * 1e3, 3), **prof})
        pose = openpose.OpenPose(device=0, short_side=368, state=sd_p, precision=prec)
        dt, prof = timed(ctx, lambda: pose.call_frames(c4), args.reps)
        rows.append({'config': 'C4 OpenPose 16x368x656', 'precision': prec, 'images_per_s': round(16 / dt, 1),
                     'ms_per_batch': round(dt * 1e3, 3), **prof})
        del det, arc, pose
    # C4 grouping alone on synthetic maps with P people per frame (seeds 10..12): upload of the
    # network-resolution maps + x8 bicubic + peaks + limb scoring + matching + assembly + download
    for seed, P in ((10, 1), (11, 4), (12, 16)):
        hm, paf = synth.pose_maps_batch(seed, 16, P, 46, 82)
        dt, prof = timed(ctx, lambda: openpose.group(ctx, paf, hm), args.reps)
        humans = sum(len(x) for x in openpose.group(ctx, paf, hm)) / 16.0
        rows.append({'config': 'C4 grouping only, 16 x (57,46,82) maps, P=%d people/frame' % P, 'precision': '-',
                     'images_per_s': round(16 / dt, 1), 'ms_per_batch': round(dt * 1e3, 3),
                     'humans_per_frame': humans, 'other_kernels_ms': prof['other_kernels_ms'],
                     'other_launches': prof['other_launches']})
    for r in rows:
        print(json.dumps(r))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, 'w') as f:
            json.dump(rows, f, indent=1)


if __name__ == '__main__':
    main()

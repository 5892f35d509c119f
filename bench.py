"""bench.py -- end-to-end 1080p detect + embed + pose throughput on MI355X.

    python bench.py --gpus 1 --steps 30 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[4], the configuration the metric is quoted on; it fits one GPU):
a batch of B=32 synthetic 1080p RGB frames per GPU per step, ALREADY RESIDENT in HBM, through
  Detection(short_side=416)  -> RetinaFace 416x739 + decode/NMS          (face/detection/__init__.py)
  Recognition(top-F faces)   -> similarity warp + ArcFace-R100 + L2 norm   (face/recognition/__init__.py)
  Estimation(short_side=184) -> OpenPose 184x327 + x8 bicubic + grouping   (pose/__init__.py)
with random-init weights of the exact architectures (no checkpoints offline) and the full host
side of the wrappers (result download, dict construction, landmark alignment math).
One "step" = one such batch (F = 2 faces per frame by default; F = 1 and F = 4, the counts SURVEY.md 8d quotes, are
measured as well and reported under `other_faces_per_frame`).  Per GPU the host keeps two batches in flight, each on
three threads / HIP streams (detect -> queue -> embed, pose); all K steps complete inside the timed region.  Frames
shard embarrassingly: every rank owns its own batches, there is no data-path collective ("scaling": "weak").

Prints ONE JSON line on rank 0 (see README/DESIGN.md for `roofline` and `cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

H, W = 1080, 1920
# Dense MFMA peaks from /opt/skills/guides/MI355X_MICROARCH.md (spec): f32-input 157.3 TFLOP/s, bf16 2.5 PFLOP/s.
# precision -> (peak of the MFMA opcode used, note, MFMA flops issued per algorithmic flop)
PEAKS = {
    'f32': (157.3, 'v_mfma_f32_32x32x2_f32 (exact f32)', 1),
    'bf16x3': (2500.0, 'v_mfma_f32_32x32x16_bf16 x3 (hi*hi + hi*lo + lo*hi, f32 accumulate)', 3),
    'bf16': (2500.0, 'v_mfma_f32_32x32x16_bf16', 1),
}
DTYPES = {'f32': 'f32',
          'bf16x3': 'bf16x3 (operands x = hi + lo as two bf16, ~16 mantissa bits; 3 bf16 MFMAs per product term, f32 '
                    'accumulate; RetinaFace activations stay f32) -- passes the same 1e-3 / bit-exact parity suite as f32',
          'bf16': 'bf16 (f32 accumulate)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='frames per GPU per step (examples/video.py:12)')
    ap.add_argument('--faces', type=int, default=2, help='faces embedded per frame (top-F detections)')
    ap.add_argument('--cpu-frames', type=int, default=16, help='frames in the bounded CPU-baseline sample')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', default=None, choices=['f32', 'bf16x3', 'bf16'],
                    help='conv arithmetic mode of the headline number (default: $TERRAN_AMD_PRECISION or bf16x3)')
    ap.add_argument('--single-mode', action='store_true', help='skip the secondary f32-MFMA measurement')
    ap.add_argument('--inflight', type=int, default=2, help='batches in flight per GPU (pipelines of 3 streams each)')
    ap.add_argument('--serial', action='store_true',
                    help='one kernel at a time (detect, embed, pose back to back on one host thread): the mode the '
                         'rocprofv3 kernel statistics under profiles/ are taken in, so that their per-kernel averages '
                         'are comparable with the HIP-event roofline figures')
    ap.add_argument('--join-steps', action='store_true', help='join the face and pose threads after every step')
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): everything libraries print to fd 1 (RCCL's version banner, for
    # one) is sent to stderr instead, and the result is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    dist = None
    # RCCL (backend "nccl") on the GPU box; TA_BENCH_BACKEND=gloo lets the multi-rank path be exercised on a
    # single-GPU box (ranks then share device LOCAL_RANK % device_count).
    backend = os.environ.get('TA_BENCH_BACKEND', 'nccl')
    n_dev = max(torch.cuda.device_count(), 1)
    device_index = local_rank % n_dev
    use_dist = world > 1 or bool(os.environ.get('TA_BENCH_FORCE_DIST'))     # FORCE: exercise the RCCL calls at N=1
    if use_dist:
        import torch.distributed as dist
        if backend == 'nccl':
            torch.cuda.set_device(device_index)
            dist.init_process_group(backend='nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', device_index))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    from terran_amd import Detection, Recognition, Estimation, runtime, synth, weights

    # Host side: a pipeline of three host threads per GPU, each with its own context (HIP stream + scratch):
    # detection, embedding (fed the detections of its batch through a queue) and pose.  Kernels of the streams
    # interleave on the device -- one stream's conv fills the CUs another's tail and launch gap leave idle -- and the
    # host-side result handling of one thread hides under the others' device time.  Measured on one MI355X
    # (frames/s): 2 threads joined per step 1262, free-running 1300, 3 threads 1490, and 1620 once released frame
    # buffers are parked per context instead of hipFree'd (hipFree waits for every stream of the process).
    # `--inflight L` runs L such pipelines on alternate batches (default 2 batches in flight: with one workgroup per
    # CU per conv kernel the extra streams fill more gaps: 1890 -> 1945 frames/s; L = 3 adds nothing).
    import queue
    from concurrent.futures import ThreadPoolExecutor
    # a thread coming back from a (GIL-free) library call must not wait a whole 5 ms interpreter time slice behind
    # another thread's result handling before it can queue its next launch
    sys.setswitchinterval(float(os.environ.get('TA_BENCH_SWITCH', '2e-4')))
    sd_r, sd_a, sd_p = weights.make_retinaface_state(), weights.make_arcface_state(), weights.make_openpose_state()
    frames_host = synth.frames(4 + rank, args.batch, H, W)        # C5 seed 4
    F = args.faces
    face_state = {'F': F}                                               # pick_faces reads the current faces-per-frame
    fallback_lm = synth.landmarks(77, 4 if F < 4 else F, H, W)
    L = max(1, args.inflight)
    pool = ThreadPoolExecutor(max_workers=3 * L)

    def pick_faces(dets):
        faces, nf = [], face_state['F']
        for d in dets:
            f = [{'landmarks': x['landmarks']} for x in d[:nf]]
            for k in range(len(f), nf):                             # fewer than F detections: synthetic landmarks
                f.append({'landmarks': fallback_lm[k]})
            faces.append(f)
        return faces

    class Pipeline:
        """Three streams on one GPU and this pipeline's handles on the resident batch."""

        def __init__(self, first):
            self.ctxs = [runtime.get_context(device_index) if first else runtime.new_context(device_index),
                         runtime.new_context(device_index), runtime.new_context(device_index)]
            self.frames = [c.upload(frames_host) for c in self.ctxs]      # resident in HBM before timing
            self.det = self.rec = self.est = None

        def load(self, precision):
            c_det, c_rec, c_pose = self.ctxs
            self.det = Detection(short_side=416, device=device_index, state=sd_r, ctx=c_det, precision=precision)
            self.rec = Recognition(device=device_index, state=sd_a, ctx=c_rec, precision=precision)
            self.est = Estimation(short_side=184, device=device_index, state=sd_p, ctx=c_pose, precision=precision)

        def unload(self):
            for m in (self.det.model, self.rec.model, self.est.model):
                m.model.free()

        def serial_step(self):
            dets = self.det(self.frames[0])
            return dets, self.rec.model.call(self.frames[1], pick_faces(dets)), self.est(self.frames[2])

        def start(self, k):
            """Queue k steps on this pipeline's three threads; returns a callable that joins them."""
            if k == 0:
                return lambda: None
            q = queue.Queue()

            def detect_loop():
                res = []
                for _ in range(k):
                    res.append(self.det(self.frames[0]))
                    q.put(res[-1])
                return res

            def embed_loop():
                return [self.rec.model.call(self.frames[1], pick_faces(q.get())) for _ in range(k)]

            def pose_loop():
                return [self.est(self.frames[2]) for _ in range(k)]
            futs = [pool.submit(f) for f in (detect_loop, embed_loop, pose_loop)]

            def join():
                d, e, p_ = [f.result() for f in futs]
                assert len(d) == k and len(e) == k and len(p_) == k       # every step produced all three results
                return d[-1], e[-1], p_[-1]
            return join

        def sync(self):
            for c in self.ctxs:
                c.sync()

        def free(self):
            for f in self.frames:
                f.free()

    pipes = [Pipeline(i == 0) for i in range(L)]

    def sync():
        for p in pipes:
            p.sync()
        if use_dist:
            if backend == 'nccl':
                torch.cuda.synchronize()
            dist.barrier()

    def run_steps(k):
        """k steps, pipeline p taking steps p, p+L, ...; with --join-steps one step at a time on pipeline 0."""
        if args.serial:
            for _ in range(k):
                res = pipes[0].serial_step()
            return res
        if args.join_steps:
            for _ in range(k):
                res = pipes[0].start(1)()
            return res
        joins = [p.start(len(range(i, k, L))) for i, p in enumerate(pipes)]
        outs = [j() for j in joins]
        return next(o for o in outs if o is not None)

    def run_mode(precision):
        """Warm up, time `steps` steps (barrier + sync on both sides, max over ranks), then one serial
        step with a HIP event pair around every launch for the per-kernel roofline."""
        for p in pipes:
            p.load(precision)
        if args.warmup:
            run_steps(max(args.warmup, L))                          # every pipeline warms its plans
        sync()
        t0 = time.perf_counter()
        out = run_steps(args.steps)
        sync()
        elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        p0 = pipes[0]
        for c in p0.ctxs:
            c.profile_reset()
            c.profile(True)
        p0.serial_step()
        klass = {}
        for k, name in enumerate(('conv_igemm', 'dw_pool_copy', 'preprocess', 'postprocess')):
            ms = n = work = 0
            for c in p0.ctxs:
                a, b, w_ = c.profile_read(k)
                ms, n, work = ms + a, n + b, work + w_
            klass[name] = {'ms': round(ms, 3), 'launches': n, 'work': work}
        for c in p0.ctxs:
            c.profile(False)
        for p in pipes:
            p.unload()
        return elapsed, out, klass

    def roofline(precision, klass):
        conv = klass['conv_igemm']
        achieved = conv['work'] / (conv['ms'] * 1e-3) / 1e12 if conv['ms'] > 0 else 0.0
        peak, note, factor = PEAKS[precision]
        r = {
            'kernel': 'conv_igemm / conv_igemm_pipe, %s' % note,
            'bound': 'mfma',
            'achieved': round(achieved, 2),
            'peak': peak,
            'unit': 'TFLOP/s',
            'frac': round(achieved / peak, 4),
            'traffic': None,
            'launches_per_step': conv['launches'],
            'avg_launch_ms': round(conv['ms'] / max(conv['launches'], 1), 4),
            'algorithmic_gflop_per_step': round(conv['work'] / 1e9, 1),
            'mfma_flops_per_algorithmic_flop': factor,
            'mfma_issue_frac': round(achieved * factor / peak, 4),
        }
        pmc = os.path.join(REPO, 'profiles', 'pmc_conv_%s.json' % precision)
        if os.path.exists(pmc):                       # rocprofv3 --pmc passes of this same command (profiles/README.md)
            r['traffic'] = round(json.load(open(pmc))['hbm_bytes_per_launch'])
            r['traffic_source'] = 'profiles/pmc_conv_%s.json (FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)' % precision
        pm = os.path.join(REPO, 'profiles', 'pmc_mfma.json')
        if os.path.exists(pm):                        # hardware counters on the dominant layers (profiles/README.md)
            c = json.load(open(pm)).get(precision)
            if c:
                r['pmc_dominant_layers'] = dict(c, source='profiles/pmc_mfma.json (tools/clock_probe.sh)')
        return r

    primary = runtime.resolve_precision(args.precision or os.environ.get('TERRAN_AMD_PRECISION') or 'bf16x3')
    elapsed, out, klass = run_mode(primary)
    others = {}
    if not args.single_mode:
        for prec in ('f32',):
            if prec != primary:
                e2, _, k2 = run_mode(prec)
                others[prec] = {'value': round(args.batch * args.steps * world / e2, 3),
                                'ms_per_step': round(e2 / args.steps * 1e3, 3), 'roofline': roofline(prec, k2)}

    # SURVEY.md 8(d) quotes the workload at F = 1 and F = 4 faces per frame: same pipeline, headline precision
    other_faces = {}
    if not args.single_mode:
        for nf in (1, 4):
            if nf != F:
                face_state['F'] = nf
                e3, _, k3 = run_mode(primary)
                other_faces[str(nf)] = {'value': round(args.batch * args.steps * world / e3, 3),
                                        'ms_per_step': round(e3 / args.steps * 1e3, 3),
                                        'algorithmic_gflop_per_step': round(k3['conv_igemm']['work'] / 1e9, 1)}
        face_state['F'] = F

    result = None
    if rank == 0:
        dets, feats, poses = out
        total_frames = args.batch * args.steps * world
        result = {
            'metric': 'frames/sec 1080p detect+embed+pose',
            'value': round(total_frames / elapsed, 3),
            'unit': 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': DTYPES[primary],
            'data': 'synthetic',
            'config': {
                'workload': 'BASELINE configs[4]: 1080p frames, %d per GPU per step, resident in HBM; '
                            'Detection(short_side=416) + Recognition(top-%d faces/frame) + '
                            'Estimation(short_side=184); random-init weights (seeds 100/101/102)'
                            % (args.batch, F),
                'precision': primary,
                'frames_per_gpu_step': args.batch,
                'faces_per_frame': F,
                'detections_per_frame': round(float(np.mean([len(d) for d in dets])), 1),
                'humans_per_frame': round(float(np.mean([len(p) for p in poses])), 2),
                # the random-weight pose net rarely assembles a person but keeps the grouping stage busy:
                'pose_peaks_per_frame': round(pipes[0].ctxs[2].pose_stats()[0] / float(args.batch), 1),
                'pose_limb_connections_per_frame': round(pipes[0].ctxs[2].pose_stats()[1] / float(args.batch), 1),
                'sharding': 'frames split over ranks, no data-path collective',
                'streams_per_gpu': 3 * L,
                'batches_in_flight_per_gpu': L,
                'step_overlap': 'serial: one kernel at a time' if args.serial else
                                'one step at a time' if args.join_steps else
                                '%d pipeline(s) of detect / embed / pose host threads, pipeline p takes steps p, p+%d, '
                                '... (embed consumes the detections of its batch through a queue); joined once at '
                                'the end of the timed region' % (L, L),
            },
            'roofline': roofline(primary, klass),
            'stage_ms_per_step': {k: v['ms'] for k, v in klass.items()},
            # algorithmic bytes / kernel time of the HBM-bound kernel classes (peak 8000 GB/s); post-processing
            # mixes the 430 MB x8-upsample stream with latency-bound selection / grouping kernels
            'stage_hbm_gbps': {k: round(v['work'] / (v['ms'] * 1e-3) / 1e9, 1) for k, v in klass.items()
                               if k != 'conv_igemm' and v['ms'] > 0},
            'other_precisions': others,
            'other_faces_per_frame': other_faces,
        }
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(frames_host[:args.cpu_frames], F, sd_r, sd_a, sd_p, fallback_lm)
    for p in pipes:
        p.free()
    pool.shutdown()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(result) + '\n').encode())
    os.close(real_stdout)


def cpu_baseline(frames_host, F, sd_r, sd_a, sd_p, fallback_lm):
    """The oracle (this repo's CPU restatement of Terran's path: torch-CPU fp32 nets + numpy
    post-processing, pinned to the reference by tests/golden) on a bounded sample of the same
    workload, all host cores.  kind = "port": the reference's Python cannot travel to the GPU box."""
    import torch
    from oracle import pipeline
    n = len(frames_host)
    t0 = time.perf_counter()
    dets = pipeline.detection(sd_r, frames_host, short_side=416)
    t1 = time.perf_counter()
    faces = []
    for d in dets:
        f = [{'landmarks': x['landmarks']} for x in d[:F]]
        for k in range(len(f), F):
            f.append({'landmarks': fallback_lm[k]})
        faces.append(f)
    pipeline.recognition(sd_a, list(frames_host), faces)
    t2 = time.perf_counter()
    pipeline.estimation(sd_p, frames_host, short_side=184, bicubic_impl='torch')
    t3 = time.perf_counter()
    dt = t3 - t0
    return {'value': round(n / dt, 4), 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d of the same 1080p frames through oracle.pipeline detection+recognition(top-%d)+estimation '
                      '(torch-CPU fp32, %.1f s)' % (n, F, dt),
            'stage_ms_per_frame': {'detection': round((t1 - t0) / n * 1e3, 1), 'recognition': round((t2 - t1) / n * 1e3, 1),
                                   'estimation': round((t3 - t2) / n * 1e3, 1)}}


if __name__ == '__main__':
    main()

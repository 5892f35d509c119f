"""bench.py -- end-to-end 1080p detect + embed + pose throughput on MI355X.

    python bench.py --gpus 1 --steps 30 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[4], the configuration the metric is quoted on; it fits one GPU):
a batch of B=32 synthetic 1080p RGB frames per GPU per step, ALREADY RESIDENT in HBM, through
  Detection(short_side=416)  -> RetinaFace 416x739 + decode/NMS          (face/detection/__init__.py)
  Recognition(top-F faces)   -> similarity warp + ArcFace-R100 + L2 norm   (face/recognition/__init__.py)
  Estimation(short_side=184) -> OpenPose 184x327 + x8 bicubic + grouping   (pose/__init__.py)
with random-init weights of the exact architectures (no checkpoints offline) and the full host side of the wrappers
(result download, dict construction, landmark alignment math).  The frames carry pose maps and the pose weights hold the
matching decoder path (terran_amd/weights.py:make_openpose_decoder_state), so people DO assemble (4 per frame).
One "step" = one such batch (F = 2 faces per frame by default; F = 1 and F = 4, the counts SURVEY.md 8d quotes, are
measured as well and reported under `other_faces_per_frame`).  Per GPU the host keeps two batches in flight, each on
three threads / HIP streams (detect -> queue -> embed, pose); all K steps complete inside the timed region.  Frames
shard embarrassingly: every rank owns its own batches, there is no data-path collective ("scaling": "weak").

stdout carries ONE JSON line of <= 8 KB (`compact_line`): the headline (`value`: f16x3 mode, the library default --
split-half operands, 22 bits, three f16 MFMAs per product, float32-grade results: tests/test_gpu_decisions_vs_oracle.py --
frames resident in HBM, ONE resident batch re-used by every step: `config.resident_batch_reused`), the dominant kernel
instance's `roofline`, `cpu_baseline`, and beside them
  value_f32 / roofline_f32      the same workload with every conv on the exact-f32 MFMA (the like-for-like arithmetic)
  value_f16x2                   the opt-in tolerance mode: the embedder alone on two of the three products (<= 5e-4 guarded at load)
  value_ingest                  the same workload fed from HOST memory: a raw rgb24 byte stream read into pinned buffers and
                                uploaded by video.RawVideoReader threads (upload overlapped with compute), every step a fresh
                                batch, results gathered in order on rank 0 WHILE the region runs (a gather thread per rank) --
                                the figure comparable with the reference's `.call`, whose first act is the host -> device
                                copy of the batch (retinaface/wrapper.py:144)
  value_k_steps                 the exact K = --steps region (`value` itself is quoted from a region of >= 2.5 s: a 0.25 s
                                region reads ~8 % above what the loop sustains -- the chip clocks to its power budget)
  per_model                     BASELINE configs C2 (RetinaFace 32x640x640), C3 (ArcFace 256 crops), C4 (OpenPose
                                16x368x656): images/s and roofline fraction each (N = 1 only)
Everything else this run measured (every leg's full roofline object, power, per-rank host figures, per-model rows, the prose)
goes to `gpurun_out/bench_detail.json` (`--detail PATH`) and to stderr; `--full` adds the legs of earlier rounds (bf16x3 / f16 /
bf16 precisions, 1 and 4 faces per frame, pose weights without structured zeros); `--full-line` prints the detail object on
stdout instead of the compact line (tools/*.sh).
`python bench.py --gpus N --single-process` (no torchrun) drives N devices from ONE process through
terran_amd.pipeline.StreamPipeline (per device: lanes of upload / detect -> embed / pose threads) instead of one process
per GPU; `--devices 0,0` puts two replicas on one card.
"""
import argparse
import io
import json
import os
import queue
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

H, W = 1080, 1920
# Dense MFMA peaks from /opt/skills/guides/MI355X_MICROARCH.md (spec): f32-input 157.3 TFLOP/s, bf16 2.5 PFLOP/s; HBM 8 TB/s.
# precision -> (peak of the MFMA opcode used, note, MFMA flops issued per algorithmic flop)
PEAKS = {
    'f32': (157.3, 'v_mfma_f32_32x32x2_f32 (exact f32)', 1),
    'f16x3': (2500.0, 'v_mfma_f32_32x32x16_f16 x3 (hi*hi + hi*lo + lo*hi on split-half operands, f32 accumulate)', 3),
    'bf16x3': (2500.0, 'v_mfma_f32_32x32x16_bf16 x3 (hi*hi + hi*lo + lo*hi, f32 accumulate)', 3),
    'bf16': (2500.0, 'v_mfma_f32_32x32x16_bf16', 1),
    'f16': (2500.0, 'v_mfma_f32_32x32x16_f16: x3 (split-half operands) in the detector and the pose network, x1 in the embedder', 3),
    'f16x2': (2500.0, 'v_mfma_f32_32x32x16_f16: x3 (split-half operands) in the detector and the pose network, x2 in the embedder ((w_hi + w_lo) * x_hi)', 3),
}
HBM_PEAK_GBPS = 8000.0
# The mode `value` is measured in = the LIBRARY DEFAULT (runtime.DEFAULT_PRECISION): every network on the float32-grade split-half
# arithmetic (x = hi + lo, 22 significant bits, three f16 MFMAs per product; 0 decision flips vs the oracle, embeddings to 5e-7).
# Beside it in the same line: `value_f16x2` (the opt-in tolerance mode: the embedder alone on two of the three products, guarded
# at load time, arcface.calibrate_f16x2) and `value_f32` (every conv on the exact-f32 MFMA).
HEADLINE = 'f16x3'
SIDE_PRECISIONS = ('f32', 'f16x2')                                    # default side legs; --full: + bf16x3, f16, bf16
# `dtype` of the compact line: <= 200 characters
DTYPE_SHORT = {'f32': 'f32 (v_mfma_f32_32x32x2_f32)',
               'f16x3': 'f16x3: operands x = hi + lo (two IEEE halfs, 22 bits), 3 f16 MFMAs per product, f32 accumulate; float32-grade results',
               'bf16x3': 'bf16x3: operands x = hi + lo (two bf16, 16 bits), 3 bf16 MFMAs per product, f32 accumulate',
               'bf16': 'bf16 (f32 accumulate); throughput mode outside the parity bar',
               'f16x2': 'f16x3 for detector + pose; embedder on 2 of the 3 products ((w_hi + w_lo) * x_hi), guarded <= 5e-4 at load',
               'f16': 'f16x3 for detector + pose; embedder on one f16 MFMA per product (tolerance mode)'}
DTYPES = {'f32': 'f32',
          'f16x3': 'f16x3 (ArcFace / OpenPose operands x = hi + lo as two IEEE half floats = 22 significant bits, weight rows '
                   'scaled by a power of two per OUTPUT CHANNEL, stored activations by one per channel; 3 f16 MFMAs per product term (every product exact), f32 '
                   'accumulate; the detector RetinaFace: refiner + deep base the same, the raw-pixel front exact f32) -- float32-grade: 0 decision flips vs '
                   'the oracle over 224 frames per task where the exact-f32 mode has 2 (profiles/r03_decisions_vs_oracle.txt)',
          'bf16x3': 'bf16x3 (ArcFace / OpenPose operands x = hi + lo as two bf16, ~16 mantissa bits; 3 bf16 MFMAs per '
                    'product term, f32 accumulate; the detector RetinaFace runs on the exact-f32 MFMA) -- passes the '
                    'same 1e-3 / bit-exact parity suite as f32',
          'bf16': 'bf16 (f32 accumulate)',
          'f16x2': 'f16x3 for the detector and the pose network (every discrete decision at float32 grade, as in the f16x3 mode); the embedder '
                   'ArcFace -- no decisions, north_star bar 1e-3 on the unit-norm embedding -- on TWO of the three products, (w_hi + w_lo) * x_hi: '
                   'weights and the shortcut trunk keep 22 bits, every activation enters a contraction as its hi half (11 bits): 1.8e-4 worst '
                   'embedding component vs the oracle on the seeded weights, 8.2e-4 on the wild-statistics weights (tests/probe_embedder_modes.py)',
          'f16': 'f16x3 for the detector and the pose network (every discrete decision at float32 grade, as in the f16x3 mode); the '
                 'embedder ArcFace -- no decisions, north_star bar 1e-3 on the unit-norm embedding -- with ONE f16 MFMA per product '
                 'on 2-byte half-float activations: 3.6e-4 worst embedding component vs the oracle (tolerance mode for that one task)'}
KLASSES = ('conv_igemm', 'dw_pool_copy', 'preprocess', 'postprocess')


def _inst_factor(precision, name):
    """MFMAs per product of a kernel instance: PREC_F16X2 (template argument 5) instances are the embedder's two-product kernels."""
    if precision == 'f32':
        return 1
    a = name[name.index('<') + 1:name.rindex('>')].split(',') if '<' in name else []
    if name.startswith('conv_igemm_split') and len(a) > 3:
        prec = a[3]
    elif name.startswith('conv_igemm_win') and len(a) > 2:
        prec = a[2]
    elif name.startswith('conv_igemm_pipe') and len(a) > 4:
        prec = a[4]
    elif name.startswith(('conv_igemm<', 'conv_dwpw')) and len(a) > 4:
        prec = a[4]
    else:
        prec = ''
    return {'0': 1.0 / 16, '1': 3, '2': 1, '3': 3, '4': 1, '5': 2}.get(prec.strip(), 3)     # (an exact-f32 op inside a 16-bit program: 1/16 of an f16 MFMA's FLOPs)


def issued_mfma_frac(precision, instances):
    """MFMA FLOPs ISSUED (algorithmic FLOPs x products per term of each instance) per second of the profiled serial step's conv time,
    over the dense f16 / bf16 peak: the modes that mix two- and three-product kernels have no single factor."""
    ms = sum(v[2] for v in instances.values()) if instances else 0.0
    if precision == 'f32' or ms <= 0:
        return None
    issued = sum(v[1] * _inst_factor(precision, k) for k, v in instances.items() if _inst_factor(precision, k) >= 1)
    return round(issued / (ms * 1e-3) / 1e12 / PEAKS[precision][0], 4)


def dominant_kernel(precision, instances):
    """The dense-conv kernel INSTANCE with the most HIP-event time in the profiled serial step: algorithmic FLOPs per launch over its
    average launch duration, against the dense MFMA peak of the opcode (`roofline.dominant`; the rocprofv3 table of the same
    command, profiles/r05_<mode>_kernel_stats.csv, carries the same instance with the profiler's own durations)."""
    if not instances:
        return None
    name, (n, fl, ms) = max(instances.items(), key=lambda kv: kv[1][2])
    if ms <= 0 or n <= 0:
        return None
    peak = PEAKS[precision][0]
    tf = fl / (ms * 1e-3) / 1e12
    factor = _inst_factor(precision, name)
    total_ms = sum(v[2] for v in instances.values())
    return {'kernel': name, 'launches_per_step': n, 'avg_launch_ms': round(ms / n, 4), 'gflop_per_launch': round(fl / n / 1e9, 2),
            'achieved': round(tf, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(tf / peak, 4),
            'mfma_issue_frac': round(tf * factor / peak, 4), 'share_of_conv_time': round(ms / total_ms, 3),
            'source': 'driver-run: HIP events around every launch of this instance in one serial step of this very process'}


def roofline_object(precision, klass, step_seconds):
    allc = dict(conv_roofline(precision, klass['conv_igemm']),
                mfma_issue_frac_by_instance=issued_mfma_frac(precision, klass.get('instances')),
                pipelined_achieved=round(klass['conv_igemm']['work'] / step_seconds / 1e12, 2),
                pipelined_frac=round(klass['conv_igemm']['work'] / step_seconds / 1e12 / PEAKS[precision][0], 4))
    dom = dominant_kernel(precision, klass.get('instances'))
    if dom is None:                                    # no per-instance times (should not happen): the aggregate stands in
        return dict(allc, all_conv_kernels=None)
    r = {'kernel': dom['kernel'], 'bound': 'mfma', 'achieved': dom['achieved'], 'peak': dom['peak'], 'unit': 'TFLOP/s', 'frac': dom['frac'],
         'traffic': None, 'mfma_issue_frac': dom['mfma_issue_frac'], 'launches_per_step': dom['launches_per_step'],
         'avg_launch_ms': dom['avg_launch_ms'], 'algorithmic_gflop_per_launch': dom['gflop_per_launch'],
         'share_of_conv_time': dom['share_of_conv_time'], 'source': dom['source']}
    pmc = os.path.join(REPO, 'profiles', 'pmc_conv_%s.json' % precision)
    if os.path.exists(pmc):
        inst = json.load(open(pmc)).get('per_instance', {}).get(dom['kernel'])
        if inst:
            r['traffic'] = inst['hbm_bytes_per_launch']
            r['traffic_source'] = ('builder-run: profiles/pmc_conv_%s.json per_instance (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes: '
                                   'HBM-side bytes per launch of this instance)' % precision)
    for k_ in ('pmc_dominant_layers',):
        if k_ in allc:
            r[k_] = allc.pop(k_)
    r['all_conv_kernels'] = allc
    return r


def conv_roofline(precision, conv):
    """`conv` = {'ms','launches','work'} of the implicit-GEMM kernels from HIP events on their launch streams."""
    achieved = conv['work'] / (conv['ms'] * 1e-3) / 1e12 if conv['ms'] > 0 else 0.0
    peak, note, factor = PEAKS[precision]
    r = {
        'kernel': 'conv_igemm_split / conv_igemm_pipe / conv_igemm, %s' % note,
        'bound': 'mfma',
        'achieved': round(achieved, 2),
        'peak': peak,
        'unit': 'TFLOP/s',
        'frac': round(achieved / peak, 4),
        'traffic': None,
        'launches_per_step': conv['launches'],
        'avg_launch_ms': round(conv['ms'] / max(conv['launches'], 1), 4),
        'algorithmic_gflop_per_step': round(conv['work'] / 1e9, 1),
        'mfma_flops_per_algorithmic_flop': factor if precision not in ('f16', 'f16x2') else '3 (detector, pose) / %d (embedder)' % (1 if precision == 'f16' else 2),
        'mfma_issue_frac': round(achieved * factor / peak, 4) if precision not in ('f16', 'f16x2') else None,
        'source': 'driver-run: HIP events around every conv launch of one serial step of this very process',
    }
    # The fields below are NOT measured by this run: they replay the builder's rocprofv3 --pmc passes of the same
    # command (profiles/README.md), committed as JSON.
    pmc = os.path.join(REPO, 'profiles', 'pmc_conv_%s.json' % precision)
    if os.path.exists(pmc):
        r['traffic'] = round(json.load(open(pmc))['hbm_bytes_per_launch'])
        r['traffic_source'] = ('builder-run: profiles/pmc_conv_%s.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes '
                               'per conv launch)' % precision)
    pm = os.path.join(REPO, 'profiles', 'pmc_mfma.json')
    if os.path.exists(pm):
        c = json.load(open(pm)).get('f16x3' if precision in ('f16', 'f16x2') else precision)      # f16: the dominant layers are the pose network's, on f16x3
        if c:
            r['pmc_dominant_layers'] = dict(c, source='builder-run: profiles/pmc_mfma.json (tools/clock_probe.sh)')
    return r


class LoopStream(io.RawIOBase):
    """`n_batches` copies of one frame batch as a raw rgb24 byte stream (what `ffmpeg -f rawvideo -pix_fmt rgb24 pipe:`
    delivers, terran/io/video/reader.py:421-465): readinto() is one host memcpy out of the batch, like a pipe read."""

    def __init__(self, batch, n_batches):
        self.buf = np.ascontiguousarray(batch).reshape(-1)
        self.left = len(self.buf) * n_batches
        self.pos = 0

    def readable(self):
        return True

    def readinto(self, b):
        n = min(len(b), self.left, len(self.buf) - self.pos)
        if n <= 0:
            return 0
        # numpy copies release the GIL (a memoryview slice assignment does not, and would stall every launch thread of the
        # process for the ~25 ms a 199 MB batch takes): like the read() system call behind a real pipe
        np.copyto(np.frombuffer(b, dtype=np.uint8, count=n), self.buf[self.pos:self.pos + n])
        self.pos = (self.pos + n) % len(self.buf)
        self.left -= n
        return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=144)       # >= 2 s timed region at ~15 ms per step
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--batch', type=int, default=32, help='frames per GPU per step (examples/video.py:12)')
    ap.add_argument('--faces', type=int, default=2, help='faces embedded per frame (top-F detections)')
    ap.add_argument('--cpu-frames', type=int, default=16, help='frames in the bounded CPU-baseline sample')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', default=None, choices=['f32', 'f16x3', 'bf16x3', 'bf16', 'f16', 'f16x2'],
                    help='conv arithmetic mode of the headline number (default: $TERRAN_AMD_PRECISION or f16x3)')
    ap.add_argument('--single-mode', action='store_true',
                    help='headline measurement only (no f32 / faces-per-frame / sustained / ingest / per-model legs)')
    ap.add_argument('--full', action='store_true',
                    help='also the legs of earlier rounds: bf16x3 / f16 / bf16 precisions, 1 and 4 faces per frame, pose weights '
                         'without structured zeros, the embedder modes beside each other at C3 (detail file only)')
    ap.add_argument('--full-line', action='store_true', help='print the detail object on stdout instead of the compact line (tools)')
    ap.add_argument('--detail', default=os.path.join(REPO, 'gpurun_out', 'bench_detail.json'),
                    help='where rank 0 writes everything the run measured (the stdout line is the <= 8 KB summary of it)')
    ap.add_argument('--no-side-legs', action='store_true',
                    help='headline + ingest legs only (no other precisions / faces per frame / random-weights / per-model legs): multi-rank rehearsals')
    ap.add_argument('--inflight', type=int, default=2, help='lanes per GPU: batches in flight, each on its own upload / detect / embed / pose streams '
                    '(2 / 3 / 4 measure alike to +4 % for 2 once the host threads block in their syncs, 1 is 4.5 % slower: profiles/r06_ab_inflight.txt, r06_energy_ab.txt)')
    ap.add_argument('--serial', action='store_true',
                    help='one kernel at a time (detect, embed, pose back to back on one host thread): the mode the '
                         'rocprofv3 kernel statistics under profiles/ are taken in, so that their per-kernel averages '
                         'are comparable with the HIP-event roofline figures')
    ap.add_argument('--window', type=int, default=0,
                    help='bounded run-ahead inside a pipeline: a task starts step i once all three finished step i - W (0 = unbounded)')
    ap.add_argument('--join-steps', action='store_true', help='join the face and pose threads after every step')
    ap.add_argument('--sustain-seconds', type=float, default=2.5,
                    help='minimum length of the timed region `value` is quoted from (the K-step region is reported beside it)')
    ap.add_argument('--side-seconds', type=float, default=1.5,
                    help='minimum length of the timed regions of the side legs (other precisions, faces per frame)')
    ap.add_argument('--lane-embedders', action='store_true',
                    help='A/B: one embed thread + ArcFace model per lane (64 crops per launch) instead of one embed worker per '
                         'device that launches on the faces of several batches')
    ap.add_argument('--embed-min-crops', type=int, default=320, help='embed worker: launch once this many faces are waiting ...')
    ap.add_argument('--embed-max-crops', type=int, default=512, help='... on at most this many ...')
    ap.add_argument('--embed-max-wait', type=float, default=0.020, help='... or this many seconds after the first of them arrived')
    ap.add_argument('--single-process', action='store_true',
                    help='--gpus N devices driven by ONE process through the facades\' device-list fan-out')
    ap.add_argument('--devices', default=None, help='with --single-process: comma-separated device ids (repeats allowed)')
    args = ap.parse_args()
    args.side = not (args.single_mode or args.no_side_legs)

    os.environ.setdefault('GPU_MAX_HW_QUEUES', '12')     # before torch initialises HIP: see terran_amd/lib.py:load
    os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')  # likewise (kernel arguments in device memory)
    # stdout carries exactly ONE line (the JSON): everything libraries print to fd 1 (RCCL's version banner, for
    # one) is sent to stderr instead, and the result is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        result = run_single_process(args) if args.single_process else run(args)
    finally:
        sys.stdout.flush()
    if result is not None:
        detail = json.dumps(result)
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.detail)), exist_ok=True)
            with open(args.detail, 'w') as f:
                f.write(detail + '\n')
        except OSError as e:
            print('bench: could not write %s: %s' % (args.detail, e), file=sys.stderr)
        print(detail, file=sys.stderr)
        sys.stderr.flush()
        line = detail if (args.full_line or args.serial) else compact_line(result, os.path.relpath(args.detail, REPO))
        os.write(real_stdout, (line + '\n').encode())
    os.close(real_stdout)


LINE_LIMIT = 8192          # the driver keeps the last 8 KB of stdout: the line must fit (tests/test_host_cpu.py::test_bench_line_is_compact)
ROOFLINE_KEYS = ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms', 'launches_per_step',
                 'algorithmic_gflop_per_launch', 'mfma_issue_frac', 'share_of_conv_time')


def _roof(r, keys=ROOFLINE_KEYS):
    return {k: r[k] for k in keys if k in r} if isinstance(r, dict) else None


def compact_line(result, detail_path=None):
    """The ONE stdout line: the contract's keys, the dominant kernel's roofline, the CPU baseline and the side figures the
    reader needs next to the headline -- no prose, <= LINE_LIMIT bytes.  `result` is what run() / run_single_process() return
    (the detail object).  Optional blocks are dropped, least important first, should the line ever outgrow the limit."""
    r = result
    cfg = r.get('config', {})
    host = (cfg.get('host_per_rank') or [{}])[0]
    prec = cfg.get('precision')
    line = {k: r.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'timed_region_s', 'timed_steps',
                                  'higher_is_better', 'scaling', 'vs_baseline')}
    line['dtype'] = DTYPE_SHORT.get(prec, str(r.get('dtype')))[:200]
    line['data'] = r.get('data')
    line['config'] = {k: v for k, v in (
        ('workload', cfg.get('workload_short') or str(cfg.get('workload'))[:200]),
        ('precision', prec),
        ('frames_per_gpu_step', cfg.get('frames_per_gpu_step', cfg.get('frames_per_step'))),
        ('faces_per_frame', cfg.get('faces_per_frame')),
        ('resident_batch_reused', cfg.get('resident_batch_reused', True)),
        ('batches_in_flight_per_gpu', cfg.get('batches_in_flight_per_gpu')),
        ('detections_per_frame', cfg.get('detections_per_frame')),
        ('humans_per_frame', cfg.get('humans_per_frame')),
        ('host_cpu_s_per_step', host.get('cpu_s_per_step')),
        ('host_threads', host.get('threads')),
        ('sharding', 'frames over ranks, no collective')) if v is not None}
    line['roofline'] = _roof(r.get('roofline'))
    cb = r.get('cpu_baseline')
    if isinstance(cb, dict):
        line['cpu_baseline'] = {k: (v[:160] if isinstance(v, str) else v) for k, v in cb.items()
                                if k in ('value', 'unit', 'cores', 'kind', 'sample', 'error')}
    for k in ('value_k_steps', 'value_f32', 'value_f16x3', 'value_f16x2', 'value_ingest'):
        if r.get(k) is not None:
            line[k] = r[k]
    if isinstance(r.get('roofline_f32'), dict):
        line['roofline_f32'] = _roof(r['roofline_f32'], ('kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_ms', 'launches_per_step'))
    dd = r.get('decision_drift')
    if isinstance(dd, dict):
        line['decision_drift'] = {k: dd[k] for k in ('detections', 'detections_differ', 'people', 'people_differ', 'embedding_max_abs_diff') if k in dd}
        line['decision_drift']['vs'] = 'f32 leg'
    pw = r.get('power')
    if isinstance(pw, dict):
        line['power'] = {k: pw[k] for k in ('power_w_mean', 'cap_w', 'sclk_mhz_mean', 'energy_j_per_frame', 'throttle') if k in pw}
    ing = r.get('ingest')
    if isinstance(ing, dict):
        line['ingest'] = {k: ing[k] for k in ('steps', 'ms_per_step', 'gather_tail_s', 'gather_after_last_rank_s', 'gather_messages', 'steps_gathered_on_rank0', 'error') if k in ing}
    pm = r.get('per_model')
    if isinstance(pm, dict):
        rows = {}
        for name, row in pm.items():
            if isinstance(row, dict) and 'roofline' in row and '(' not in name:
                rf = row['roofline']
                rows[name] = {'images_per_s': row.get('images_per_s'), 'bound': rf.get('bound'), 'achieved': rf.get('achieved'),
                              'unit': rf.get('unit'), 'frac': rf.get('frac')}
                if isinstance(row.get('images_per_s_packed'), (int, float)):      # C2: the same call returning the packed result arrays of the C ABI
                    rows[name]['images_per_s_packed'] = row['images_per_s_packed']   # (`images_per_s` builds the reference's list of dicts per detection)
        line['per_model'] = rows or {k: str(v)[:120] for k, v in pm.items()}
    if r.get('c2_retinaface_640'):
        line['c2_retinaface_640_images_per_s'] = r['c2_retinaface_640'].get('images_per_s')
    if detail_path:
        line['detail'] = detail_path
    out = json.dumps(line)
    for k in ('per_model', 'ingest', 'power', 'decision_drift', 'roofline_f32'):     # never expected: the line is ~3 KB
        if len(out) < LINE_LIMIT:
            break
        line.pop(k, None)
        out = json.dumps(line)
    assert len(out) < LINE_LIMIT, len(out)
    return out


def make_workload(args, rank):
    from terran_amd import synth, weights
    sd = (weights.make_retinaface_state(), weights.make_arcface_state(), weights.make_openpose_decoder_state())
    # C5 seed 4.  1080p frames whose bilinear 184 x 327 reduction is a frame that carries 4 people
    frames_host = synth.upscale_for_resize(synth.pose_code_frames(4 + rank, args.batch, 184, 327, 4), H, W)
    fallback_lm = synth.landmarks(77, max(4, args.faces), H, W)
    return sd, frames_host, fallback_lm


def run(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d (or add --single-process)'
                     % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    dist = None
    gather_group = None
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
            try:
                gather_group = dist.new_group(backend='gloo')        # host-side ordered gather of result objects
            except Exception as e:                                   # no gloo beside RCCL here: gather over RCCL instead
                print('bench: gloo group unavailable (%s); gathering over the default group' % e, file=sys.stderr)
                gather_group = None
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    from terran_amd import Detection, Recognition, Estimation, affinity, runtime, shard, telemetry, video
    # this rank's host threads (task threads, reader threads, the pinned staging they touch first) on the cores of the socket
    # its GPU hangs off: at 2 600 frames/s a rank moves 16 GB/s through pinned memory (terran_amd/affinity.py)
    placement = affinity.bind(device_index)
    power_hw = telemetry.hwmon_of_pci(placement.get('pci'))

    # Host side: a pipeline of three host threads per GPU, each with its own context (HIP stream + scratch):
    # detection, embedding (fed the detections of its batch through a queue) and pose.  Kernels of the streams
    # interleave on the device -- one stream's conv fills the CUs another's tail and launch gap leave idle -- and the
    # host-side result handling of one thread hides under the others' device time.  Measured on one MI355X
    # (frames/s): 2 threads joined per step 1262, free-running 1300, 3 threads 1490, and 1620 once released frame
    # buffers are parked per context instead of hipFree'd (hipFree waits for every stream of the process).
    # `--inflight L` runs L such pipelines on alternate batches (default 2 batches in flight: with one workgroup per
    # CU per conv kernel the extra streams fill more gaps: 1890 -> 1945 frames/s; L = 3 adds nothing).
    from concurrent.futures import ThreadPoolExecutor
    # a thread coming back from a (GIL-free) library call must not wait a whole 5 ms interpreter time slice behind
    # another thread's result handling before it can queue its next launch
    sys.setswitchinterval(float(os.environ.get('TA_BENCH_SWITCH', '2e-4')))
    (sd_r, sd_a, sd_p), frames_host, fallback_lm = make_workload(args, rank)
    # what the pose network and the frames ARE can be swapped per leg (`value_random_pose_weights`): `work` holds the current pair
    work = {'sd_p': sd_p, 'frames': frames_host}
    F = args.faces
    face_state = {'F': F}                                               # pick_faces reads the current faces-per-frame
    L = max(1, args.inflight)
    pool = ThreadPoolExecutor(max_workers=5 * L + 2)    # per pipeline: detect, embed, pose (+ feeder, collector when streaming)

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
            self.frames = [c.upload(work['frames']) for c in self.ctxs]      # resident in HBM before timing
            self.frames_of = work['frames']
            self.det = self.rec = self.est = None

        def load(self, precision):
            c_det, c_rec, c_pose = self.ctxs
            if self.frames_of is not work['frames']:                      # a leg on other frames
                for f in self.frames:
                    f.free()
                self.frames = [c.upload(work['frames']) for c in self.ctxs]
                self.frames_of = work['frames']
            self.det = Detection(short_side=416, device=device_index, state=sd_r, ctx=c_det, precision=precision)
            self.rec = Recognition(device=device_index, state=sd_a, ctx=c_rec, precision=precision)
            self.est = Estimation(short_side=184, device=device_index, state=work['sd_p'], ctx=c_pose, precision=precision)

        def unload(self):
            for m in (self.det.model, self.rec.model, self.est.model):
                m.model.free()

        def serial_step(self):
            dets = self.det(self.frames[0])
            return dets, self.rec.model.call(self.frames[1], pick_faces(dets)), self.est(self.frames[2])

        def start(self, k, reader=None, on_step=None):
            """Queue k steps on this pipeline's three threads; returns a callable that joins them.
            reader: a video.RawVideoReader -- every step then takes its OWN batch from it (host -> HBM inside the timed
            region) instead of the resident one, and frees it once all three tasks are through with it.
            on_step(dets, feats, poses): called per finished step (ordered gather)."""
            if k == 0:
                return lambda: None
            q = queue.Queue()
            src = [queue.Queue() for _ in range(3)]
            done = queue.Queue()
            WAIT = 300.0                                     # a task thread that died must not block the others forever
            # bounded run-ahead (--window W): a task starts step i only once every task has finished step i - W, the
            # way a live stream with W batches of buffering behaves.  Without it the light detector runs hundreds of
            # milliseconds ahead, and the tail of the region is the pose network alone on the GPU with nothing filling
            # its launch gaps and partial rounds.
            W_ = args.window
            prog = [0, 0, 0]
            cv = threading.Condition()

            def gate(who, i):
                if W_ <= 0:
                    return
                with cv:
                    if not cv.wait_for(lambda: min(prog) >= i - W_, timeout=WAIT):
                        raise RuntimeError('pipeline stalled at step %d' % i)

            def tick(who):
                if W_ <= 0:
                    return
                with cv:
                    prog[who] += 1
                    cv.notify_all()

            def feeder():                                   # hands batch i to the three task threads
                for _ in range(k):
                    batch = next(reader)
                    for s in src:
                        s.put(batch)

            def detect_loop():
                res = []
                for i in range(k):
                    gate(0, i)
                    fr = src[0].get(timeout=WAIT) if reader else self.frames[0]
                    res.append(self.det(fr))
                    q.put(res[-1])
                    tick(0)
                    if reader:
                        done.put(('det', (fr, res[-1])))
                return res

            def embed_loop():
                res = []
                for i in range(k):
                    gate(1, i)
                    fr = src[1].get(timeout=WAIT) if reader else self.frames[1]
                    res.append(self.rec.model.call(fr, pick_faces(q.get(timeout=WAIT))))
                    tick(1)
                    if reader:
                        done.put(('rec', res[-1]))
                return res

            def pose_loop():
                res = []
                for i in range(k):
                    gate(2, i)
                    fr = src[2].get(timeout=WAIT) if reader else self.frames[2]
                    res.append(self.est(fr))
                    tick(2)
                    if reader:
                        done.put(('est', res[-1]))
                return res
            futs = [pool.submit(f) for f in (detect_loop, embed_loop, pose_loop)]
            if reader:
                pool.submit(feeder)

                def collect():
                    # the three tasks of one batch finish in any order relative to OTHER batches' tasks, but each task
                    # thread walks the batches in order: the i-th message of every kind belongs to batch i
                    kinds = {'det': [], 'rec': [], 'est': []}
                    emitted = 0
                    for _ in range(3 * k):
                        name, val = done.get(timeout=WAIT)
                        kinds[name].append(val)
                        while emitted < min(len(v) for v in kinds.values()):
                            fr, dets = kinds['det'][emitted]
                            fr.free()
                            if on_step:
                                on_step(dets, kinds['rec'][emitted], kinds['est'][emitted])
                            emitted += 1
                futs.append(pool.submit(collect))

            def join():
                outs = [f.result(timeout=2 * WAIT) for f in futs]
                d, e, p_ = outs[:3]
                assert len(d) == k and len(e) == k and len(p_) == k       # every step produced all three results
                return d[-1], e[-1], p_[-1]
            return join

        def sync(self):
            for c in self.ctxs:
                c.sync()

        def free(self):
            for f in self.frames:
                f.free()

    # The timed regions run on the product's own engine, terran_amd.pipeline.StreamPipeline over this rank's device: L
    # lanes of (upload, detect -> embed, pose) threads with a context each.  The class above stays for the one-kernel-at-
    # a-time modes (--serial / --join-steps) and for the event-profiled serial step the roofline figures come from.
    from terran_amd.pipeline import StreamPipeline
    streaming = not (args.serial or args.join_steps)
    pipes = [Pipeline(True)]
    engine = {}

    def sync():
        for p in pipes:
            p.sync()
        if engine.get('sp') is not None:
            for c in engine['sp'].contexts():
                c.sync()
        if use_dist:
            if backend == 'nccl':
                torch.cuda.synchronize()
            dist.barrier()

    def run_steps(k, readers=None, on_step=None):
        """k steps, pipeline p taking steps p, p+L, ...; with --join-steps one step at a time on pipeline 0."""
        if args.serial:
            for _ in range(k):
                res = pipes[0].serial_step()
            return res
        if args.join_steps:
            for _ in range(k):
                res = pipes[0].start(1)()
            return res
        sp = engine['sp']
        if readers:                                   # every step takes a fresh batch a reader thread uploaded; freed after use
            batches = ([next(readers[i % len(readers)])] for i in range(k))
        else:
            batches = (engine['resident'] for _ in range(k))
        out, got = None, 0
        for out in sp.run(batches, free_resident=bool(readers)):
            got += 1
            if on_step:
                on_step(*out)
        assert got == k                               # every step produced its detections, embeddings and poses
        return out

    def timed(k, **kw):
        sync()
        t0 = time.perf_counter()
        out = run_steps(k, **kw)
        sync()
        elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, out

    def profile_serial_step():
        """One serial step with a HIP event pair around every launch (per-kernel-class time and algorithmic work)."""
        p0 = pipes[0]
        if streaming:                                 # the timed region ran on the lanes: these three models are still cold
            for _ in range(2):                        # (plans, result-capacity retry of the first detector call)
                p0.serial_step()
        for c in p0.ctxs:
            c.profile_reset()
            c.profile(True)
        before = [c.kernel_work() for c in p0.ctxs]
        p0.serial_step()
        klass = {}
        for k, name in enumerate(KLASSES):
            ms = n = work = 0
            for c in p0.ctxs:
                a, b, w_ = c.profile_read(k)
                ms, n, work = ms + a, n + b, work + w_
            klass[name] = {'ms': round(ms, 3), 'launches': n, 'work': work}
        for c in p0.ctxs:
            c.profile(False)
        # the same step per dense-conv KERNEL INSTANCE (launches, algorithmic FLOPs, HIP-event time): `roofline.dominant`
        inst = {}
        for c, w0 in zip(p0.ctxs, before):
            for name, (n1, f1, m1) in c.kernel_work().items():
                n0, f0, m0 = w0.get(name, (0, 0.0, 0.0))
                a = inst.setdefault(name, [0, 0.0, 0.0])
                a[0] += n1 - n0
                a[1] += f1 - f0
                a[2] += m1 - m0
        klass['instances'] = {k: v for k, v in inst.items() if v[0] > 0}
        return klass

    def run_mode(precision, steps, extra=None, min_seconds=0.0):
        """Warm up, time `steps` steps (barrier + sync on both sides, max over ranks), then the profiled serial step.
        extra(result_dict): further legs measured while this mode's models are loaded."""
        for p in pipes:
            p.load(precision)
        if streaming:
            engine['sp'] = StreamPipeline([device_index], inflight=L, pick_faces=pick_faces, shared_embedder=not args.lane_embedders, embed_min_crops=args.embed_min_crops,
                                          embed_max_crops=args.embed_max_crops, embed_max_wait=args.embed_max_wait,
                                          detection_kw=dict(short_side=416, state=sd_r, precision=precision),
                                          recognition_kw=dict(state=sd_a, precision=precision),
                                          estimation_kw=dict(short_side=184, state=work['sd_p'], precision=precision))
            engine['resident'] = engine['sp'].scatter(work['frames'])            # resident in HBM before timing
        if args.warmup:
            run_steps(args.warmup if not streaming else max(args.warmup, 2 * L))      # every lane warms its plans
        sampler = telemetry.PowerSampler(power_hw, bdf=placement.get('pci')).start()   # this rank's GPU: socket power / shader clock (sysfs), energy / throttlers (SMI)
        cpu0 = time.process_time()
        elapsed, out = timed(steps)
        res = {'elapsed_k': elapsed, 'steps_k': steps, 'elapsed': elapsed, 'steps': steps, 'out': out}
        # host side of this rank over the K-step region: CPU-seconds of the whole process (all lanes' threads) per step, live threads
        res['host'] = {'cpu_s_per_step': round((time.process_time() - cpu0) / steps, 5), 'threads': threading.active_count()}
        # The chip clocks to its power budget: a region of a few tenths of a second reads ~8 % above what the same loop
        # sustains.  Every figure of the line therefore comes from a region of >= `min_seconds`; the exact-K region the
        # driver asked for is reported beside it (`value_k_steps`).  The step count follows from the max-over-ranks time of
        # the K-step region, so it is the same on every rank.
        if min_seconds > 0 and elapsed < min_seconds and not (args.serial or args.join_steps):
            k = int(np.ceil(min_seconds / (elapsed / steps)))
            sampler.stop()
            sampler = telemetry.PowerSampler(power_hw, bdf=placement.get('pci')).start()
            cpu0 = time.process_time()
            e2, out = timed(k)
            res.update(elapsed=e2, steps=k, out=out)
            res['host'] = {'cpu_s_per_step': round((time.process_time() - cpu0) / k, 5), 'threads': threading.active_count()}
        sampler.stop()
        # the sensor reports a moving average: the first 0.5 s of a region still hold what ran before it
        res['power'] = sampler.summary(skip_seconds=min(0.5, 0.25 * res['elapsed']))
        reg = (res['power'] or {}).get('region')
        if reg:                                          # the accumulators over the whole region: joules per frame, why the clock is where it is
            if reg.get('energy_j'):
                res['power']['energy_j_per_frame'] = round(reg['energy_j'] / (args.batch * res['steps']), 4)
            if reg.get('throttle_residency_pct') is not None:
                res['power']['throttle'] = reg['throttle_residency_pct']
        res['klass'] = profile_serial_step()
        if extra:
            extra(res)
        if streaming:
            for fr in engine['resident']:
                fr.free()
            n_l, n_c = engine['sp'].embed_stats()
            res['crops_per_embed_launch'] = n_c / n_l if n_l else 0.0
            engine.pop('sp').close()
        for p in pipes:
            p.unload()
        return res

    def fps(elapsed, steps):
        return round(args.batch * steps * world / elapsed, 3)

    def extra_headline(res):
        if args.single_mode or args.serial or args.join_steps:
            return
        try:
            ingest_leg(res)
        except Exception as ex:                                   # the headline must survive a failing secondary leg
            import traceback
            traceback.print_exc()
            res['ingest'] = {'error': '%s: %s' % (type(ex).__name__, ex)}

    def ingest_leg(res):
        # -- ingest: frames come from host memory through RawVideoReader (pinned double buffers, own upload stream per
        #    pipeline), the region's per-step results are gathered in rank order on rank 0 inside the timed region
        per_step = res['elapsed'] / res['steps']
        k = max(4 * L, 48, int(np.ceil(args.side_seconds / per_step)))     # a region of >= --side-seconds (a 20-step one is mostly fill and drain)
        gathered = []
        lock = threading.Lock()

        def rows_of(per_image, key, tail, dtype):
            """All `key` arrays of a step's dicts as ONE array: the dicts hold row VIEWS into the packed result array of the
            batch (terran_amd.results), so the array is their common base -- no per-detection work under the GIL."""
            total = sum(len(d) for d in per_image)
            first = next((d[0][key] for d in per_image if len(d)), None)
            base = getattr(first, 'base', None)
            if base is not None and base.shape == (total,) + tail and base.dtype == dtype:
                return base
            return np.array([x[key] for d in per_image for x in d], dtype).reshape((-1,) + tail)

        def pack_step(dets, feats, poses):
            """The per-frame results of one step as a few arrays (what travels to rank 0; the dicts can be rebuilt there)."""
            nd = sum(len(d) for d in dets)
            return (np.array([len(d) for d in dets], np.int32), rows_of(dets, 'bbox', (4,), np.int32), rows_of(dets, 'landmarks', (5, 2), np.int32),
                    np.fromiter((x['score'] for d in dets for x in d), np.float32, nd),
                    [np.asarray(f) for f in feats],
                    np.array([len(p) for p in poses], np.int32),
                    np.array([x['keypoints'] for p in poses for x in p], np.int32).reshape(-1, 18, 3),
                    np.array([x['score'] for p in poses for x in p], np.float64))

        # reader threads (pinned double buffers + upload stream each): a stream read is a single-thread memcpy of the 199 MB batch, 25 - 40 ms
        # depending on the box, against a 14 ms step: four readers whatever the number of lanes (two read 1 390 frames/s on a slow box)
        R = max(L, 4)
        readers = [video.RawVideoReader(LoopStream(frames_host, len(range(i, k + 2 * L, R))), W, H,
                                        batch_size=args.batch, device=device_index) for i in range(R)]
        n_on_rank0 = [0]
        n_messages = [0]
        # steps per gather message: ONE -- a message is pickled under the GIL (a few hundred KB of result arrays per step: ~0.1 ms);
        # eight steps per message held the GIL for ~10 ms at a time, most of a step, and the lanes' launch threads waited for it
        # (world-1 rehearsal: value_ingest 18 % under `value`)
        G = int(os.environ.get('TA_BENCH_GATHER_STEPS', '1'))
        distlike = None
        if use_dist:
            distlike = dist if gather_group is None else _GroupDist(dist, gather_group)
        # The ordered gather runs WHILE the region does: a gather thread per rank sends every G finished steps' results (G = 1) to rank 0
        # (gather_object on the host-side gloo group; every rank runs the same k steps, so every rank issues the same
        # ceil(k / G) collectives in the same order).  What is left for the end of the region is the last message.
        # Without a gloo group beside RCCL the gather stays ONE message at region end (a collective on the default group from
        # a side thread would race the barrier).
        streamed = use_dist and (gather_group is not None or backend != 'nccl')
        sg = [None]

        def on_step(dets, feats, poses):                 # called per finished step, in step order
            if streamed:
                sg[0].put(pack_step(dets, feats, poses))
            else:
                with lock:
                    gathered.append(pack_step(dets, feats, poses) if use_dist else (dets, feats, poses))

        def gather_all():
            if streamed:
                sg[0].finish()
                n_on_rank0[0], n_messages[0] = sg[0].steps, sg[0].messages
            elif use_dist:
                allr = shard.gather_results(list(gathered), distlike)
                n_messages[0] = 1
                n_on_rank0[0] = len(allr) if allr is not None else 0
            else:
                n_on_rank0[0] = len(gathered)
        try:
            run_steps(2 * L, readers=readers)                                        # warm the readers' buffers (2 batches each)
            del gathered[:]
            sync()
            if streamed:                                  # terran_amd.shard.StreamedGather: a gather thread of this rank
                sg[0] = shard.StreamedGather(distlike, steps_per_message=G, keep=False)
            t0 = time.perf_counter()
            run_steps(k, readers=readers, on_step=on_step)
            tg = time.perf_counter()
            t_steps_done = time.time()                   # host wall clock: comparable across the ranks of one node
            gather_all()
            gather_s = time.perf_counter() - tg
            t_gathered = time.time()
            sync()
            e = time.perf_counter() - t0
            if use_dist:
                t = torch.tensor([e], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                e = float(t.item())
                # what the gather costs once the LAST rank has finished its steps (gather_tail_s on rank 0 also contains its wait
                # for slower ranks, which is not the gather's doing)
                t = torch.tensor([t_steps_done], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                gather_after_last = max(0.0, t_gathered - float(t.item()))
            else:
                gather_after_last = gather_s
        finally:
            for r in readers:
                r.close()
        res['ingest'] = {
            'value': fps(e, k), 'unit': 'frames/s', 'steps': k, 'ms_per_step': round(e / k * 1e3, 3),
            'host_to_device_mb_per_step': round(frames_host.nbytes / 1e6, 1),
            'steps_gathered_on_rank0': n_on_rank0[0],
            'gather_messages': n_messages[0],
            'gather_tail_s': round(gather_s, 4),           # rank 0: from ITS last step to the end of the gather (its wait for slower ranks included)
            'gather_after_last_rank_s': round(gather_after_last, 4),   # from the moment the LAST rank finished its steps to the end of the gather on rank 0
            'what': 'same workload, every batch read from a raw rgb24 byte stream in host memory into pinned buffers and '
                    'uploaded by video.RawVideoReader (one reader thread + upload stream per pipeline, overlapped with '
                    'compute), the per-step results of all ranks gathered on rank 0 by a gather thread while the region runs (one message per step); stream reads are single-thread host memcpys '
                    '(a pipe read in the reference, terran/io/video/reader.py:88-117)'}

    primary = runtime.resolve_precision(args.precision or os.environ.get('TERRAN_AMD_PRECISION') or HEADLINE)
    head = run_mode(primary, args.steps, extra_headline, min_seconds=args.sustain_seconds)
    # -- the same region with pose weights that hold NO structured zeros (weights.make_openpose_state: every row random) and frames of
    #    noise: the chip clocks to its power budget and the power an MFMA draws depends on its operands, so the decoder weights
    #    (20 - 45 % zeros in conv3_x / conv4_x / stage 6: terran_amd/weights.py) may flatter the figure.  Both legs are in the line
    #    with their power objects; if they differ by more than 2 % the un-zeroed one IS the headline (`value`).
    legs = {'decoder_pose_weights': head}
    headline_leg = 'decoder_pose_weights'
    if (args.side and args.full and not (args.serial or args.join_steps)) or os.environ.get('TA_BENCH_RANDOM_LEG'):
        from terran_amd import synth, weights
        work.update(sd_p=weights.make_openpose_state(), frames=synth.frames(900 + rank, args.batch, H, W))
        try:
            legs['random_pose_weights'] = run_mode(primary, args.steps, min_seconds=args.sustain_seconds)
        finally:
            work.update(sd_p=sd_p, frames=frames_host)
        a_, b_ = (args.batch * r['steps'] / r['elapsed'] for r in (legs['decoder_pose_weights'], legs['random_pose_weights']))
        if abs(a_ - b_) > 0.02 * max(a_, b_):
            headline_leg = 'random_pose_weights'
            for k_ in ('ingest',):                                # measured with the decoder workload's models loaded: stays with the line
                if k_ in head:
                    legs['random_pose_weights'][k_] = head[k_]
            head = legs['random_pose_weights']
    elapsed, steps_timed, out, klass = head['elapsed'], head['steps'], head['out'], head['klass']

    def drift(res):
        """Discrete results of a leg's last step against the exact-f32 leg's on the same frames: detections (rounded boxes, as sets per
        frame), people (keypoint arrays as sets per frame), and the embeddings of the faces both legs embedded."""
        base = others.get('f32', {}).get('_out') if res is not others.get('f32') else None
        if base is None or res.get('_out') is None:
            return None
        (d0, f0, p0), (d1, f1, p1) = base, res['_out']
        key = lambda d: tuple(np.rint(d['bbox']).astype(int).tolist())
        nd = sum(len(x) for x in d0)
        dd = sum(len({key(x) for x in a} ^ {key(x) for x in b}) for a, b in zip(d0, d1))
        nh = sum(len(x) for x in p0)
        dh = sum(len({x['keypoints'].tobytes() for x in a} ^ {x['keypoints'].tobytes() for x in b}) for a, b in zip(p0, p1))
        emb = 0.0
        for a, b, fa, fb in zip(d0, d1, f0, f1):
            for i in range(min(len(fa), len(fb), len(a), len(b))):
                if key(a[i]) == key(b[i]):
                    emb = max(emb, float(np.abs(np.asarray(fa[i]) - np.asarray(fb[i])).max()))
        return {'vs': 'the exact-f32 leg, last step of the region (same frames)', 'detections': nd, 'detections_differ': dd,
                'people': nh, 'people_differ': dh, 'embedding_max_abs_diff': round(emb, 7)}
    others = {}
    primary_drift = None
    if args.side:
        for prec in (('f32', 'f16x3', 'bf16x3', 'f16', 'f16x2', 'bf16') if args.full else SIDE_PRECISIONS):
            if prec != primary:
                r2 = run_mode(prec, max(L, args.steps // 2), min_seconds=args.side_seconds)
                others[prec] = {'value': fps(r2['elapsed'], r2['steps']), 'steps': r2['steps'],
                                'timed_region_s': round(r2['elapsed'], 3),
                                'ms_per_step': round(r2['elapsed'] / r2['steps'] * 1e3, 3),
                                'roofline': roofline_object(prec, r2['klass'], r2['elapsed'] / r2['steps']), 'power': r2.get('power'), '_out': r2['out']}
        legs['decoder_pose_weights']['_out'] = legs['decoder_pose_weights']['out']
        for prec, o in list(others.items()) + [(primary, legs['decoder_pose_weights'])]:
            dr = drift(o)
            if dr is not None:
                o['decision_drift'] = dr
        primary_drift = legs['decoder_pose_weights'].get('decision_drift')
        for o in others.values():
            o.pop('_out', None)

    # SURVEY.md 8(d) quotes the workload at F = 1 and F = 4 faces per frame: same pipeline, headline precision
    other_faces = {}
    if args.side and args.full:
        for nf in (1, 4):
            if nf != F:
                face_state['F'] = nf
                r3 = run_mode(primary, max(L, args.steps // 2), min_seconds=args.side_seconds)
                other_faces[str(nf)] = {'value': fps(r3['elapsed'], r3['steps']), 'steps': r3['steps'],
                                        'timed_region_s': round(r3['elapsed'], 3),
                                        'ms_per_step': round(r3['elapsed'] / r3['steps'] * 1e3, 3),
                                        'algorithmic_gflop_per_step': round(r3['klass']['conv_igemm']['work'] / 1e9, 1)}
        face_state['F'] = F

    # BASELINE configs[1] (RetinaFace 640 x 640, batch 32) at N > 1: every rank runs its own batches, aggregate images/s over a
    # region of >= 1 s (barrier + sync on both sides, max over ranks).  north_star asks for the 640 x 640 figures beside the
    # 1080p ones at every N; at N = 1 `per_model` carries the same row with its roofline.
    c2_multi = None
    if use_dist and args.side:
        from terran_amd import retinaface, synth
        det640 = retinaface.RetinaFace(device=device_index, state=sd_r, precision=primary, ctx=pipes[0].ctxs[0])
        fr640 = pipes[0].ctxs[0].upload(synth.frames(1 + rank, 32, 640, 640))

        def c2_region(k):
            sync()
            t0 = time.perf_counter()
            for _ in range(k):
                det640.call_frames(fr640)
            sync()
            e = time.perf_counter() - t0
            t = torch.tensor([e], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        c2_region(3)
        e0 = c2_region(20)
        k2 = max(20, int(np.ceil(1.0 / (e0 / 20))))
        e1 = c2_region(k2)
        c2_multi = {'images_per_s': round(32 * k2 * world / e1, 1), 'batches_per_rank': k2, 'seconds': round(e1, 3),
                    'ms_per_batch': round(e1 / k2 * 1e3, 3), 'precision': primary,
                    'what': 'RetinaFace.call on a resident 32 x 640 x 640 batch per rank (network + decode + NMS + result lists), '
                            'aggregate over all ranks'}
        fr640.free()
        det640.model.free()
    placements = [affinity.describe(placement)]
    host_per_rank = [dict(legs['decoder_pose_weights'].get('host') or {}, cores_allowed=len(os.sched_getaffinity(0)))]
    if use_dist:
        gathered_pl = [None] * world
        dist.all_gather_object(gathered_pl, affinity.describe(placement), group=gather_group)
        placements = gathered_pl
        gathered_h = [None] * world
        dist.all_gather_object(gathered_h, host_per_rank[0], group=gather_group)
        host_per_rank = gathered_h

    result = None
    if rank == 0:
        dets, feats, poses = out
        result = {
            'metric': 'frames/sec 1080p detect+embed+pose',
            # `value`, `ms_per_step`, `timed_region_s`: the region of >= --sustain-seconds (`timed_steps` steps, barrier + sync on
            # both sides, max over ranks); the EXACT K = `steps` region the caller asked for, timed the same way right before
            # it, is `value_k_steps` / `ms_per_step_k_steps` / `timed_region_s_k_steps` (a 0.25 s region reads ~8 % high: the
            # chip has not settled on its sustained clock yet)
            'value': fps(elapsed, steps_timed),
            'unit': 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(elapsed / steps_timed * 1e3, 3),
            'timed_region_s': round(elapsed, 3),
            'timed_steps': steps_timed,
            'value_k_steps': fps(head['elapsed_k'], head['steps_k']),
            'ms_per_step_k_steps': round(head['elapsed_k'] / head['steps_k'] * 1e3, 3),
            'timed_region_s_k_steps': round(head['elapsed_k'], 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': DTYPES[primary],
            'data': 'synthetic',
            'config': {
                'workload': 'BASELINE configs[4]: 1080p frames, %d per GPU per step, resident in HBM; '
                            'Detection(short_side=416) + Recognition(top-%d faces/frame) + '
                            'Estimation(short_side=184); random-init weights (seeds 100/101/102); %s'
                            % (args.batch, F,
                               'the pose weights carry the decoder path that turns the frames\' embedded pose maps into 4 people per frame'
                               if headline_leg == 'decoder_pose_weights' else
                               'pose weights WITHOUT structured zeros (weights.make_openpose_state) on frames of noise: this leg differs by more '
                               'than 2 % from the decoder-weights leg (`value_decoder_pose_weights`), so it is the headline'),
                'workload_short': 'BASELINE configs[4]: 1080p detect+embed+pose; Detection(short_side=416) + Recognition(top-%d) + '
                                  'Estimation(short_side=184); %d frames per GPU per step resident in HBM; random-init weights' % (F, args.batch),
                'headline_leg': headline_leg,
                # `value`: every step runs on the SAME resident batch (no upload in the region); `value_ingest`: a fresh upload per step
                'resident_batch_reused': True,
                'precision': primary,
                'frames_per_gpu_step': args.batch,
                'faces_per_frame': F,
                'detections_per_frame': round(float(np.mean([len(d) for d in dets])), 1),
                'humans_per_frame': round(float(np.mean([len(p) for p in poses])), 2),
                'pose_peaks_per_frame': round(pipes[0].ctxs[2].pose_stats()[0] / float(args.batch), 1),
                'pose_limb_connections_per_frame': round(pipes[0].ctxs[2].pose_stats()[1] / float(args.batch), 1),
                'sharding': 'frames split over ranks, no data-path collective',
                'host_placement_per_rank': placements,
                # per rank: CPU-seconds the whole process (every lane's threads) burned per step of the timed region, its live threads and
                # the cores it may run on -- the host side has to stay under the GPU's step time on its share of the node's cores
                'host_per_rank': host_per_rank,
                'streams_per_gpu': 4 * L if args.lane_embedders else 3 * L + 1,
                'gpu_max_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES'),
                'batches_in_flight_per_gpu': L,
                'step_overlap': 'serial: one kernel at a time' if args.serial else
                                'one step at a time' if args.join_steps else
                                'terran_amd.pipeline.StreamPipeline: %d lanes of upload / detect / pose host threads (a context = HIP '
                                'stream each), lane l takes steps l, l+%d, ...; %s; results collected in step order; the region '
                                'ends when the last step is collected'
                                % (L, L, 'an embed thread per lane consumes the detections of its batch' if args.lane_embedders else
                                   'ONE embed worker takes the detections of all lanes and launches ArcFace on the faces of several '
                                   'batches at once (>= %d crops or %.0f ms; %.0f crops per launch measured)'
                                   % (args.embed_min_crops, args.embed_max_wait * 1e3, head.get('crops_per_embed_launch', 0))),
            },
            # `roofline` = the DOMINANT kernel instance (most HIP-event time in one serial step of this process): algorithmic FLOPs per
            # launch over its average launch duration.  `roofline.all_conv_kernels` = every implicit-GEMM launch of the step together
            # (serial, and over the pipelined step `value` is made of -- above the serial figure because concurrent streams fill the
            # CUs one kernel's tail and launch gaps leave idle).
            'roofline': roofline_object(primary, klass, elapsed / steps_timed),
            'stage_ms_per_step': {k: v['ms'] for k, v in klass.items() if k != 'instances'},
            # algorithmic bytes / kernel time of the HBM-bound kernel classes (peak 8000 GB/s); post-processing
            # mixes the pose-map stream with latency-bound selection / grouping kernels
            'stage_hbm_gbps': {k: round(v['work'] / (v['ms'] * 1e-3) / 1e9, 1) for k, v in klass.items()
                               if k not in ('conv_igemm', 'instances') and v['ms'] > 0},
        }
        if args.serial:
            # --serial is the mode the rocprofv3 kernel statistics under profiles/ are taken in: the algorithmic FLOPs every
            # dense-conv kernel INSTANCE did over this whole run (warm-up, timed and event-profiled steps), so that the
            # profiler's per-kernel time table turns into TFLOP/s per template instance (profiles/summarize_round.py)
            work = {}
            for c in pipes[0].ctxs:
                for name, (n_, fl, _ms) in c.kernel_work().items():
                    a_, b_ = work.get(name, (0, 0.0))
                    work[name] = (a_ + n_, b_ + fl)
            result['kernel_work_run'] = {k_: {'launches': v[0], 'gflop': round(v[1] / 1e9, 2)} for k_, v in sorted(work.items())}
        for name, leg in legs.items():                               # both workloads of the headline mode, each with its power object
            result['value_' + name] = fps(leg['elapsed'], leg['steps'])
            result['power_' + name] = leg.get('power')
            result['ms_per_step_' + name] = round(leg['elapsed'] / leg['steps'] * 1e3, 3)
        if primary_drift is not None:
            result['decision_drift'] = primary_drift
        if 'f32' in others:                                          # the like-for-like reference arithmetic, top level
            result['value_f32'] = others['f32']['value']
            result['ms_per_step_f32'] = others['f32']['ms_per_step']
            result['roofline_f32'] = others['f32']['roofline']
            result['power_f32'] = others['f32']['power']
        if 'f16x3' in others:                                        # every network float32-grade (embeddings to 5e-7)
            result['value_f16x3'] = others['f16x3']['value']
            result['ms_per_step_f16x3'] = others['f16x3']['ms_per_step']
            result['roofline_f16x3'] = others['f16x3']['roofline']
        if 'bf16' in others:                                         # BASELINE configs[1]'s "bf16": the throughput mode, with its measured drift
            result['value_bf16'] = others['bf16']['value']
            result['decision_drift_bf16'] = others['bf16'].get('decision_drift')
        if 'f16x2' in others:                                        # opt-in tolerance mode: the embedder on two of the three products
            result['value_f16x2'] = others['f16x2']['value']
            result['ms_per_step_f16x2'] = others['f16x2']['ms_per_step']
        if 'f16' in others:                                          # opt-in tolerance mode for the embedder alone (see DTYPES['f16'])
            result['value_f16_embedder'] = others['f16']['value']
            result['ms_per_step_f16_embedder'] = others['f16']['ms_per_step']
            result['roofline_f16_embedder'] = others['f16']['roofline']
        # rank 0's GPU over the region `value` comes from, driver-run: amdgpu hwmon power1_* / freq1_input every 50 ms.  In
        # f16x3 the chip sits at its package power cap with the shader clock ~20 % under the 2.4 GHz the peak is quoted at
        # (DESIGN.md section 4); null when the box exposes no sensors
        if head.get('power') and head['power'].get('sclk_mhz_mean') and result['roofline'].get('all_conv_kernels'):
            # the peak in `roofline.peak` is quoted at 2400 MHz; under its power cap the chip holds less: the same pipelined
            # figure against the peak AT THE CLOCK IT RAN AT (informative; `frac` stays against the nominal peak)
            held = head['power']['sclk_mhz_mean']
            allc = result['roofline']['all_conv_kernels']
            allc['held_clock_mhz'] = held
            allc['pipelined_frac_at_held_clock'] = round(allc['pipelined_achieved'] / (PEAKS[primary][0] * held / 2400.0), 4)
        result['power'] = dict(head['power'], what='socket power (W) / shader clock (MHz) / hottest sensor (C) of rank 0\'s GPU over the '
                               'timed region, sysfs hwmon of its PCI function') if head.get('power') else None
        if 'ingest' in head:
            result['ingest'] = head['ingest']
        if c2_multi is not None:
            result['c2_retinaface_640'] = c2_multi
        if isinstance(head.get('ingest', {}).get('value'), float):
            result['value_ingest'] = head['ingest']['value']
        result['other_precisions'] = others
        result['other_faces_per_frame'] = other_faces
    for p in pipes:
        p.free()
    pool.shutdown()
    if rank == 0 and world == 1 and args.side:
        try:
            result['per_model'] = per_model(runtime.get_context(device_index), [primary, 'f32'] if primary != 'f32' else ['f32'],
                                            embedder_modes=args.full)
        except Exception as ex:
            import traceback
            traceback.print_exc()
            result['per_model'] = {'error': '%s: %s' % (type(ex).__name__, ex)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result['cpu_baseline'] = cpu_baseline(frames_host[:args.cpu_frames], F, sd_r, sd_a, sd_p, fallback_lm)
        except Exception as ex:
            import traceback
            traceback.print_exc()
            result['cpu_baseline'] = {'error': '%s: %s' % (type(ex).__name__, ex)}
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return result


class _GroupDist:
    """`shard.gather_results` over a specific process group (the host-side gloo group next to RCCL)."""

    def __init__(self, dist, group):
        self._dist, self._group = dist, group

    def is_initialized(self):
        return True

    def get_world_size(self):
        return self._dist.get_world_size()

    def get_rank(self):
        return self._dist.get_rank()

    def gather_object(self, obj, bucket, dst=0):
        return self._dist.gather_object(obj, bucket, dst=dst, group=self._group)


# ---- BASELINE configs C2 / C3 / C4 on one GPU (tools/model_bench.py prints the same rows) -------------------------
# Algorithmic work per unit, SURVEY.md 8(d) / BASELINE.md section 3 (conv + linear, FLOP = 2 MAC; bytes = layer-wise
# unfused activation traffic at 2 B/element + weights once per batch).
C2_BYTES_PER_IMAGE, C2_WEIGHT_BYTES = 56.3e6, 0.84e6
C3_GFLOP_PER_CROP, C4_GFLOP_PER_IMAGE, C2_GFLOP_PER_IMAGE = 24.1792, 484.634, 1.9623


def per_model(ctx, precisions, reps=8, embedder_modes=False):
    from terran_amd import arcface, openpose, retinaface, runtime, synth, weights

    def timed(fn, reps, warm=2):
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
        prof = {name: ctx.profile_read(k) for k, name in enumerate(KLASSES)}
        ctx.profile(False)
        return dt, prof
    sd_r, sd_a, sd_p = weights.make_retinaface_state(), weights.make_arcface_state(), weights.make_openpose_decoder_state()
    c2 = ctx.upload(synth.frames(1, 32, 640, 640))                                   # SURVEY.md 8(d) seeds 1 / 2 / 3
    c3 = np.random.default_rng(2).integers(0, 256, (256, 3, 112, 112), dtype=np.uint8)
    c4 = ctx.upload(synth.pose_code_frames(3, 16, 368, 656, 6))
    rows = {}
    for prec in precisions:
        det = retinaface.RetinaFace(device=ctx.device_id, state=sd_r, precision=prec, ctx=ctx)
        dt, prof = timed(lambda: det.call_frames(c2), reps)
        dev_ms = sum(p[0] for p in prof.values())
        bytes_batch = 32 * C2_BYTES_PER_IMAGE + C2_WEIGHT_BYTES
        rows['C2 RetinaFace 32x640x640 ' + prec] = {
            'images_per_s': round(32 / dt, 1), 'ms_per_batch': round(dt * 1e3, 3), 'device_kernel_ms': round(dev_ms, 3),
            'host_share': round(max(0.0, 1.0 - dev_ms * 1e-3 / dt), 3), 'kernel_launches': int(sum(p[1] for p in prof.values())),
            'conv_tflops': round(prof['conv_igemm'][2] / max(prof['conv_igemm'][0], 1e-9) / 1e9, 1),
            'roofline': {'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBPS,
                         'achieved': round(bytes_batch / (dev_ms * 1e-3) / 1e9, 1),
                         'frac': round(bytes_batch / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         'achieved_wall': round(bytes_batch / dt / 1e9, 1),
                         'algorithmic_bytes_per_image': C2_BYTES_PER_IMAGE,
                         'note': 'SURVEY.md 8(d) unfused bf16 activation bytes (56.3 MB/img + 0.84 MB weights per batch) / '
                                 'summed kernel time of the whole call (achieved) and / wall time per batch (achieved_wall)'}}
        # The wrapper call spends ~40 % of its wall time building the reference's return type (one dict of three
        # ndarray views per detection, ~4.9 k of them per batch here) under the GIL.  `packed`: the same C-ABI call
        # returning the packed (counts, boxes, landmarks, scores) arrays (RetinaFace.detect_arrays) -- from one host
        # thread, and from two threads on two contexts (what the pipelined headline does with every model).
        row = rows['C2 RetinaFace 32x640x640 ' + prec]
        try:
            from concurrent.futures import ThreadPoolExecutor
            dtp, _ = timed(lambda: det.detect_arrays(c2), reps)
            row['images_per_s_packed'] = round(32 / dtp, 1)
            row['detections_per_batch'] = int(det.detect_arrays(c2)[0].sum())
            ctx_b = runtime.new_context(ctx.device_id)
            det_b = retinaface.RetinaFace(device=ctx.device_id, state=sd_r, precision=prec, ctx=ctx_b)
            c2_b = ctx_b.upload(synth.frames(1, 32, 640, 640))
            pairs = [(det, c2), (det_b, c2_b)]

            def loop(k, n):
                d, f = pairs[k]
                for _ in range(n):
                    d.detect_arrays(f)
            with ThreadPoolExecutor(2) as ex:
                list(ex.map(lambda k: loop(k, 2), range(2)))
                t0 = time.perf_counter()
                list(ex.map(lambda k: loop(k, 2 * reps), range(2)))
                dt2 = (time.perf_counter() - t0) / (4 * reps)
            row['images_per_s_packed_two_in_flight'] = round(32 / dt2, 1)
            row['roofline']['achieved_wall_packed'] = round(bytes_batch / dtp / 1e9, 1)
            row['roofline']['achieved_wall_packed_two_in_flight'] = round(bytes_batch / dt2 / 1e9, 1)
            c2_b.free()
            det_b.model.free()
        except Exception as ex:                      # secondary figures: never take the leg down
            row['images_per_s_packed'] = 'error: %s' % ex
        det.model.free()
        arc = arcface.ArcFace(device=ctx.device_id, state=sd_a, precision=prec, ctx=ctx)
        dt, prof = timed(lambda: arc.embed_crops(c3), max(2, reps // 2))
        tf = prof['conv_igemm'][2] / max(prof['conv_igemm'][0], 1e-9) / 1e9
        peak, _, factor = PEAKS[prec]
        emb_factor = {'f16': 1, 'f16x2': 2}.get(prec, factor)      # the embedder itself: MFMAs per product
        rows['C3 ArcFace 256x3x112x112 ' + prec] = {
            'images_per_s': round(256 / dt, 1), 'ms_per_batch': round(dt * 1e3, 3), 'conv_ms': round(prof['conv_igemm'][0], 3),
            'roofline': {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': peak, 'achieved': round(tf, 1), 'frac': round(tf / peak, 4),
                         'mfma_issue_frac': round(tf * emb_factor / peak, 4), 'algorithmic_gflop_per_crop': C3_GFLOP_PER_CROP}}
        arc.model.free()
        for alt in ([m for m in ('f16x2', 'f16x3', 'f16') if m != prec] if embedder_modes and prec in ('f16x2', 'f16x3', 'f16') else []):     # the embedder's other modes beside it
            arc = arcface.ArcFace(device=ctx.device_id, state=sd_a, precision=alt, ctx=ctx)
            dt, prof = timed(lambda: arc.embed_crops(c3), max(2, reps // 2))
            tf = prof['conv_igemm'][2] / max(prof['conv_igemm'][0], 1e-9) / 1e9
            rows['C3 ArcFace 256x3x112x112 (embedder in the %s mode)' % alt] = {
                'images_per_s': round(256 / dt, 1), 'ms_per_batch': round(dt * 1e3, 3), 'conv_ms': round(prof['conv_igemm'][0], 3),
                'roofline': {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': 2500.0, 'achieved': round(tf, 1), 'frac': round(tf / 2500.0, 4),
                             'mfma_issue_frac': round(tf * {'f16': 1, 'f16x2': 2}.get(alt, 3) / 2500.0, 4), 'algorithmic_gflop_per_crop': C3_GFLOP_PER_CROP,
                             'note': 'f16: one f16 MFMA per product on 2-byte half-float activations, embeddings 3.3e-4 (wild-statistics weights 1.8e-3) vs the 1e-3 bar; '
                                     'f16x2 (opt-in, guarded at load): two per product, (w_hi + w_lo) * x_hi, embeddings 1.8e-4 (8.2e-4); '
                                     'f16x3: three per product on split-half operands, embeddings 5e-7'}}
            arc.model.free()
        pose = openpose.OpenPose(device=ctx.device_id, short_side=368, state=sd_p, precision=prec, ctx=ctx)
        res = []
        dt, prof = timed(lambda: res.append(pose.call_frames(c4)), max(2, reps // 2))
        tf = prof['conv_igemm'][2] / max(prof['conv_igemm'][0], 1e-9) / 1e9
        rows['C4 OpenPose 16x368x656 ' + prec] = {
            'images_per_s': round(16 / dt, 1), 'ms_per_batch': round(dt * 1e3, 3), 'conv_ms': round(prof['conv_igemm'][0], 3),
            'grouping_ms': round(prof['postprocess'][0], 3), 'humans_per_frame': round(sum(len(p) for p in res[-1]) / 16.0, 2),
            'roofline': {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': peak, 'achieved': round(tf, 1), 'frac': round(tf / peak, 4),
                         'mfma_issue_frac': round(tf * factor / peak, 4), 'algorithmic_gflop_per_image': C4_GFLOP_PER_IMAGE}}
        pose.model.free()
    c2.free()
    c4.free()
    return rows


def run_single_process(args):
    """--single-process: ONE process drives all devices through terran_amd.pipeline.StreamPipeline -- per device
    `--inflight` lanes of (upload, detect -> embed, pose) threads with a context each, frames sharded contiguously over
    the devices, results gathered in frame order (SURVEY.md 8e as specified).  Two legs: `value` with the shards resident
    in HBM (the per-process headline's condition) and `ingest` with every batch scattered from host memory inside the
    timed region."""
    from terran_amd import runtime
    from terran_amd.pipeline import StreamPipeline
    devices = [int(d) for d in args.devices.split(',')] if args.devices else list(range(args.gpus))
    n = len(devices)
    (sd_r, sd_a, sd_p), one, fallback_lm = make_workload(args, 0)
    frames_host = np.concatenate([one] * n) if n > 1 else one          # B frames per device per step
    prec = runtime.resolve_precision(args.precision or os.environ.get('TERRAN_AMD_PRECISION') or HEADLINE)
    F = args.faces
    sys.setswitchinterval(float(os.environ.get('TA_BENCH_SWITCH', '2e-4')))

    def pick(dets):
        return [[{'landmarks': x['landmarks']} for x in d[:F]] + [{'landmarks': fallback_lm[k]} for k in range(len(d[:F]), F)]
                for d in dets]
    L = max(1, args.inflight)
    pipe = StreamPipeline(devices, inflight=L, pick_faces=pick, shared_embedder=not args.lane_embedders, embed_min_crops=args.embed_min_crops,
                                          embed_max_crops=args.embed_max_crops, embed_max_wait=args.embed_max_wait,
                          detection_kw=dict(short_side=416, state=sd_r, precision=prec),
                          recognition_kw=dict(state=sd_a, precision=prec),
                          estimation_kw=dict(short_side=184, state=sd_p, precision=prec))
    resident = pipe.scatter(frames_host)

    def timed(batches, k):
        t0 = time.perf_counter()
        out, got = None, 0
        for out in pipe.run(batches):
            got += 1
        assert got == k
        return time.perf_counter() - t0, out
    warm = max(args.warmup, 2 * L)
    timed((resident for _ in range(warm)), warm)
    elapsed, out = timed((resident for _ in range(args.steps)), args.steps)
    # one batch alone with a HIP event pair around every launch of lane 0 of device 0 -> the per-class times of a step
    lane = pipe.lanes[0][0]
    ctxs = [lane.det.model.ctx, lane.est.model.ctx] + ([lane.rec.model.ctx] if lane.rec is not None else
                                                      [pipe.embedders[runtime.device_index(devices[0])].ctx])
    for c in ctxs:
        c.profile_reset()
        c.profile(True)
    timed(([resident[0]] + [None] * (n - 1) for _ in range(1)), 1)
    klass = {}
    for k_, name in enumerate(KLASSES):
        ms = cnt = work = 0
        for c in ctxs:
            a_, b_, w_ = c.profile_read(k_)
            ms, cnt, work = ms + a_, cnt + b_, work + w_
        klass[name] = {'ms': round(ms, 3), 'launches': cnt, 'work': work}
    for c in ctxs:
        c.profile(False)
    k_in = max(2 * L, min(args.steps, 60))
    timed((frames_host for _ in range(2 * L)), 2 * L)
    e_in, _ = timed((frames_host for _ in range(k_in)), k_in)
    for fr in resident:
        if fr is not None:
            fr.free()
    pipe.close()
    per_step = elapsed / args.steps
    roof = conv_roofline(prec, klass['conv_igemm'])
    work_step = klass['conv_igemm']['work'] * n                    # every device does one device-0 share per step
    roof['pipelined_achieved'] = round(work_step / per_step / 1e12, 2)
    roof['pipelined_frac'] = round(work_step / per_step / 1e12 / (PEAKS[prec][0] * len(set(devices))), 4)
    return {
        'metric': 'frames/sec 1080p detect+embed+pose', 'value': round(len(frames_host) * args.steps / elapsed, 3),
        'unit': 'frames/s', 'n_gpus': len(set(devices)), 'steps': args.steps, 'warmup': warm,
        'ms_per_step': round(per_step * 1e3, 3), 'timed_region_s': round(elapsed, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': DTYPES[prec], 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[4], ONE process: %d device replica(s) %s x %d lanes of upload / detect -> embed / pose '
                               'threads (terran_amd.pipeline.StreamPipeline), %d frames per replica per step, shards resident in HBM, '
                               'results gathered in frame order on the calling thread' % (n, devices, L, args.batch),
                   'precision': prec, 'frames_per_step': len(frames_host), 'faces_per_frame': F,
                   'detections_per_frame': round(float(np.mean([len(d) for d in out[0]])), 1),
                   'humans_per_frame': round(float(np.mean([len(p) for p in out[2]])), 2)},
        'roofline': roof,
        'stage_ms_per_step': {k_: v['ms'] for k_, v in klass.items()},
        'ingest': {'value': round(len(frames_host) * k_in / e_in, 3), 'unit': 'frames/s', 'steps': k_in,
                   'ms_per_step': round(e_in / k_in * 1e3, 3), 'host_to_device_mb_per_step': round(frames_host.nbytes / 1e6, 1),
                   'what': 'the same loop fed from host memory: every batch is cut into contiguous shards, each uploaded by its '
                           "lane's upload thread on its own stream (scatter), results gathered in order (gather), all inside the timed region"},
        'value_ingest': round(len(frames_host) * k_in / e_in, 3)}


def cpu_baseline(frames_host, F, sd_r, sd_a, sd_p, fallback_lm):
    """The oracle (this repo's CPU restatement of Terran's path: torch-CPU fp32 nets + numpy
    post-processing, pinned to the reference by tests/golden) on a bounded sample of the same
    workload, all host cores.  kind = "port": the reference's Python cannot travel to the GPU box."""
    import torch
    from oracle import pipeline
    # this leg is the reference's CPU path on ALL host cores: undo the rank's NUMA binding (terran_amd/affinity.py) before torch
    # builds its thread pool -- the pool's threads inherit the mask of the thread that creates them
    if hasattr(os, 'sched_setaffinity'):
        try:
            os.sched_setaffinity(0, range(os.cpu_count() or 1))
        except OSError:
            pass
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
    return {'value': round(n / dt, 4), 'unit': 'frames/s', 'cores': torch.get_num_threads(),
            'cpus_allowed': len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None, 'kind': 'port',
            'sample': '%d of the same 1080p frames through oracle.pipeline detection+recognition(top-%d)+estimation '
                      '(torch-CPU fp32, %.1f s)' % (n, F, dt),
            'stage_ms_per_frame': {'detection': round((t1 - t0) / n * 1e3, 1), 'recognition': round((t2 - t1) / n * 1e3, 1),
                                   'estimation': round((t3 - t2) / n * 1e3, 1)}}


if __name__ == '__main__':
    main()

"""`StreamPipeline`: the reference's video loop (examples/video.py:16-44)

    for frames in video:
        faces = face_detection(frames); features = extract_features(frames, faces); poses = pose_estimation(frames)

as ONE process driving a list of devices with several batches in flight -- SURVEY.md 8(e): "one host thread + HIP stream
set + pinned staging per device", frames sharded contiguously over the devices, weights replicated, no collective, host
gather in frame order.

A synchronous facade call (`facade._Fanout`) is scatter -> run -> gather with the device idle during the host parts
and only one of the three networks on the GPU at a time.  Here every device owns `inflight` LANES, a lane being what
`bench.py` runs per rank: three task threads with a context (HIP stream, scratch) and a model each -- detect ->
(queue) -> embed, and pose beside them -- plus an upload thread with its own context and stream, so H2D copies, the three
networks' kernels and the host-side result handling of different batches overlap.  Batch b's shard for device d goes to
lane b % inflight of that device; every lane works its shards in order; a collector hands out (detections, features,
poses) per batch in batch order, each concatenated over the devices in device (= frame) order.  Results equal the
one-device facades' bit for bit: frames are independent through the whole path and every conv sums in a
batch-independent order.
"""
import queue
import threading

import numpy as np

from . import lib, runtime
from .shard import shard_bounds

_STOP = object()
_WAIT = 600.0


class _Lane:
    """One (device, slot): upload thread + detect / embed / pose threads, each with its own context."""

    def __init__(self, device, make, depth, out_q, fail):
        d = runtime.device_index(device)
        self.ctx_up = runtime.new_context(d)
        ctxs = [runtime.new_context(d) for _ in range(3)]
        self.det, self.rec, self.est = make(d, ctxs)
        self.out_q, self.fail = out_q, fail
        self.in_q = queue.Queue(maxsize=depth)                  # back-pressure: at most `depth` shards waiting per lane
        self.q_det, self.q_rec, self.q_est, self.q_faces = queue.Queue(), queue.Queue(), queue.Queue(), queue.Queue()
        self.threads = [threading.Thread(target=self._guard, args=(f,), daemon=True, name='terran_amd-lane')
                        for f in (self._upload, self._detect, self._embed, self._pose)]
        for t in self.threads:
            t.start()

    def _guard(self, fn):
        try:
            fn()
        except BaseException as e:                              # noqa: BLE001  (re-raised in the consumer)
            self.fail(e)
            for q in (self.q_det, self.q_rec, self.q_est, self.q_faces):     # a lane that lost a thread stops as a whole
                q.put(_STOP)
            while True:                                         # ... and lets go of what is still queued for it
                try:
                    self.in_q.get_nowait()
                except queue.Empty:
                    break

    def _upload(self):
        while True:
            job = self.in_q.get()
            if job is _STOP:
                for q in (self.q_det, self.q_rec, self.q_est):
                    q.put(_STOP)
                return
            key, shard, pick, free_resident = job
            own = not isinstance(shard, lib.Frames)
            frames = self.ctx_up.upload(shard) if own else shard
            refs = [3, threading.Lock(), frames if (own or free_resident) else None]   # freed by whichever task finishes with it last
            for q in (self.q_det, self.q_rec, self.q_est):
                q.put((key, frames, refs, pick))

    @staticmethod
    def _release(refs):
        with refs[1]:
            refs[0] -= 1
            last = refs[0] == 0
        if last and refs[2] is not None:
            refs[2].free()

    def _detect(self):
        while True:
            job = self.q_det.get()
            if job is _STOP:
                self.q_faces.put(_STOP)
                return
            key, frames, refs, pick = job
            dets = self.det(frames)
            self.q_faces.put(pick(dets))
            self.out_q.put((key, 0, dets))
            self._release(refs)

    def _embed(self):
        while True:
            job = self.q_rec.get()
            if job is _STOP:
                return
            key, frames, refs, _ = job
            faces = self.q_faces.get(timeout=_WAIT)               # the detections of the same shard (both queues are FIFO)
            if faces is _STOP:
                return
            self.out_q.put((key, 1, self.rec(frames, faces)))
            self._release(refs)

    def _pose(self):
        while True:
            job = self.q_est.get()
            if job is _STOP:
                return
            key, frames, refs, _ = job
            self.out_q.put((key, 2, self.est(frames)))
            self._release(refs)


    def close(self):
        """After the threads have stopped: release device memory NOW (models, plans, scratch, streams) rather than whenever
        the garbage collector finds the objects -- a late hipFree waits for every stream of the process, i.e. it would stall
        whatever pipeline is running by then."""
        for facade in (self.det, self.rec, self.est):
            wrapper = facade.model
            if wrapper is None:
                continue
            for m in (getattr(wrapper, 'model', None), getattr(wrapper, '_fb_model', None)):
                if m is not None:
                    m.free()
        for c in (self.det.model.ctx, self.rec.model.ctx, self.est.model.ctx, self.ctx_up):
            c.close()


def all_faces(dets):
    """Default face selection: every detection of every frame is embedded (examples/video.py:27-31)."""
    return [[{'landmarks': x['landmarks']} for x in d] for d in dets]


class StreamPipeline:
    """devices: list of device ids (repeats allowed: several replicas on one card).  `states` / `precision` /
    `short_sides` go to the three facades of every lane.  pick_faces(dets) -> faces_per_image chooses what is embedded.

        pipe = StreamPipeline([0, 1, 2, 3, 4, 5, 6, 7])
        for dets, feats, poses in pipe.run(batches):    # batches: iterable of (N,H,W,3) uint8 arrays
            ...
        pipe.close()
    """

    def __init__(self, devices, inflight=2, depth=2, pick_faces=all_faces, detection_kw=None, recognition_kw=None,
                 estimation_kw=None, switch_interval=2e-4):
        """switch_interval: the lanes' threads spend their time inside GIL-free library calls; one that comes back must not
        wait a whole 5 ms interpreter time slice behind another thread's result handling before it can queue its next
        launches.  The interpreter-wide switch interval is lowered to this value while the pipeline lives (None: left alone)
        and restored by close()."""
        import sys
        from .facade import Detection, Recognition, Estimation
        self._old_switch = None
        if switch_interval is not None and sys.getswitchinterval() > switch_interval:
            self._old_switch = sys.getswitchinterval()
            sys.setswitchinterval(switch_interval)
        self.devices = list(devices)
        if not self.devices:
            raise ValueError('`devices` is empty')
        self.pick_faces = pick_faces
        self._err = []
        self._out = queue.Queue()
        dkw, rkw, ekw = dict(detection_kw or {}), dict(recognition_kw or {}), dict(estimation_kw or {})

        def make(d, ctxs):
            return (Detection(device=d, ctx=ctxs[0], **dkw), Recognition(device=d, ctx=ctxs[1], **rkw),
                    Estimation(device=d, ctx=ctxs[2], **ekw))
        self.lanes = [[_Lane(d, make, depth, self._out, self._fail) for _ in range(max(1, inflight))] for d in self.devices]
        self.inflight = max(1, inflight)

    def _fail(self, e):
        self._err.append(e)
        self._out.put((None, -1, e))

    def scatter(self, images):
        """Host batch -> one resident `lib.Frames` per device (contiguous shards), for `run(..., resident=True)` loops that
        re-use a batch; None where a device gets no frame."""
        images = np.asarray(images)
        k = len(self.devices)
        out = []
        for r in range(k):
            lo, hi = shard_bounds(len(images), k, r)
            out.append(self.lanes[r][0].ctx_up.upload(images[lo:hi]) if hi > lo else None)
        return out

    def run(self, batches, free_resident=False):
        """Generator: one (detections, features, poses) triple per batch, in batch order; lists over the batch's frames.
        A batch is a host array / list of equally sized frames, or a list with one resident `lib.Frames` (or None) per
        device -- what `scatter` returned, or batches a `video.RawVideoReader` uploaded; those are left alone unless
        free_resident (the pipeline then frees each one when its three tasks are through with it)."""
        k = len(self.devices)
        pending = {}                       # batch -> {(device, kind): result}
        n_shards = {}
        fed = [0]
        done_feeding = threading.Event()

        def feeder():
            try:
                for b, batch in enumerate(batches):
                    if self._err:
                        break
                    if isinstance(batch, list) and batch and all(x is None or isinstance(x, lib.Frames) for x in batch):
                        shards = list(batch)
                    else:
                        arr = np.asarray(batch)
                        shards = []
                        for r in range(k):
                            lo, hi = shard_bounds(len(arr), k, r)
                            shards.append(arr[lo:hi] if hi > lo else None)
                    n_shards[b] = sum(s is not None for s in shards)
                    fed[0] = b + 1
                    for r, s in enumerate(shards):
                        if s is not None:
                            self.lanes[r][b % self.inflight].in_q.put(((b, r), s, self.pick_faces, free_resident))
                    if n_shards[b] == 0:
                        self._out.put(((b, -1), 3, None))           # an empty batch still yields its (empty) triple
            except BaseException as e:                              # noqa: BLE001
                self._fail(e)
            finally:
                done_feeding.set()
                self._out.put((None, -2, None))
        t = threading.Thread(target=feeder, daemon=True, name='terran_amd-feeder')
        t.start()
        nxt = 0
        while True:
            if done_feeding.is_set() and nxt >= fed[0] and not self._err:
                break
            key, kind, val = self._out.get(timeout=_WAIT)
            if kind == -1:
                raise val
            if kind >= 0 and key is not None:
                b, r = key
                pending.setdefault(b, {})[(r, kind)] = val
            while nxt in n_shards and len([1 for (r, kd) in pending.get(nxt, {}) if kd < 3]) == 3 * n_shards[nxt]:
                res = pending.pop(nxt, {})
                triple = tuple([x for r in range(k) if (r, kind_) in res for x in res[(r, kind_)]] for kind_ in range(3))
                nxt += 1
                yield triple
        if self._err:
            raise self._err[0]

    def close(self):
        for lanes in self.lanes:
            for lane in lanes:
                try:
                    lane.in_q.put(_STOP, timeout=5)
                except queue.Full:                              # its upload thread is gone: the task threads were told already
                    pass
        for lanes in self.lanes:
            for lane in lanes:
                for t in lane.threads:
                    t.join(timeout=30)
                if not any(t.is_alive() for t in lane.threads):
                    lane.close()
        self.lanes = []
        if self._old_switch is not None:
            import sys
            sys.setswitchinterval(self._old_switch)
            self._old_switch = None

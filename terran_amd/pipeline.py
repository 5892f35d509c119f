"""`StreamPipeline`: the reference's video loop (examples/video.py:16-44)

    for frames in video:
        faces = face_detection(frames); features = extract_features(frames, faces); poses = pose_estimation(frames)

as ONE process driving a list of devices with several batches in flight -- SURVEY.md 8(e): "one host thread + HIP stream
set + pinned staging per device", frames sharded contiguously over the devices, weights replicated, no collective, host
gather in frame order.

A synchronous facade call (`facade._Fanout`) is scatter -> run -> gather with the device idle during the host parts
and only one of the three networks on the GPU at a time.  Here every device owns `inflight` LANES -- an upload thread, a
detect thread and a pose thread, each with its own context (HIP stream, scratch) and model -- so H2D copies, the networks'
kernels and the host-side result handling of different batches overlap.  Batch b's shard for device d goes to lane
b % inflight of that device; every lane works its shards in order.

Embedding: the reference's loop is ONE loop, so one embedder sees every face of the video.  Per device a single EMBED
WORKER (own context + ArcFace model) takes the detections of all lanes and launches on the faces of several batches at
once (>= `embed_min_crops`, bounded wait `embed_max_wait`): ArcFace's 14 x 14 / 7 x 7 layers fill the chip's 256 CUs in whole
rounds at 320 crops (stage 3: 490 tiles of 128 x 256 pixels; 392 at 256 crops = 1.5 rounds), not at the 64 one batch brings.  A face's embedding does not depend on the launch it rides in (every conv sums in a
batch-independent order), so this changes no bit of any result.  `shared_embedder=False` keeps one embed thread per lane.

A collector hands out (detections, features, poses) per batch in batch order, each concatenated over the devices in
device (= frame) order.  Results equal the one-device facades' bit for bit.

Robustness (a video loop breaks out of its generator all the time): every `run()` is a GENERATION; jobs and results carry
its id, results of an abandoned generation are dropped, its queued work is skipped, and leaving the generator stops its
feeder.  After a lane error the pipeline is dead: `run()` raises at once instead of waiting for results that cannot come.
"""
import itertools
import collections
import queue
import threading
import time

import numpy as np

from . import affinity, lib, runtime
from .shard import shard_bounds

_STOP = object()
_WAIT = 600.0


class _Refs:
    """Frames shared by the tasks of one shard: freed (when the pipeline owns them) by whichever task lets go last."""

    def __init__(self, n, frames):
        self.n, self.lock, self.frames = n, threading.Lock(), frames

    def release(self):
        with self.lock:
            self.n -= 1
            last = self.n == 0
        if last and self.frames is not None:
            self.frames.free()


class _Embedder:
    """One per device: embeds the faces of every lane's shards, several shards per launch."""

    def __init__(self, device, make_rec, out_q, fail, live, min_crops, max_crops, max_wait):
        self.device = runtime.device_index(device)
        self.out_q, self.fail, self.live = out_q, fail, live
        self.min_crops, self.max_crops, self.max_wait = int(min_crops), int(max_crops), float(max_wait)
        self.ctx = runtime.new_context(self.device)
        self.rec = make_rec(self.device, self.ctx)
        # the worker's inbox: a deque and the `upstream` count under ONE condition -- deliver() / skip() wake the worker, which
        # otherwise sleeps (no timed polling under the GIL while lanes are busy)
        self.items = collections.deque()
        self.cv = threading.Condition()
        self.upstream = 0                                   # shards uploaded on this device whose detections have not arrived here yet
        self.launches = self.crops = 0                      # statistics: crops per launch = crops / launches
        self.thread = threading.Thread(target=self._guard, daemon=True, name='terran_amd-embed')
        self.thread.start()

    def announce(self):
        """A lane uploaded a shard: its detections WILL arrive (deliver / skip)."""
        with self.cv:
            self.upstream += 1

    def deliver(self, item):
        with self.cv:
            self.items.append(item)
            self.upstream -= 1
            self.cv.notify()

    def skip(self):
        """An announced shard will NOT deliver (no faces, abandoned run, or its lane failed)."""
        with self.cv:
            self.upstream -= 1
            self.cv.notify()

    def _take(self, timeout=None, while_upstream=False):
        """The next item; None once `timeout` seconds have passed or (while_upstream) nothing is between upload and detect on
        this device any more -- an interactive / tracking loop that feeds one batch at a time: no more faces are coming,
        waiting would only add `max_wait` to every batch."""
        end = None if timeout is None else time.perf_counter() + timeout
        with self.cv:
            while not self.items:
                if while_upstream and self.upstream <= 0:
                    return None
                left = None if end is None else end - time.perf_counter()
                if left is not None and left <= 0:
                    return None
                self.cv.wait(left)
            return self.items.popleft()

    def _guard(self):
        try:
            affinity.bind(self.device)
            self._loop()
        except BaseException as e:                              # noqa: BLE001  (re-raised in the consumer)
            self.fail(e)
            with self.cv:                                       # what is still queued keeps its frames in HBM otherwise
                left, self.items = list(self.items), collections.deque()
            for it in left:
                if it is not _STOP:
                    it[4].release()

    def _loop(self):
        carry = None
        while True:
            first = carry if carry is not None else self._take()
            carry = None
            if first is _STOP:
                return
            items, n = [first], first[5]
            deadline = time.perf_counter() + self.max_wait
            stop = False
            while n < self.min_crops:                           # more shards, until enough crops or the wait is over
                nxt = self._take(deadline - time.perf_counter(), while_upstream=True)
                if nxt is None:
                    break
                if nxt is _STOP:
                    stop = True
                    break
                if n + nxt[5] > self.max_crops:
                    carry = nxt
                    break
                items.append(nxt)
                n += nxt[5]
            self._embed(items)
            if stop:
                return

    def _embed(self, items):
        todo = [it for it in items if it[0] in self.live]       # shards of an abandoned generation are let go unworked
        try:
            if todo:
                feats = self.rec.model.call_multi([(it[2], it[3]) for it in todo])
                self.launches += 1
                self.crops += sum(it[5] for it in todo)
                for it, f in zip(todo, feats):
                    self.out_q.put((it[0], it[1], 1, f))
        finally:
            for it in items:
                it[4].release()

    def close(self):
        with self.cv:
            self.items.append(_STOP)
            self.cv.notify()
        self.thread.join(timeout=30)
        if not self.thread.is_alive():
            wrapper = self.rec.model
            for m in (getattr(wrapper, 'model', None), getattr(wrapper, '_fb_model', None)):
                if m is not None:
                    m.free()
            self.ctx.close()


class _Lane:
    """One (device, slot): upload thread + detect / pose threads (+ an embed thread when the device has no shared embed
    worker), each with its own context."""

    def __init__(self, device, make, depth, out_q, fail, live, embedder):
        d = runtime.device_index(device)
        self.device = d
        self.ctx_up = runtime.new_context(d)
        ctxs = [runtime.new_context(d) for _ in range(3 if embedder is None else 2)]
        self.det, self.rec, self.est = make(d, ctxs, embedder is None)
        self.out_q, self.fail, self.live, self.embedder = out_q, fail, live, embedder
        self.dead = False
        self.in_q = queue.Queue(maxsize=depth)                  # back-pressure: at most `depth` shards waiting per lane
        self.q_det, self.q_rec, self.q_est, self.q_faces = queue.Queue(), queue.Queue(), queue.Queue(), queue.Queue()
        tasks = [self._upload, self._detect, self._pose] + ([self._embed] if embedder is None else [])
        self.threads = [threading.Thread(target=self._guard, args=(f,), daemon=True, name='terran_amd-lane') for f in tasks]
        for t in self.threads:
            t.start()

    def _guard(self, fn):
        try:
            affinity.bind(self.device)                          # host threads + pinned staging on the GPU's NUMA node
            fn()
        except BaseException as e:                              # noqa: BLE001  (re-raised in the consumer)
            self.dead = True
            self.fail(e)
            for q in (self.q_det, self.q_rec, self.q_est, self.q_faces):     # a lane that lost a thread stops as a whole
                q.put(_STOP)
            while True:                                         # ... and lets go of what is still queued for it
                try:
                    self.in_q.get_nowait()
                except queue.Empty:
                    break

    def _upload(self):
        tasks = (self.q_det, self.q_est) + ((self.q_rec,) if self.embedder is None else ())
        while True:
            job = self.in_q.get()
            if job is _STOP:
                for q in tasks:
                    q.put(_STOP)
                return
            gen, key, shard, pick, free_resident = job
            own = not isinstance(shard, lib.Frames)
            if gen not in self.live:                            # the consumer left that run: nothing to do for it
                if not own and free_resident:
                    shard.free()
                continue
            if self.dead:                                       # a task thread of this lane failed: nothing would take the shard
                if not own and free_resident:                   # (and nothing must be announced to the embed worker for it)
                    shard.free()
                continue
            frames = self.ctx_up.upload(shard) if own else shard
            refs = _Refs(3, frames if (own or free_resident) else None)
            if self.embedder is not None:
                self.embedder.announce()
            for q in tasks:
                q.put((gen, key, frames, refs, pick))

    def _detect(self):
        while True:
            job = self.q_det.get()
            if job is _STOP:
                self.q_faces.put(_STOP)
                return
            gen, key, frames, refs, pick = job
            if gen not in self.live:
                if self.embedder is None:
                    self.q_faces.put(None)                      # keeps the embed thread's two queues in step
                else:
                    self.embedder.skip()
                    refs.release()
                refs.release()
                continue
            try:
                dets = self.det(frames)
                faces = pick(dets)
            except BaseException:
                # the shard was announced to the embed worker (`upstream`): a lane that dies here must take the announcement
                # back, or every later launch of the device's embedder waits its full `max_wait` for faces that never come
                if self.embedder is not None:
                    self.embedder.skip()
                    refs.release()
                refs.release()
                raise
            if self.embedder is None:
                self.q_faces.put(faces)
            elif not any(len(f) for f in faces):                # no face in the shard: nothing to wait for, nothing to launch
                self.embedder.skip()
                self.out_q.put((gen, key, 1, [np.empty((0, 512)) for _ in faces]))
                refs.release()
            else:                                               # the embed worker lets go of the frames for its task
                self.embedder.deliver((gen, key, frames, faces, refs, sum(len(f) for f in faces)))
            self.out_q.put((gen, key, 0, dets))
            refs.release()

    def _embed(self):
        while True:
            job = self.q_rec.get()
            if job is _STOP:
                return
            gen, key, frames, refs, _ = job
            faces = self.q_faces.get(timeout=_WAIT)               # the detections of the same shard (both queues are FIFO)
            if faces is _STOP:
                return
            if faces is not None and gen in self.live:
                self.out_q.put((gen, key, 1, self.rec(frames, faces)))
            refs.release()

    def _pose(self):
        while True:
            job = self.q_est.get()
            if job is _STOP:
                return
            gen, key, frames, refs, _ = job
            if gen in self.live:
                self.out_q.put((gen, key, 2, self.est(frames)))
            refs.release()

    def close(self):
        """After the threads have stopped: release device memory NOW (models, plans, scratch, streams) rather than whenever
        the garbage collector finds the objects -- a late hipFree waits for every stream of the process, i.e. it would stall
        whatever pipeline is running by then."""
        ctxs = [self.ctx_up]
        for facade in (self.det, self.rec, self.est):
            if facade is None:
                continue
            wrapper = facade.model
            if wrapper is None:
                continue
            for m in (getattr(wrapper, 'model', None), getattr(wrapper, '_fb_model', None)):
                if m is not None:
                    m.free()
            ctxs.append(wrapper.ctx)
        for c in ctxs:
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
                 estimation_kw=None, switch_interval=2e-4, shared_embedder=True, embed_min_crops=320, embed_max_crops=512,
                 embed_max_wait=0.020):
        """switch_interval: the lanes' threads spend their time inside GIL-free library calls; one that comes back must not
        wait a whole 5 ms interpreter time slice behind another thread's result handling before it can queue its next
        launches.  The interpreter-wide switch interval is lowered to this value while the pipeline lives (None: left alone)
        and restored by close().
        shared_embedder: one embed worker per DISTINCT device takes the faces of all its lanes (see the module text);
        it launches once `embed_min_crops` faces are waiting or `embed_max_wait` seconds after the first of them arrived,
        on at most `embed_max_crops`."""
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
        self._gen = itertools.count(1)
        self._live = set()                                     # generations whose consumer is still listening
        self._running = threading.Lock()
        dkw, rkw, ekw = dict(detection_kw or {}), dict(recognition_kw or {}), dict(estimation_kw or {})

        def make(d, ctxs, with_rec):
            return (Detection(device=d, ctx=ctxs[0], **dkw), Recognition(device=d, ctx=ctxs[2], **rkw) if with_rec else None,
                    Estimation(device=d, ctx=ctxs[1], **ekw))
        self.embedders = {}
        if shared_embedder:
            for d in dict.fromkeys(runtime.device_index(x) for x in self.devices):
                self.embedders[d] = _Embedder(d, lambda dev, ctx: Recognition(device=dev, ctx=ctx, **rkw), self._out, self._fail,
                                              self._live, embed_min_crops, embed_max_crops, embed_max_wait)
        self.lanes = [[_Lane(d, make, depth, self._out, self._fail, self._live, self.embedders.get(runtime.device_index(d)))
                       for _ in range(max(1, inflight))] for d in self.devices]
        self.inflight = max(1, inflight)

    def _fail(self, e):
        self._err.append(e)
        self._out.put((None, None, -1, e))

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
        free_resident (the pipeline then frees each one when its three tasks are through with it).
        One run at a time; leaving the generator early abandons the batches still in flight (their results are dropped)."""
        if self._err:
            raise RuntimeError('this StreamPipeline is dead (a lane failed earlier); build a new one') from self._err[0]
        if not self.lanes:
            raise RuntimeError('this StreamPipeline is closed')
        if not self._running.acquire(blocking=False):
            raise RuntimeError('StreamPipeline.run: another run of this pipeline is still being consumed')
        gen = next(self._gen)
        self._live.add(gen)
        k = len(self.devices)
        pending = {}                       # batch -> {(device, kind): result}
        n_shards = {}
        fed = [0]
        done_feeding = threading.Event()
        cancelled = threading.Event()

        def feeder():
            try:
                for b, batch in enumerate(batches):
                    if self._err or cancelled.is_set():
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
                        if s is None:
                            continue
                        lane = self.lanes[r][b % self.inflight]
                        while True:                                 # a dead lane never drains its queue: do not wait for it
                            if self._err or cancelled.is_set() or lane.dead:
                                return
                            try:
                                lane.in_q.put((gen, (b, r), s, self.pick_faces, free_resident), timeout=0.2)
                                break
                            except queue.Full:
                                continue
                    if n_shards[b] == 0:
                        self._out.put((gen, (b, -1), 3, None))      # an empty batch still yields its (empty) triple
            except BaseException as e:                              # noqa: BLE001
                self._fail(e)
            finally:
                done_feeding.set()
                self._out.put((gen, None, -2, None))
        t = threading.Thread(target=feeder, daemon=True, name='terran_amd-feeder')
        t.start()
        nxt = 0
        try:
            while True:
                if self._err:
                    raise self._err[0]
                if done_feeding.is_set() and nxt >= fed[0]:
                    break
                g, key, kind, val = self._out.get(timeout=_WAIT)
                if kind == -1:
                    raise val
                if g != gen:                                        # left over from a run the consumer abandoned
                    continue
                if kind >= 0 and key is not None:
                    b, r = key
                    pending.setdefault(b, {})[(r, kind)] = val
                while nxt in n_shards and len([1 for (r, kd) in pending.get(nxt, {}) if kd < 3]) == 3 * n_shards[nxt]:
                    res = pending.pop(nxt, {})
                    triple = tuple([x for r in range(k) if (r, kind_) in res for x in res[(r, kind_)]] for kind_ in range(3))
                    nxt += 1
                    yield triple
        finally:
            cancelled.set()
            self._live.discard(gen)                                 # queued shards of this run are skipped, late results dropped
            t.join(timeout=5)
            self._running.release()

    def contexts(self):
        """Every context (HIP stream) the pipeline's threads launch on: what a caller has to sync for a hard time stamp."""
        out = [e.ctx for e in self.embedders.values()]
        for lanes in self.lanes:
            for lane in lanes:
                out.append(lane.ctx_up)
                out += [f.model.ctx for f in (lane.det, lane.rec, lane.est) if f is not None]
        return out

    def embed_stats(self):
        """(launches, crops) of the shared embed workers since construction."""
        return (sum(e.launches for e in self.embedders.values()), sum(e.crops for e in self.embedders.values()))

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
        for e in self.embedders.values():                       # after the detect threads: nothing feeds them any more
            e.close()
        for lanes in self.lanes:
            for lane in lanes:
                if not any(t.is_alive() for t in lane.threads):
                    lane.close()
        self.lanes = []
        self.embedders = {}
        if self._old_switch is not None:
            import sys
            sys.setswitchinterval(self._old_switch)
            self._old_switch = None

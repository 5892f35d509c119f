"""Frame sharding across the GPUs of one node (one process per GPU).

Frames are independent through the whole path, so the only multi-GPU machinery is a
contiguous split of a batch over ranks (keeps video order) and an order-preserving
gather of the variable-length per-frame results on the host.  No data-path collective:
`torch.distributed` (RCCL on the GPU box, gloo in CPU tests) is used for the barrier and
for gathering Python result objects only.
"""


def shard_bounds(n, world_size, rank):
    """[lo, hi) of the contiguous slice of `n` frames owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(seq, world_size, rank):
    lo, hi = shard_bounds(len(seq), world_size, rank)
    return seq[lo:hi]


def gather_results(local_results, dist=None, dst=0):
    """Concatenate per-frame result lists of all ranks in rank order on `dst`.

    `dist` is `torch.distributed` (already initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local_results)
    world, rank = dist.get_world_size(), dist.get_rank()
    bucket = [None] * world if rank == dst else None
    dist.gather_object(list(local_results), bucket, dst=dst)
    if rank != dst:
        return None
    out = []
    for part in bucket:
        out.extend(part)
    return out


def run_sharded(frames, fn, dist=None, dst=0):
    """Run `fn(frames_slice) -> list of per-frame results` on this rank's shard and gather."""
    if dist is None or not dist.is_initialized():
        return fn(frames)
    lo, hi = shard_bounds(len(frames), dist.get_world_size(), dist.get_rank())
    return gather_results(fn(frames[lo:hi]), dist, dst)


class StreamedGather:
    """The ordered gather of a run's per-step results on `dst` WHILE the run goes on: `put(step_result)` hands a finished step to a
    gather thread of this rank, which sends every `steps_per_message` of them to `dst` (`gather_results`); `finish()` flushes the
    rest and joins.  Every rank must `put` the SAME number of steps: every rank then issues the same collectives in the same
    order, whatever the ranks' relative speed.  On `dst`, `results` ends up as the list of messages, each the rank-ordered
    concatenation of the ranks' steps of that message; `steps` counts them.  `dist` = torch.distributed (or an object with its
    gather_object / get_rank / get_world_size / is_initialized: bench.py passes the gloo side group); None = single process.
    An exception of the gather thread is re-raised by `finish()`."""

    def __init__(self, dist=None, dst=0, steps_per_message=1, keep=True):
        import queue
        import threading
        self.dist, self.dst, self.n, self.keep = dist, dst, max(1, int(steps_per_message)), keep
        self.results, self.steps, self.messages = [], 0, 0
        self._q = queue.Queue()
        self._err = None
        self._thread = threading.Thread(target=self._run, name='terran_amd-gather', daemon=True)
        self._thread.start()

    def _flush(self, chunk):
        allr = gather_results(chunk, self.dist, self.dst)
        self.messages += 1
        if allr is not None:
            self.steps += len(allr)
            if self.keep:
                self.results.append(allr)

    def _run(self):
        buf = []
        try:
            while True:
                item = self._q.get()
                if item is _END:
                    break
                buf.append(item)
                if len(buf) == self.n:
                    self._flush(buf)
                    buf = []
            if buf:
                self._flush(buf)
        except BaseException as e:             # noqa: BLE001  (re-raised in finish())
            self._err = e

    def put(self, step_result):
        self._q.put(step_result)

    def finish(self, timeout=600.0):
        self._q.put(_END)
        self._thread.join(timeout)
        if self._thread.is_alive():
            raise TimeoutError('the gather thread did not finish: do all ranks put the same number of steps?')
        if self._err is not None:
            raise self._err
        return self.results


_END = object()

"""Raw-video frame batches for the MI355X path.

Mirrors the reader contract of terran/io/video/reader.py:88-162,467-501: a `rawvideo` / `rgb24` byte stream
(what `ffmpeg ... -f rawvideo -pix_fmt rgb24 pipe:` writes) is cut into batches of `batch_size` frames of
`height x width x 3` bytes; a short final read yields a partial batch, a zero-length read ends the video; one
batch is prefetched by a daemon thread (DEFAULT_READER_BUFFER_SIZE = 1, io/video/__init__.py:6).

MI355X-first differences: the thread reads straight into page-locked host buffers (double-buffered) and
uploads on its OWN context/stream, so decode, H2D and the previous batch's kernels overlap; what the consumer
gets is a `lib.Frames` batch already resident in HBM (detection, recognition and pose all read it in place).
Building the ffmpeg command line (reader.py:421-465) stays Terran's job: pass its `proc.stdout` as `stream`.
"""
import threading
from queue import Full as QueueFull, Queue

import numpy as np

DEFAULT_READER_BUFFER_SIZE = 1


class EndOfVideo(Exception):
    pass


class VideoClosed(Exception):
    pass


def read_batch(stream, out):
    """Fill `out` (batch, H, W, 3) uint8 from `stream`; returns the number of whole frames read
    (0 at end of stream).  Trailing bytes of an incomplete frame are dropped."""
    view = memoryview(out.reshape(-1))
    got = 0
    while got < len(view):
        n = stream.readinto(view[got:])
        if not n:
            break
        got += n
    return got // int(np.prod(out.shape[1:]))


class RawVideoReader:
    """Iterate device-resident frame batches from a raw rgb24 stream.

        for frames in RawVideoReader(proc.stdout, 1920, 1080, batch_size=32):
            faces = face_detection(frames)          # lib.Frames go straight into the facades
    """

    def __init__(self, stream, width, height, batch_size=32, device=None, prefetch=DEFAULT_READER_BUFFER_SIZE,
                 upload=None, framerate=None):
        self.stream, self.width, self.height, self.batch_size = stream, int(width), int(height), int(batch_size)
        self.framerate = framerate                 # frames per second of the source, as terran.io.video.Video exposes it
                                                   # (reader.py:229-236); `face_tracking(video=reader)` derives its ages from it
        self._queue = Queue(max(1, int(prefetch)))
        self._stop = threading.Event()
        self._closed = False
        self._error = None
        self._upload = upload                      # test hook: callable(ndarray) -> batch object
        self._device = device
        self._thread = threading.Thread(target=self._worker, daemon=True)
        self._thread.start()

    def _worker(self):
        ctx = bufs = None
        try:
            shape = (self.batch_size, self.height, self.width, 3)
            if self._upload is None:
                from . import affinity, runtime
                # this thread's memcpy into the pinned buffers, the buffers themselves (first touch) and the DMA out of them
                # stay on the NUMA node the GPU hangs off
                affinity.bind(runtime.device_index(self._device))
                ctx = runtime.new_context(self._device)            # own HIP stream: uploads overlap the consumer's kernels
                bufs = [ctx.pinned_array(shape) for _ in range(2)]
                arrays = [b[0] for b in bufs]
                upload = ctx.upload
            else:
                arrays = [np.empty(shape, np.uint8) for _ in range(2)]
                upload = self._upload
            i = 0
            while not self._stop.is_set():
                n = read_batch(self.stream, arrays[i])
                if n == 0:
                    break
                batch = upload(arrays[i][:n])                       # synchronous on this thread's stream
                while not self._stop.is_set():
                    try:
                        self._queue.put(batch, timeout=1.0)
                        break
                    except QueueFull:
                        continue
                i ^= 1
                if n < self.batch_size:
                    break
        except Exception as e:                                      # surfaced to the consumer
            self._error = e
        finally:
            while True:                                             # end-of-video sentinel; give up if the consumer closed
                try:
                    self._queue.put(None, timeout=0.2)
                    break
                except QueueFull:
                    if self._stop.is_set():
                        break
            if bufs:
                for _, ptr in bufs:
                    ctx.free_pinned(ptr)

    def __iter__(self):
        return self

    def __next__(self):
        if self._closed:
            raise VideoClosed()
        batch = self._queue.get()
        if batch is None:
            self._closed = True
            if self._error is not None:
                raise self._error
            raise StopIteration
        return batch

    def read(self):
        """Like `next()` but raises EndOfVideo at the end (the reference's explicit-read protocol)."""
        try:
            return next(self)
        except StopIteration:
            raise EndOfVideo()

    def close(self):
        self._stop.set()
        self._closed = True
        try:
            while self._queue.get_nowait() is not None:
                pass
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

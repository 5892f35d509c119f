"""PCIe-inclusive ingest rate note for DESIGN.md: upload of a 32-frame 1080p batch from pageable vs pinned memory."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from terran_amd import lib
ctx = lib.Context(0)
a = np.random.default_rng(0).integers(0, 256, (32, 1080, 1920, 3), dtype=np.uint8)
pin, ptr = ctx.pinned_array(a.shape)
pin[...] = a
for name, src in (('pageable', a), ('pinned', pin)):
    f = ctx.upload(src); f.free()
    t = time.perf_counter()
    for _ in range(5):
        f = ctx.upload(src); f.free()
    dt = (time.perf_counter() - t) / 5
    print('%-9s %.2f ms per 32x1080p batch  %.1f GB/s  -> %.0f frames/s ceiling' % (name, dt * 1e3, a.nbytes / dt / 1e9, 32 / dt))
ctx.free_pinned(ptr)

"""GPU busy fraction from a rocprofv3 kernel trace (rocpd sqlite): union of kernel intervals / span, and how much of
the span has 1, 2, 3+ kernels in flight.

    python tools/busy_fraction.py <results.db> [trim]

The window is found, not guessed: the trace is cut wherever the device sits idle for more than 20 ms (model loading,
packing, plan creation between legs), the segment holding the most kernel time is the pipelined timed region, and its
middle (1 - 2 trim, default trim = 0.1) is reported."""
import sqlite3
import sys


def main(db, trim=0.1):
    trim = float(trim)
    cur = sqlite3.connect(db).cursor()
    iv = sorted(cur.execute('select start, end from kernels').fetchall())
    segs, cur_seg, reach = [], [iv[0]], iv[0][1]
    for s, e in iv[1:]:
        if s - reach > 20e6:
            segs.append(cur_seg)
            cur_seg = []
        cur_seg.append((s, e))
        reach = max(reach, e)
    segs.append(cur_seg)
    seg = max(segs, key=lambda g: sum(e - s for s, e in g))
    t0, t1 = seg[0][0], max(e for _, e in seg)
    lo, hi = t0 + (t1 - t0) * trim, t1 - (t1 - t0) * trim
    ev = []
    for s, e in seg:
        if e <= lo or s >= hi:
            continue
        ev.append((max(s, lo), 1))
        ev.append((min(e, hi), -1))
    ev.sort()
    depth, last, hist = 0, lo, {}
    for t, d in ev:
        hist[depth] = hist.get(depth, 0) + (t - last)
        depth += d
        last = t
    hist[depth] = hist.get(depth, 0) + (hi - last)
    span = hi - lo
    print('%d segments; busiest %.1f ms (%d kernels), window %.1f ms; idle %.2f %%; ' %
          (len(segs), (t1 - t0) / 1e6, len(seg), span / 1e6, 100.0 * hist.get(0, 0) / span) +
          ', '.join('%d in flight %.1f %%' % (k, 100.0 * v / span) for k, v in sorted(hist.items()) if k))


if __name__ == '__main__':
    main(*sys.argv[1:])

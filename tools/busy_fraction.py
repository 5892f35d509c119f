"""GPU busy fraction from a rocprofv3 kernel trace (rocpd sqlite): union of kernel intervals / span, and how much of
the span has 1, 2, 3+ kernels in flight.   python tools/busy_fraction.py <results.db> [skip_fraction [end_fraction]]"""
import sqlite3
import sys


def main(db, skip=0.4, end=1.0):
    cur = sqlite3.connect(db).cursor()
    iv = sorted(cur.execute('select start, end from kernels').fetchall())
    t0, t1 = iv[0][0], max(e for _, e in iv)
    lo = t0 + (t1 - t0) * float(skip)            # drop model loading / warm-up at the head of the trace
    hi = t0 + (t1 - t0) * float(end)             # ... and the serial profiling step / teardown at its tail
    ev = []
    for s, e in iv:
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
    print('span %.1f ms; idle %.2f %%; ' % (span / 1e6, 100.0 * hist.get(0, 0) / span) +
          ', '.join('%d in flight %.1f %%' % (k, 100.0 * v / span) for k, v in sorted(hist.items()) if k))


if __name__ == '__main__':
    main(*sys.argv[1:])

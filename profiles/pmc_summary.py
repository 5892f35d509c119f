"""HBM traffic of the conv kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs
of the same bench.py command, as MI355X_MICROARCH.md prescribes):
    python profiles/pmc_summary.py <fetch_db> <write_db> <steps_in_run> profiles/rNN_pmc_conv.json
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane)
streaming reads, so the read side is doubled (guide section HBM)."""
import json
import sqlite3
import sys


def total(db_path, counter, like):
    cur = sqlite3.connect(db_path).cursor()
    q = ("select sum(value), count(*) from counters_collection where counter_name=? and kernel_name like ?")
    v, n = cur.execute(q, (counter, like)).fetchone()
    return float(v or 0.0), int(n or 0)


def main(fetch_db, write_db, steps, out):
    steps = int(steps)
    f, n = total(fetch_db, 'FETCH_SIZE', '%conv_igemm%')
    w, n2 = total(write_db, 'WRITE_SIZE', '%conv_igemm%')
    launches_per_step = n / steps
    res = {
        'kernel': 'conv_igemm*',
        'steps_in_run': steps,
        'launches_per_step': launches_per_step,
        'fetch_size_kib_per_step_raw': f / steps,
        'write_size_kib_per_step_raw': w / steps,
        'fetch_correction': 2.0,
        'hbm_bytes_per_step': (2.0 * f + w) * 1024 / steps,
        'hbm_bytes_per_launch': (2.0 * f + w) * 1024 / max(n, 1),
    }
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main(*sys.argv[1:5])

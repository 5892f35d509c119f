"""Summaries of one profiles/collect.sh run (rocprofv3 rocpd sqlite outputs under <collect dir>):
    python profiles/summarize_round.py gpurun_out/collect profiles/r02
writes, per precision P in {f16x3, f32}:
  <prefix>_<P>_kernel_stats.csv     per-kernel totals of the --kernel-trace --stats run (bench.py --serial)
  <prefix>_<P>_top_dispatches.csv   the 60 longest dispatches (grid, LDS, VGPRs, duration)
  <prefix>_pmc_conv_<P>.json        HBM bytes of the conv kernels (FETCH_SIZE x2 + WRITE_SIZE, separate passes; the x2 is
                                    the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md), effective shader clock
                                    (GRBM_GUI_ACTIVE / 8 XCDs / kernel time) and MFMA busy share
                                    (SQ_VALU_MFMA_BUSY_CYCLES / (busy cycles x 1024 SIMDs)) of the conv kernels
Each PMC pass ran 3 steps (1 warm-up, 1 timed, 1 event-profiled)."""
import csv
import glob
import json
import os
import sqlite3
import sys

CONV_LIKE = ['%conv_igemm%', '%conv_dwpw%', '%rf_stem%']


def find_db(d):
    c = glob.glob(os.path.join(d, '**', '*_results.db'), recursive=True)
    return c[0] if c else None


def _norm(name):
    """'void conv_igemm_split<2, 4, 4, 3, 3>(ta_conv_launch)' -> 'conv_igemm_split<2,4,4,3,3>' (the library's own naming)."""
    n = name.replace('void ', '').split('(')[0].replace(' ', '')
    return n


def kernel_tables(db_path, prefix, work=None):
    """work: {kernel instance: {'launches', 'gflop'}} of the same run (bench.py --serial prints it as `kernel_work_run`):
    adds the algorithmic GFLOP and TFLOP/s of every dense-conv template instance to the table."""
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    unit_us = 1e3 if max(r[2] for r in rows) > 1e6 else 1.0           # rocpd's top_kernels view: ns (older) or us
    with open(prefix + '_kernel_stats.csv', 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'pct', 'algorithmic_gflop', 'tflops'])
        for name, calls, total, avg, pct in rows:
            total_us = total / unit_us
            wk = (work or {}).get(_norm(name))
            gf = wk['gflop'] if wk else ''
            tf = round(wk['gflop'] / 1e3 / (total_us * 1e-6), 1) if wk and total_us > 0 else ''
            w.writerow([name, calls, round(total_us, 3), round(avg, 3), round(pct, 3), gf, tf])
    disp = list(cur.execute('select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, '
                            '(end-start) from kernels order by (end-start) desc limit 60'))
    with open(prefix + '_top_dispatches.csv', 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'grid_x', 'workgroup_x', 'lds_bytes', 'vgpr', 'agpr', 'sgpr', 'duration_ns'])
        w.writerows(disp)
    conv = [(n, c, t) for n, c, t, _, _ in rows if 'conv_' in n or 'rf_stem' in n]
    unit = 1e6 if max(r[2] for r in rows) > 1e6 else 1e3            # rocpd's top_kernels view: ns (older) or us
    return {'conv_launches': sum(c for _, c, _ in conv), 'conv_total_ms': sum(t for _, _, t in conv) / unit,
            'all_kernels_ms': sum(r[2] for r in rows) / unit, 'steps_in_trace': 5}


def pmc_total(db_path, counter):
    cur = sqlite3.connect(db_path).cursor()
    v = ns = n = 0
    for like in CONV_LIKE:
        rows = cur.execute('select c.value, (k.end - k.start) from counters_collection c join kernels k on c.dispatch_id = k.dispatch_id '
                           'where c.counter_name=? and c.kernel_name like ?', (counter, like)).fetchall()
        v += sum(r[0] for r in rows)
        ns += sum(r[1] for r in rows)
        n += len(rows)
    return float(v), float(ns), n


def pmc_per_instance(db_path, counter):
    """{kernel instance: (sum of counter values, dispatches)} of the dense-conv kernels of one --pmc pass."""
    cur = sqlite3.connect(db_path).cursor()
    out = {}
    for like in CONV_LIKE:
        for name, v in cur.execute('select kernel_name, value from counters_collection where counter_name=? and kernel_name like ?', (counter, like)):
            e = out.setdefault(_norm(name), [0.0, 0])
            e[0] += v
            e[1] += 1
    return out


def main(collect, prefix, steps=3):
    for P in ('f16x2', 'f16', 'f16x3', 'f32'):
        if not os.path.isdir(os.path.join(collect, 'kt_' + P)):
            continue
        kt = find_db(os.path.join(collect, 'kt_' + P))
        out = {}
        if kt:
            work = None
            try:                                                    # the bench line of the traced run (stdout of kt_<P>.log)
                for line in open(os.path.join(collect, 'kt_%s.log' % P)):
                    if line.startswith('{') and 'kernel_work_run' in line:
                        work = json.loads(line)['kernel_work_run']
            except (OSError, ValueError):
                pass
            out['kernel_trace'] = kernel_tables(kt, '%s_%s' % (prefix, P), work)
        pm = {}
        for C in ('FETCH_SIZE', 'WRITE_SIZE', 'GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES'):
            db = find_db(os.path.join(collect, 'pmc_%s_%s' % (C, P)))
            if db:
                pm[C] = pmc_total(db, C)
        res = {'kernels': 'conv_igemm* + conv_dwpw + rf_stem_kernel (every dense-conv launch)', 'steps_in_run': steps}
        if 'FETCH_SIZE' in pm and 'WRITE_SIZE' in pm:
            f, _, n = pm['FETCH_SIZE']
            w, _, _ = pm['WRITE_SIZE']
            res.update({'launches_per_step': n / steps, 'fetch_size_kib_per_step_raw': f / steps,
                        'write_size_kib_per_step_raw': w / steps, 'fetch_correction': 2.0,
                        'hbm_bytes_per_step': (2.0 * f + w) * 1024 / steps, 'hbm_bytes_per_launch': (2.0 * f + w) * 1024 / max(n, 1)})
        if 'GRBM_GUI_ACTIVE' in pm and 'SQ_VALU_MFMA_BUSY_CYCLES' in pm:
            cyc, ns, _ = pm['GRBM_GUI_ACTIVE']
            busy, ns2, _ = pm['SQ_VALU_MFMA_BUSY_CYCLES']
            res.update({'effective_clock_ghz': cyc / 8.0 / ns, 'conv_kernel_ms_per_step': ns / 1e6 / steps,
                        'mfma_busy_frac_of_simd_cycles': (busy * (ns / ns2)) / (cyc / 8.0 * 1024.0)})
        # the same two traffic counters per kernel INSTANCE (bench.py: `roofline.traffic` of the dominant instance)
        dbf, dbw = find_db(os.path.join(collect, 'pmc_FETCH_SIZE_' + P)), find_db(os.path.join(collect, 'pmc_WRITE_SIZE_' + P))
        if dbf and dbw:
            pf, pw = pmc_per_instance(dbf, 'FETCH_SIZE'), pmc_per_instance(dbw, 'WRITE_SIZE')
            res['per_instance'] = {k: {'launches': pf[k][1], 'hbm_bytes_per_launch': round((2.0 * pf[k][0] + pw.get(k, [0.0, 0])[0]) * 1024 / max(pf[k][1], 1))}
                                   for k in sorted(pf)}
        res.update(out)
        json.dump(res, open('%s_pmc_conv_%s.json' % (prefix, P), 'w'), indent=1)
        print(P, json.dumps(res))


def detector_table(collect, prefix, forwards=3):
    """Per kernel of ONE detector forward at C2 (tools/detector_profile.py 32 640 640, the last of its `forwards` passes): HBM-side
    bytes read (FETCH_SIZE, KiB, doubled per the gfx950 note of MI355X_MICROARCH.md) and written (WRITE_SIZE), separate --pmc passes,
    in launch order -> <prefix>_pmc_detector.json.  `algorithmic` = SURVEY.md 8(d)'s 56.3 MB per image."""
    rows = {}
    for C in ('FETCH_SIZE', 'WRITE_SIZE'):
        db = find_db(os.path.join(collect, 'pmc_det_' + C))
        if not db:
            return
        cur = sqlite3.connect(db).cursor()
        r = cur.execute('select dispatch_id, kernel_name, value, duration from counters_collection where counter_name=? order by dispatch_id', (C,)).fetchall()
        r = [x for x in r if 'conv_' in x[1] or 'rf_' in x[1] or 'dwconv' in x[1] or 'maxpool' in x[1] or 'copych' in x[1]]
        per = len(r) // forwards
        rows[C] = r[-per:]
    ops = []
    for (d0, name, f, dur), (_, name2, w, _) in zip(rows['FETCH_SIZE'], rows['WRITE_SIZE']):
        assert _norm(name) == _norm(name2)
        ops.append({'kernel': _norm(name), 'us': round(dur / 1e3, 1), 'read_mb': round(2.0 * f * 1024 / 1e6, 2), 'write_mb': round(w * 1024 / 1e6, 2)})
    tot_r, tot_w, tot_us = sum(o['read_mb'] for o in ops), sum(o['write_mb'] for o in ops), sum(o['us'] for o in ops)
    res = {'what': 'RetinaFace forward at C2 (32 x 640 x 640, f16x3 program), one pass, kernels in launch order; the durations come from the '
                   'counter passes (serialised dispatches, slower than an un-profiled run)',
           'ops': ops, 'total_read_mb': round(tot_r, 1), 'total_write_mb': round(tot_w, 1), 'total_us_under_pmc': round(tot_us, 1),
           'algorithmic_mb': round(32 * 56.3 + 0.84, 1), 'traffic_over_algorithmic': round((tot_r + tot_w) / (32 * 56.3 + 0.84), 3)}
    json.dump(res, open(prefix + '_pmc_detector.json', 'w'), indent=1)
    print('detector', json.dumps({k: v for k, v in res.items() if k != 'ops'}))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
    detector_table(sys.argv[1], sys.argv[2])

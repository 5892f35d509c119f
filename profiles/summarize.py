"""Turn a rocprofv3 (ROCm 7.2, default rocpd sqlite output) kernel trace into the committed summaries:
    python profiles/summarize.py gpurun_out/prof_r1/r1_results.db profiles/r01
writes <prefix>_kernel_stats.csv (name, calls, total_us, avg_us, pct) and <prefix>_top_dispatches.csv."""
import csv
import sqlite3
import sys


def main(db_path, prefix):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    with open(prefix + '_kernel_stats.csv', 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'pct'])
        for name, calls, total, avg, pct in rows:
            w.writerow([name, calls, round(total / 1e3, 3) if total > 1e6 else round(total, 3), round(avg, 3), round(pct, 3)])
    disp = list(cur.execute('select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, '
                            '(end-start) from kernels order by (end-start) desc limit 60'))
    with open(prefix + '_top_dispatches.csv', 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'grid_x', 'workgroup_x', 'lds_bytes', 'vgpr', 'agpr', 'sgpr', 'duration_ns'])
        w.writerows(disp)
    print('wrote', prefix + '_kernel_stats.csv', len(rows), 'kernels')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])

set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/steppmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for P in f32 bf16x3; do
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/${P}_clk -o c -- python $R/bench.py --steps 1 --warmup 1 --serial --no-cpu-baseline --single-mode --precision $P > $O/${P}_clk.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/${P}_mfma -o c -- python $R/bench.py --steps 1 --warmup 1 --serial --no-cpu-baseline --single-mode --precision $P > $O/${P}_mfma.log 2>&1
  python - $O/${P}_clk/c_results.db $O/${P}_mfma/c_results.db $P <<'PY'
import sqlite3, sys
def tot(db, counter, like):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select c.value, (k.end - k.start) from counters_collection c join kernels k on c.dispatch_id = k.dispatch_id "
                       "where c.counter_name=? and c.kernel_name like ?", (counter, like)).fetchall()
    return sum(r[0] for r in rows), sum(r[1] for r in rows), len(rows)
for like, name in (('%conv_igemm%', 'all conv kernels'), ('%', 'all kernels')):
    cyc, ns, n = tot(sys.argv[1], 'GRBM_GUI_ACTIVE', like)
    busy, ns2, n2 = tot(sys.argv[2], 'SQ_VALU_MFMA_BUSY_CYCLES', like)
    print('%s %-16s: %d dispatches, %.2f ms; clock %.2f GHz; MFMA busy %.1f %% of SIMD cycles' % (sys.argv[3], name, n, ns / 1e6, cyc / 8.0 / ns, 100 * (busy * (ns / ns2)) / (cyc / 8.0 * 1024.0)))
PY
done

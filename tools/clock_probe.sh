#!/bin/bash
# Hardware-counter view of the conv kernels on the four shapes of tools/conv_bench.py, per precision mode:
#  * effective shader clock = GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs) / 8 / kernel duration
#    (MI355X_MICROARCH.md, "DVFS give-back");
#  * MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES (summed over all SIMDs) / (busy cycles per XCD x 1024 SIMDs).
# Two separate --pmc passes (one counter each), kernel trace only.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/clock
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for P in f32 f16x3 bf16x3; do
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/${P}_clk -o c -- python $R/tools/conv_bench.py $P > $O/${P}_clk.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/${P}_mfma -o c -- python $R/tools/conv_bench.py $P > $O/${P}_mfma.log 2>&1
  python - $O/${P}_clk/c_results.db $O/${P}_mfma/c_results.db $P <<'PY'
import sqlite3, sys
def tot(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select c.value, (k.end - k.start) from counters_collection c join kernels k on c.dispatch_id = k.dispatch_id "
                       "where c.counter_name=? and c.kernel_name like '%conv_igemm_split%'", (counter,)).fetchall()
    return sum(r[0] for r in rows), sum(r[1] for r in rows), len(rows)
cyc, ns, n = tot(sys.argv[1], 'GRBM_GUI_ACTIVE')
busy, ns2, n2 = tot(sys.argv[2], 'SQ_VALU_MFMA_BUSY_CYCLES')
ghz = cyc / 8.0 / ns
# the two passes run the same launches; scale busy cycles to the first pass' duration
util = (busy * (ns / ns2)) / (cyc / 8.0 * 1024.0)
print('%s: %d split-kernel dispatches, %.3f ms; effective clock %.2f GHz; MFMA busy %.1f %% of SIMD cycles' % (sys.argv[3], n, ns / 1e6, ghz, 100 * util))
PY
done

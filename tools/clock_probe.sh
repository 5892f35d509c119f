#!/bin/bash
# Effective shader clock under the conv kernels: GRBM_GUI_ACTIVE (cycles the GPU was busy) / kernel duration, per
# precision mode, on the four shapes of tools/conv_bench.py (MI355X_MICROARCH.md, "DVFS give-back").
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/clock
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for P in f32 bf16x3; do
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/$P -o c -- python $R/tools/conv_bench.py $P > $O/$P.log 2>&1
  python - $O/$P/c_results.db $P <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select c.kernel_name, c.value, (k.end - k.start) from counters_collection c join kernels k "
                   "on c.dispatch_id = k.dispatch_id where c.counter_name='GRBM_GUI_ACTIVE' and c.kernel_name like '%conv_igemm_pipe%'").fetchall()
cyc = sum(r[1] for r in rows); ns = sum(r[2] for r in rows)
print('%s: %d pipe-kernel dispatches, %.3e busy cycles over %.3f ms -> effective clock %.2f GHz' % (sys.argv[2], len(rows), cyc, ns / 1e6, cyc / ns))
PY
done

#!/bin/bash
# SQ wait-state counters of the conv kernels of tools/conv_bench.py (tools only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/conv_pmc
mkdir -p $O
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/$T -o c -- python $R/tools/conv_bench.py f16x3 > $O/$T.log 2>&1
done
python - <<PY
import sqlite3, glob
for db in sorted(glob.glob('$O/*/**/*_results.db', recursive=True)):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select c.kernel_name, c.grid_size_x, c.counter_name, count(*), avg(c.value), avg(k.end-k.start) from counters_collection c join kernels k on c.dispatch_id=k.dispatch_id where c.kernel_name like '%conv_igemm_split%' group by c.kernel_name, c.grid_size_x, c.counter_name").fetchall()
    for r in rows: print(r[0][:40], r[1], r[2], r[3], '%.4g' % r[4], '%.0f' % r[5])
PY

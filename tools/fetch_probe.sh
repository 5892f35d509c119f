#!/bin/bash
# Which of {precision, tile shape} changes the L2-miss (fabric) read traffic of one conv layer?  FETCH_SIZE per launch of the
# four dominant layer shapes (tools/conv_bench.py) under forced kernel variants.  Separate --pmc passes, kernel trace only.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/fetchprobe
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for P in f32 f16x3 bf16x3; do
  for V in 4 6; do
    TA_CONV_PREFER=$V timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${P}_v$V -o c -- python $R/tools/conv_bench.py $P > $O/${P}_v$V.log 2>&1
  done
done
python - $O <<'PY'
import sqlite3, sys, glob, os
for d in sorted(glob.glob(os.path.join(sys.argv[1], '*_v*'))):
    if not os.path.isdir(d): continue
    db = glob.glob(os.path.join(d, '**', '*_results.db'), recursive=True)
    if not db: continue
    cur = sqlite3.connect(db[0]).cursor()
    rows = cur.execute("select c.kernel_name, k.grid_x, count(*), sum(c.value) from counters_collection c join kernels k on c.dispatch_id = k.dispatch_id "
                       "where c.counter_name='FETCH_SIZE' and c.kernel_name like '%conv_igemm_split%' group by c.kernel_name, k.grid_x order by sum(c.value) desc").fetchall()
    print(os.path.basename(d))
    for name, grid, n, v in rows[:8]:
        print('   %-52s grid %8d  %4d launches  FETCH_SIZE x2 = %8.1f MB per launch' % (name[:52], grid, n, 2 * v * 1024 / n / 1e6))
PY

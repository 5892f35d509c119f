#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench line, rocprofv3 kernel stats and the PMC passes
# (FETCH_SIZE / WRITE_SIZE / GRBM_GUI_ACTIVE / SQ_VALU_MFMA_BUSY_CYCLES in SEPARATE runs, --kernel-trace only, as
# MI355X_MICROARCH.md prescribes) per precision.  Every step runs under its own `timeout`.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/collect
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err
SER="--steps 3 --warmup 1 --serial --no-cpu-baseline --single-mode"
for P in f16x3 f16x2 f32; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_$P -o k -- python $R/bench.py $SER --precision $P > $O/kt_$P.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do
    timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_${C}_$P -o c -- python $R/bench.py --steps 1 --warmup 1 --serial --no-cpu-baseline --single-mode --precision $P > $O/pmc_${C}_$P.log 2>&1
  done
done
# the detector alone at C2 (32 x 640 x 640): HBM-side bytes per kernel
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_det_$C -o c -- python $R/tools/detector_profile.py 32 640 640 f16x3 > $O/pmc_det_$C.log 2>&1
done
# the pipelined headline under a kernel trace: how much of the timed region has 0 / 1 / 2 / 3+ kernels in flight
timeout 600 rocprofv3 --kernel-trace -d $O/kt_pipelined -o k -- python $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline --single-mode > $O/kt_pipelined.log 2>&1
timeout 120 python $R/tools/busy_fraction.py $O/kt_pipelined/k_results.db > $O/pipelined_busy.txt 2>&1
timeout 600 python $R/profiles/summarize_round.py $O $O/summary 2>&1 | tail -40
ls $O | head -40

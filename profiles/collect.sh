#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench line, rocprofv3 kernel stats and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes) per precision.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/collect
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err
for P in bf16x3 f32; do
  rocprofv3 --kernel-trace --stats -d $O/kt_$P -o k -- python $R/bench.py --steps 3 --warmup 1 --serial --no-cpu-baseline --single-mode --precision $P > $O/kt_$P.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_$P -o f -- python $R/bench.py --steps 1 --warmup 1 --serial --no-cpu-baseline --single-mode --precision $P > $O/fetch_$P.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write_$P -o w -- python $R/bench.py --steps 1 --warmup 1 --serial --no-cpu-baseline --single-mode --precision $P > $O/write_$P.log 2>&1
done
ls -R $O | head -30

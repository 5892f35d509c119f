set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmcq
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/conv_bench.py bf16x3 2>&1 | grep -v amdgpu
python $R/bench.py --no-cpu-baseline --single-mode 2>/dev/null | tail -1 | cut -c1-160
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --single-mode > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --single-mode > $O/write.log 2>&1
python $R/profiles/pmc_summary.py $O/fetch/f_results.db $O/write/w_results.db 3 $O/pmc.json

mkdir -p gpurun_out/r06
export TMPDIR=/tmp
( echo "## --inflight 2 vs 3"; bash tools/exp_ab.sh "--inflight 2" "--inflight 3" 3; echo "## --inflight 1 vs 2"; bash tools/exp_ab.sh "--inflight 1" "--inflight 2" 2 ) > gpurun_out/r06/ab_inflight.txt 2>&1
cat gpurun_out/r06/ab_inflight.txt
timeout 600 python bench.py --inflight 2 --no-cpu-baseline --detail gpurun_out/r06/bench_inflight2_detail.json > gpurun_out/r06/bench_inflight2.json 2>/dev/null; cat gpurun_out/r06/bench_inflight2.json | cut -c1-1500
bash tools/rehearse_world8.sh > gpurun_out/r06/rehearse_world8.log 2>&1; tail -c 1500 gpurun_out/r06/rehearse_world8.log

mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06/gputests1.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r06/gputests1.log
timeout 300 python tools/smi_probe.py > gpurun_out/r06/smi_probe.txt 2>&1; echo "probe rc=$?"
timeout 900 python bench.py > gpurun_out/r06/bench1.json 2> gpurun_out/r06/bench1.err; echo "bench rc=$?"
wc -c gpurun_out/r06/bench1.json; cat gpurun_out/r06/bench1.json
cp gpurun_out/bench_detail.json gpurun_out/r06/bench1_detail.json

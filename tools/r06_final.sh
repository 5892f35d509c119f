# the round's last GPU session: the whole -m gpu suite, the profile collection (bench line, kernel stats, PMC passes) and the 5x decisions run, all on the final tree
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06/gputests_final.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06/gputests_final.log
bash profiles/collect.sh > gpurun_out/r06/collect.log 2>&1; tail -5 gpurun_out/r06/collect.log
cat gpurun_out/collect/bench.json
TA_DECISIONS_SCALE=5 timeout 1500 python -m pytest tests/test_gpu_decisions_vs_oracle.py -m gpu -q -s > gpurun_out/r06/decisions_x5.txt 2>&1; echo "decisions x5 rc=$?"; tail -3 gpurun_out/r06/decisions_x5.txt
cp gpurun_out/decisions_vs_oracle.json gpurun_out/r06/decisions_vs_oracle_x5.json

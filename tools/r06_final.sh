# the round's last GPU session: the whole -m gpu suite, the fuzzers, the profile collection (bench line, kernel stats, PMC passes), the
# multi-rank rehearsals and the 5x decisions run, all on the final tree
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06/gputests_final.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06/gputests_final.log
( timeout 600 python tests/fuzz_conv.py 600 606; timeout 600 python tests/fuzz_post.py 600 606 ) > gpurun_out/r06/fuzz_tail.txt 2>&1; tail -4 gpurun_out/r06/fuzz_tail.txt
bash profiles/collect.sh > gpurun_out/r06/collect.log 2>&1; tail -3 gpurun_out/r06/collect.log
cat gpurun_out/collect/bench.json
bash tools/rehearse_multirank.sh > gpurun_out/r06/rehearse_multirank.log 2>&1
bash tools/rehearse_world8.sh > gpurun_out/r06/rehearse_world8.log 2>&1; tail -c 600 gpurun_out/r06/rehearse_world8.log
TA_DECISIONS_SCALE=5 timeout 1500 python -m pytest tests/test_gpu_decisions_vs_oracle.py -m gpu -q -s > gpurun_out/r06/decisions_x5.txt 2>&1; echo "decisions x5 rc=$?"; tail -3 gpurun_out/r06/decisions_x5.txt
cp gpurun_out/decisions_vs_oracle.json gpurun_out/r06/decisions_vs_oracle_x5.json

mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_precision_flips.py tests/test_gpu_real_checkpoints.py tests/test_gpu_stream_pipeline.py tests/test_gpu_stress.py tests/test_tracking.py -m gpu -x -q > gpurun_out/r06/gputests2.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r06/gputests2.log
bash tools/exp_ab.sh "TERRAN_AMD_SPIN_WAIT=1" "TERRAN_AMD_SPIN_WAIT=0" 3 > gpurun_out/r06/ab_spin_vs_block.txt 2>&1
cat gpurun_out/r06/ab_spin_vs_block.txt

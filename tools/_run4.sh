mkdir -p gpurun_out/r04
timeout 1500 python tests/probe_wild_weights.py --frames 32 --stats wild > gpurun_out/r04/wild_probe2.txt 2> gpurun_out/r04/wild_probe2.err; echo probe rc=$?
tail -3 gpurun_out/r04/wild_probe2.err
cat gpurun_out/r04/wild_probe2.txt
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_decisions_vs_oracle.py 2>&1 | tail -40

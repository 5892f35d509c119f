mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_f16x3_range.py tests/test_gpu_conv.py tests/test_gpu_nets.py -x -q -m gpu 2>&1 | tail -40
timeout 1500 python tests/probe_wild_weights.py --frames 32 > gpurun_out/r04/wild_probe.txt 2> gpurun_out/r04/wild_probe.err; echo probe rc=$?
tail -5 gpurun_out/r04/wild_probe.err
cat gpurun_out/r04/wild_probe.txt

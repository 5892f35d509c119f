mkdir -p gpurun_out/r04
python tools/amax_debug.py arcface f16x3 --stem-gain 22 2>&1 | tail -16
python tools/amax_debug.py arcface f16x3 --wild 2>&1 | tail -8
python tools/amax_debug.py openpose f16x3 --wild 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_gpu_f16x3_range.py tests/test_gpu_conv.py tests/test_gpu_nets.py tests/test_gpu_conv_variants.py -x -q -m gpu 2>&1 | tail -30
timeout 1500 python tests/probe_wild_weights.py --frames 32 --stats wild > gpurun_out/r04/wild_probe3.txt 2> gpurun_out/r04/wild_probe3.err; echo probe rc=$?
tail -3 gpurun_out/r04/wild_probe3.err
grep -v "^  retinaface:\|^  arcface:\|^  openpose:" gpurun_out/r04/wild_probe3.txt

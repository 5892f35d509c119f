mkdir -p gpurun_out/r04
timeout 2400 python tests/fuzz_conv.py 900 41 > gpurun_out/r04/fuzz_conv.txt 2>&1; tail -4 gpurun_out/r04/fuzz_conv.txt
timeout 1500 python tests/fuzz_post.py 1500 43 > gpurun_out/r04/fuzz_post.txt 2>&1; tail -4 gpurun_out/r04/fuzz_post.txt

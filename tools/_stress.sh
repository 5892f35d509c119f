mkdir -p gpurun_out/stress
for seed in 4242 777; do
  timeout 900 python tests/fuzz_conv.py 1500 $seed > gpurun_out/stress/fuzz_conv_$seed.txt 2>&1
  tail -2 gpurun_out/stress/fuzz_conv_$seed.txt
done
timeout 900 python tests/fuzz_post.py 2500 99 > gpurun_out/stress/fuzz_post_99.txt 2>&1; tail -2 gpurun_out/stress/fuzz_post_99.txt
for i in 1 2 3 4; do
  timeout 900 python -m pytest tests/test_gpu_conv_variants.py tests/test_gpu_stream_pipeline.py tests/test_gpu_pipeline.py -q -m gpu -x -p no:cacheprovider > gpurun_out/stress/repeat_$i.txt 2>&1
  tail -1 gpurun_out/stress/repeat_$i.txt
done

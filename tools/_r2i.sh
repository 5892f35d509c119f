mkdir -p gpurun_out/r2i
timeout 900 python -m pytest tests/test_gpu_conv_variants.py -m gpu -q -x -k "pool or pose7x7_grouped_c5" > gpurun_out/r2i/variants.log 2>&1; tail -4 gpurun_out/r2i/variants.log
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py -m gpu -q -x > gpurun_out/r2i/nets.log 2>&1; tail -4 gpurun_out/r2i/nets.log
timeout 300 python tools/layer_profile.py bf16x3 2> gpurun_out/r2i/layers.txt; grep "model kind" gpurun_out/r2i/layers.txt
timeout 600 python bench.py --steps 60 --warmup 6 --single-mode --no-cpu-baseline > gpurun_out/r2i/bench.json 2>/dev/null; cut -c1-200 gpurun_out/r2i/bench.json

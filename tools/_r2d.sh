mkdir -p gpurun_out/r2d
python -m pytest tests/test_gpu_conv_variants.py -m gpu -q -s > gpurun_out/r2d/variants.log 2>&1; tail -4 gpurun_out/r2d/variants.log
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_nets.py -m gpu -q -s > gpurun_out/r2d/pipeline.log 2>&1; tail -15 gpurun_out/r2d/pipeline.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s > gpurun_out/r2d/fullsize.log 2>&1; tail -15 gpurun_out/r2d/fullsize.log
python -m pytest tests/test_gpu_precision_flips.py -m gpu -q -s > gpurun_out/r2d/flips.log 2>&1; tail -15 gpurun_out/r2d/flips.log

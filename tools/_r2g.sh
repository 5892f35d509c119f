mkdir -p gpurun_out/r2g
timeout 600 python -m pytest tests/test_gpu_nets.py -m gpu -q -s -k retinaface > gpurun_out/r2g/nets.log 2>&1; tail -5 gpurun_out/r2g/nets.log
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r2g/pipe.log 2>&1; tail -5 gpurun_out/r2g/pipe.log
timeout 300 python tools/layer_profile.py f32 2> gpurun_out/r2g/layers_f32.txt
TERRAN_AMD_NO_FUSED_DETECTOR=1 timeout 300 python tools/layer_profile.py f32 2> gpurun_out/r2g/layers_f32_nofuse.txt
grep "model kind 1" gpurun_out/r2g/layers_f32.txt gpurun_out/r2g/layers_f32_nofuse.txt
timeout 600 python bench.py --steps 40 --warmup 6 --single-mode --no-cpu-baseline > gpurun_out/r2g/bench.json 2>/dev/null; cut -c1-200 gpurun_out/r2g/bench.json

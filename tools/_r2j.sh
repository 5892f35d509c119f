mkdir -p gpurun_out/r2j
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x > gpurun_out/r2j/pipe.log 2>&1; tail -4 gpurun_out/r2j/pipe.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_precision_flips.py -m gpu -q -x > gpurun_out/r2j/full.log 2>&1; tail -4 gpurun_out/r2j/full.log
timeout 300 python tests/fuzz_post.py 300 > gpurun_out/r2j/fuzz.log 2>&1; tail -3 gpurun_out/r2j/fuzz.log
timeout 600 python bench.py --steps 60 --warmup 6 --single-mode --no-cpu-baseline > gpurun_out/r2j/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2j/bench.json')); print(d['value'], d['stage_ms_per_step'], d['config']['humans_per_frame'])"

mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_stream_pipeline.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python bench.py --single-mode --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r04/ab_shared.json 2> gpurun_out/r04/ab_shared.err; echo rc=$?
timeout 600 python bench.py --single-mode --no-cpu-baseline --steps 20 --warmup 3 --lane-embedders > gpurun_out/r04/ab_lanes.json 2> gpurun_out/r04/ab_lanes.err; echo rc=$?
timeout 600 python bench.py --single-mode --no-cpu-baseline --steps 20 --warmup 3 --precision f16 > gpurun_out/r04/ab_shared_f16.json 2> gpurun_out/r04/ab_shared_f16.err; echo rc=$?
timeout 600 python bench.py --single-mode --no-cpu-baseline --steps 20 --warmup 3 --precision f16 --lane-embedders > gpurun_out/r04/ab_lanes_f16.json 2> gpurun_out/r04/ab_lanes_f16.err; echo rc=$?
python - <<'PY'
import json
for f in ('ab_shared','ab_lanes','ab_shared_f16','ab_lanes_f16'):
    try:
        d=json.load(open('gpurun_out/r04/%s.json'%f))
        print(f, d['value'], d['ms_per_step'], d['timed_steps'], d['value_k_steps'], d['config']['step_overlap'][-120:])
    except Exception as e:
        print(f, 'ERR', e); print(open('gpurun_out/r04/%s.err'%f).read()[-1500:])
PY

mkdir -p gpurun_out/final4
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/final4/pytest_gpu.txt 2>&1
tail -3 gpurun_out/final4/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/final4/smoke.txt 2>&1
tail -1 gpurun_out/final4/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/final4/bench.json 2> gpurun_out/final4/bench.err
python - <<'PY'
import json
for l in open('gpurun_out/final4/bench.json'):
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d.get(k) for k in ('value', 'steps', 'timed_steps', 'value_k_steps', 'value_f16_embedder', 'value_f32', 'value_ingest')}, d['roofline']['frac'], d['power']['power_w_mean'], d['power']['sclk_mhz_mean'])
PY
timeout 300 python tools/detector_profile.py 32 640 640 f16x3 2>&1 | tail -2

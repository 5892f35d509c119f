mkdir -p gpurun_out/r04
run() { name=$1; shift; timeout 600 python bench.py --single-mode --no-cpu-baseline --steps 20 --warmup 3 "$@" > gpurun_out/r04/$name.json 2> gpurun_out/r04/$name.err; }
run sw_lanes --lane-embedders
run sw_128_5 --embed-min-crops 128 --embed-max-wait 0.005
run sw_192_10
run sw_256_20 --embed-min-crops 256 --embed-max-crops 512 --embed-max-wait 0.020
run sw_384_30 --embed-min-crops 384 --embed-max-crops 640 --embed-max-wait 0.030
run sw_lanes2 --lane-embedders
run sw_192_10_il3 --inflight 3
run sw_192_10_il6 --inflight 6
python - <<'PY'
import json
for f in ('sw_lanes','sw_128_5','sw_192_10','sw_256_20','sw_384_30','sw_lanes2','sw_192_10_il3','sw_192_10_il6'):
    try:
        d=json.load(open('gpurun_out/r04/%s.json'%f))
        print(f, d['value'], d['ms_per_step'], d['timed_steps'], d['value_k_steps'], d['config']['step_overlap'][-140:-80])
    except Exception as e:
        print(f, 'ERR', e); print(open('gpurun_out/r04/%s.err'%f).read()[-1500:])
PY

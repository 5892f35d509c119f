mkdir -p gpurun_out/r06
export TMPDIR=/tmp
for i in 1 2; do
timeout 600 python bench.py --detail gpurun_out/r06/bench_final${i}_detail.json > gpurun_out/r06/bench_final${i}.json 2> gpurun_out/r06/bench_final${i}.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r06/bench_final${i}.json')); print({k:d.get(k) for k in ('value','value_k_steps','value_f32','value_f16x2','value_ingest')}, d['power'], d['ingest'], d['roofline']['frac'], d['cpu_baseline']['value'])"
done

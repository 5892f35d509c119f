# last check of a round: the whole -m gpu suite, smoke and one default bench run on the tree as committed
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r06/gputests_verify.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r06/gputests_verify.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --detail gpurun_out/r06/bench_verify_detail.json > gpurun_out/r06/bench_verify.json 2> gpurun_out/r06/bench_verify.err; echo "bench rc=$?"; wc -c gpurun_out/r06/bench_verify.json; python -c "
import json; d=json.load(open('gpurun_out/r06/bench_verify.json')); print({k:d.get(k) for k in ('value','value_k_steps','value_f32','value_f16x2','value_ingest')}, d['roofline']['frac'], d['power'], d['per_model']['C2 RetinaFace 32x640x640 f16x3'])"

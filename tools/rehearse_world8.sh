# World-8 host rehearsal on ONE card (the only 8-rank evidence obtainable on a 1-GPU lease): 8 ranks over gloo share device 0.
# Records per-rank host CPU-seconds per step, thread counts, core placement and the rank-0 gather time; throughput at 8 ranks on
# one GPU says nothing about 8 GPUs and is not quoted.  -> profiles/r05_multirank_world8_gloo.json
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
TA_BENCH_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29518 \
  bench.py --gpus 8 --steps 3 --warmup 2 --batch 8 --inflight 2 --no-cpu-baseline --no-side-legs --sustain-seconds 0 --side-seconds 0.3 \
  > gpurun_out/r05/dist_world8_gloo.json 2> gpurun_out/r05/dist_world8_gloo.err
echo "rc=$?"
tail -c 1500 gpurun_out/r05/dist_world8_gloo.err
python - <<'EOF'
import json
d = json.load(open('gpurun_out/r05/dist_world8_gloo.json'))
print(json.dumps({k: d.get(k) for k in ('n_gpus', 'value', 'ms_per_step', 'value_ingest')}))
print(json.dumps(d['config']['host_per_rank']))
print(json.dumps(d['config']['host_placement_per_rank']))
print(json.dumps(d.get('ingest')))
EOF

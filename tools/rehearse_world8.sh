# World-8 host rehearsal on ONE card (the only 8-rank evidence obtainable on a 1-GPU lease): 8 ranks over gloo share device 0.
# Records per-rank host CPU-seconds per step, thread counts, core placement and what the streamed rank-0 gather leaves for the end of
# the region; throughput at 8 ranks on one GPU says nothing about 8 GPUs and is not quoted.  -> profiles/r06_multirank_world8_gloo.json
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
TA_BENCH_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29518 \
  bench.py --gpus 8 --steps 3 --warmup 2 --batch 8 --inflight 2 --no-cpu-baseline --no-side-legs --sustain-seconds 0 --side-seconds 0.3 \
  --detail gpurun_out/r06/dist_world8_gloo_detail.json > gpurun_out/r06/dist_world8_gloo.json 2> gpurun_out/r06/dist_world8_gloo.err
echo "rc=$?"
tail -c 1200 gpurun_out/r06/dist_world8_gloo.err | grep -v '^{'
cat gpurun_out/r06/dist_world8_gloo.json; echo
python - <<'EOF'
import json
d = json.load(open('gpurun_out/r06/dist_world8_gloo_detail.json'))
print(json.dumps(d['config']['host_per_rank']))
print(json.dumps(d.get('ingest')))
EOF

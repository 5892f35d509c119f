mkdir -p gpurun_out/r06
export TMPDIR=/tmp
bash tools/rehearse_multirank.sh > gpurun_out/r06/rehearse_multirank.log 2>&1; tail -c 3600 gpurun_out/r06/rehearse_multirank.log
TA_BENCH_GATHER_STEPS=8 bash tools/rehearse_multirank.sh 2>&1 | tail -c 3400 > gpurun_out/r06/rehearse_multirank_g8.log; python - <<'EOF'
import json
for f in ('dist_world1_rccl','dist_world2_gloo'):
    d=json.load(open('gpurun_out/r06/%s.json'%f)); print('G=8', f, d['value'], d.get('value_ingest'), d.get('ingest'))
EOF

mkdir -p gpurun_out/r04
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -30
bash tools/rehearse_multirank.sh 2>&1 | grep -E "^rc|Traceback|Error" | head
python - <<'PY'
import json
for f in ('dist_world1_rccl','dist_world2_gloo'):
    try:
        d=json.load(open('gpurun_out/r04/%s.json'%f))
        print(f, d['value'], d['n_gpus'], d['timed_steps'], d['value_k_steps'], d.get('value_f16_embedder'), d.get('value_f32'), d.get('value_ingest'), d.get('c2_retinaface_640'), d['config']['host_placement_per_rank'])
    except Exception as e:
        print(f,'ERR',e); print(open('gpurun_out/r04/%s.err'%f).read()[-2000:])
PY

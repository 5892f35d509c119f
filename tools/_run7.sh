mkdir -p gpurun_out/r04
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -30
timeout 1500 python tests/probe_wild_weights.py > gpurun_out/r04/wild_probe4.txt 2> gpurun_out/r04/wild_probe4.err; echo probe rc=$?
tail -3 gpurun_out/r04/wild_probe4.err
grep -v "^  retinaface:\|^  arcface:\|^  openpose:" gpurun_out/r04/wild_probe4.txt
timeout 600 python bench.py --single-mode --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r04/hard_f16x3.json 2> gpurun_out/r04/hard_f16x3.err; echo rc=$?
python -c "
import json; d=json.load(open('gpurun_out/r04/hard_f16x3.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['stage_ms_per_step'])"

mkdir -p gpurun_out/r2h
timeout 600 python -m pytest tests/test_gpu_nets.py -m gpu -q -s -k retinaface > gpurun_out/r2h/nets.log 2>&1; tail -3 gpurun_out/r2h/nets.log
timeout 300 python tools/layer_profile.py f32 2> gpurun_out/r2h/layers_f32.txt
grep "model kind" gpurun_out/r2h/layers_f32.txt
awk '/==== retinaface/{c++} c==2' gpurun_out/r2h/layers_f32.txt | sed -n 2,15p | cut -c1-120

mkdir -p gpurun_out/r04
python tools/layer_profile.py f32 2> gpurun_out/r04/layers_f32_new.txt >/dev/null
(cd ab_old && python tools/layer_profile.py f32 2> ../gpurun_out/r04/layers_f32_old.txt >/dev/null)
python tools/layer_profile.py f16x3 2> gpurun_out/r04/layers_f16x3_new.txt >/dev/null
(cd ab_old && python tools/layer_profile.py f16x3 2> ../gpurun_out/r04/layers_f16x3_old.txt >/dev/null)
grep "model kind" gpurun_out/r04/layers_*.txt

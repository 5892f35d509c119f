mkdir -p gpurun_out/r04
python tools/layer_profile.py f16x3 2> gpurun_out/r04/layers_f16x3_new2.txt >/dev/null
python tools/layer_profile.py f32 2> gpurun_out/r04/layers_f32_new2.txt >/dev/null
python tools/layer_profile.py f16 2> gpurun_out/r04/layers_f16_new2.txt >/dev/null
grep "model kind" gpurun_out/r04/layers_*new2.txt
for tree in new old; do
  if [ $tree = old ]; then cd ab_old; fi
  echo "== $tree"; python tools/conv_bench.py f16x3 2>/dev/null | cut -c1-100
  if [ $tree = old ]; then cd ..; fi
done
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15

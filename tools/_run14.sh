mkdir -p gpurun_out/r04
for tree in new old; do
  if [ $tree = old ]; then cd ab_old; fi
  echo "== $tree"; python tools/conv_bench.py f16x3 2>/dev/null | cut -c1-100
  if [ $tree = old ]; then cd ..; fi
done
python tools/layer_profile.py f16x3 2> gpurun_out/r04/layers_f16x3_new3.txt >/dev/null
grep "model kind" gpurun_out/r04/layers_f16x3_new3.txt
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_conv_variants.py -q -m gpu 2>&1 | tail -2; done
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -25

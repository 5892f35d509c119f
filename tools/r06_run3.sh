mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_nets.py tests/test_gpu_conv.py -m gpu -x -q -k "retinaface or dwpw or detector" 2>&1 | tail -3
for BM in 128 64 32 0; do
  echo "==== TA_DWPW_BM=$BM" 
  TA_DWPW_BM=$BM timeout 300 python tools/detector_profile.py 32 640 640 f16x3 2>&1 | tail -32 | head -14
  TA_DWPW_BM=$BM timeout 300 python tools/detector_profile.py 32 640 640 f16x3 2>&1 | tail -1
  TA_DWPW_BM=$BM timeout 300 python tools/detector_profile.py 32 416 739 f16x3 2>&1 | tail -1
done > gpurun_out/r06/dwpw_bm2.txt 2>&1
cat gpurun_out/r06/dwpw_bm2.txt

mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_f16x3_range.py tests/test_gpu_fullsize.py -m gpu -x -q -k "retinaface or detector or dwpw or c5 or range or wild" 2>&1 | tail -8
for G in "" 1; do
  echo "==== TA_DWPW_GENERIC=$G"
  if [ -n "$G" ]; then export TA_DWPW_GENERIC=1; else unset TA_DWPW_GENERIC; fi
  timeout 300 python tools/detector_profile.py 32 640 640 f16x3 2>&1 | tail -30 | head -13
  timeout 300 python tools/detector_profile.py 32 640 640 f16x3 2>&1 | tail -1
  timeout 300 python tools/detector_profile.py 32 416 739 f16x3 2>&1 | tail -1
done 2>&1 | tee gpurun_out/r06/lean_dwpw.txt

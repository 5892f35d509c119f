mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_fullsize.py tests/test_gpu_decisions_vs_oracle.py -m gpu -x -q -k "retinaface or detector or c5 or c2 or facade" 2>&1 | tail -8
for NF in "" 1; do
  echo "==== TERRAN_AMD_NO_FUSED_FRONT=$NF"
  TERRAN_AMD_NO_FUSED_FRONT=$NF timeout 300 python tools/detector_profile.py 32 640 640 f16x3 2>&1 | tail -31 | head -4
  TERRAN_AMD_NO_FUSED_FRONT=$NF timeout 300 python tools/detector_profile.py 32 640 640 f16x3 2>&1 | tail -1
  TERRAN_AMD_NO_FUSED_FRONT=$NF timeout 300 python tools/detector_profile.py 32 416 739 f16x3 2>&1 | tail -1
  TERRAN_AMD_NO_FUSED_FRONT=$NF timeout 300 python tools/detector_profile.py 32 640 640 f32 2>&1 | tail -1
done 2>&1 | tee gpurun_out/r06/fused_front.txt

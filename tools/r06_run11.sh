mkdir -p gpurun_out/r06
export TMPDIR=/tmp
for PR in 0 1 2 4 8 3 15; do
  echo "==== TA_DWPW_PROBE=$PR (bit 0 no tap loads, 1 no stores, 2 weights once, 3 no MFMAs)"
  TA_DWPW_PROBE=$PR timeout 300 python tools/detector_profile.py 32 640 640 f16x3 2>&1 | tail -27 | head -9 | cut -c1-110
done 2>&1 | tee gpurun_out/r06/lean_dwpw_probe.txt

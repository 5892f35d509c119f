mkdir -p gpurun_out/dwprobe
for pr in 0 1 2; do
  TA_DWPW_PROBE=$pr timeout 300 python tools/detector_profile.py 32 640 640 f16x3 2>&1 | tail -32 > gpurun_out/dwprobe/c2_$pr.txt
done
paste <(awk '/^op/{print $2, $3, $4, $13, $14}' gpurun_out/dwprobe/c2_0.txt | head -13) <(awk '/^op/{print $13}' gpurun_out/dwprobe/c2_1.txt | head -13) <(awk '/^op/{print $13}' gpurun_out/dwprobe/c2_2.txt | head -13)

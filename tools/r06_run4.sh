export TMPDIR=/tmp
mkdir -p gpurun_out/r06
for BM in 128 64 32; do
  echo "==== TA_DWPW_BM=$BM: dw 128 -> pw 128 @ 32 x 40 x 40"
  TA_DWPW_BM=$BM timeout 120 python tools/dwpw_trace.py 128 128 32 40 40 2>&1 | tail -9
  echo "==== TA_DWPW_BM=$BM: dw 256 -> pw 256 @ 32 x 20 x 20"
  TA_DWPW_BM=$BM timeout 120 python tools/dwpw_trace.py 256 256 32 20 20 2>&1 | tail -9
done > gpurun_out/r06/dwpw_trace_bm.txt 2>&1
cat gpurun_out/r06/dwpw_trace_bm.txt

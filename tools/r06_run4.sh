export TMPDIR=/tmp
mkdir -p gpurun_out/r06
for PROBE in 0 2; do
for BM in 128 32; do
  echo "==== TA_DWPW_BM=$BM TA_DWPW_PROBE=$PROBE: dw 128 -> pw 128 @ 32 x 40 x 40"
  TA_DWPW_PROBE=$PROBE TA_DWPW_BM=$BM timeout 120 python tools/dwpw_trace.py 128 128 32 40 40 2>&1 | tail -16
done; done > gpurun_out/r06/dwpw_trace_bm2.txt 2>&1
cat gpurun_out/r06/dwpw_trace_bm2.txt

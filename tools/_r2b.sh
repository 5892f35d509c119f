mkdir -p gpurun_out/r2b
python tools/conv_bench.py bf16x3 > gpurun_out/r2b/base.log 2>&1
TA_CONV_PROBE=1 python tools/conv_bench.py bf16x3 > gpurun_out/r2b/probe1.log 2>&1
TA_CONV_PROBE=2 python tools/conv_bench.py bf16x3 > gpurun_out/r2b/probe2.log 2>&1
TA_CONV_PREFER=6 python tools/conv_bench.py bf16x3 > gpurun_out/r2b/p6.log 2>&1
TA_CONV_PREFER=6 TA_CONV_PROBE=1 python tools/conv_bench.py bf16x3 > gpurun_out/r2b/p6probe1.log 2>&1
TA_CONV_PREFER=6 TA_CONV_PROBE=2 python tools/conv_bench.py bf16x3 > gpurun_out/r2b/p6probe2.log 2>&1
TA_CONV_PROBE=1 python tools/conv_bench.py f32 > gpurun_out/r2b/f32probe1.log 2>&1
python tools/conv_bench.py f32 > gpurun_out/r2b/f32base.log 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/r2b/pytest.log 2>&1
for f in gpurun_out/r2b/*.log; do echo "== $f"; tail -n 12 $f; done

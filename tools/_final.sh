set -x
mkdir -p gpurun_out/final
ls /sys/class/drm/ > gpurun_out/final/drm.txt 2>&1
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/final/pytest_gpu.txt 2>&1
tail -15 gpurun_out/final/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/final/smoke.txt 2>&1
tail -2 gpurun_out/final/smoke.txt
python tools/power_log.py --out gpurun_out/final/power_bench.txt -- timeout 900 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -c 600 gpurun_out/final/bench.err
python tools/power_log.py --out gpurun_out/final/power_f32.txt -- timeout 600 python bench.py --single-mode --precision f32 > gpurun_out/final/bench_f32.json 2>> gpurun_out/final/bench.err
timeout 60 rocm-smi --showpower --showclocks --showmaxpower > gpurun_out/final/rocm_smi.txt 2>&1

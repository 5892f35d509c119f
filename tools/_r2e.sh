mkdir -p gpurun_out/r2e
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py -m gpu -q -x -k "c5_fullsize or fanout or registry or stage_taps or adversarial" > gpurun_out/r2e/tests.log 2>&1; tail -5 gpurun_out/r2e/tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2e/bench20.json 2> gpurun_out/r2e/bench20.err; tail -c 600 gpurun_out/r2e/bench20.err; cut -c1-400 gpurun_out/r2e/bench20.json
python bench.py --gpus 2 --single-process --devices 0,0 --steps 10 --warmup 2 > gpurun_out/r2e/sp.json 2> gpurun_out/r2e/sp.err; tail -c 300 gpurun_out/r2e/sp.err; cat gpurun_out/r2e/sp.json
TA_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2e/mp2.json 2> gpurun_out/r2e/mp2.err; tail -c 400 gpurun_out/r2e/mp2.err; cut -c1-300 gpurun_out/r2e/mp2.json

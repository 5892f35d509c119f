mkdir -p gpurun_out/r05
export TMPDIR=/tmp
( TA_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r05/dist_world1_rccl.json 2> gpurun_out/r05/dist_world1_rccl.err; echo "rc1=$?" )
( TA_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/r05/dist_world2_gloo.json 2> gpurun_out/r05/dist_world2_gloo.err; echo "rc2=$?" )
tail -c 600 gpurun_out/r05/dist_world1_rccl.err; tail -c 1500 gpurun_out/r05/dist_world2_gloo.err
head -c 400 gpurun_out/r05/dist_world1_rccl.json; echo; head -c 400 gpurun_out/r05/dist_world2_gloo.json
rocm-smi --showtoponuma 2>/dev/null | head -20; nproc; ls /sys/class/drm/ | head; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head

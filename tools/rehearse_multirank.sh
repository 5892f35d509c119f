# World-1 over RCCL (the real backend's calls at N = 1) and world-2 over gloo (two ranks sharing the one card): launch line, barrier,
# max-reduce, the streamed ordered gather.  -> profiles/r06_multirank_world1_rccl.json, r06_multirank_world2_gloo.json
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
( TA_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-side-legs --detail gpurun_out/r06/dist_world1_rccl_detail.json > gpurun_out/r06/dist_world1_rccl.json 2> gpurun_out/r06/dist_world1_rccl.err; echo "rc1=$?" )
( TA_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-side-legs --detail gpurun_out/r06/dist_world2_gloo_detail.json > gpurun_out/r06/dist_world2_gloo.json 2> gpurun_out/r06/dist_world2_gloo.err; echo "rc2=$?" )
tail -c 400 gpurun_out/r06/dist_world1_rccl.err | grep -v '^{' ; tail -c 600 gpurun_out/r06/dist_world2_gloo.err | grep -v '^{'
cat gpurun_out/r06/dist_world1_rccl.json; echo; cat gpurun_out/r06/dist_world2_gloo.json

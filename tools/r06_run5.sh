mkdir -p gpurun_out/r06
export TMPDIR=/tmp
bash tools/r06_energy_ab.sh > gpurun_out/r06/energy_ab.txt 2>&1
cat gpurun_out/r06/energy_ab.txt
bash tools/rehearse_multirank.sh > gpurun_out/r06/rehearse_multirank.log 2>&1; tail -c 1500 gpurun_out/r06/rehearse_multirank.log
bash tools/rehearse_world8.sh > gpurun_out/r06/rehearse_world8.log 2>&1; tail -c 2500 gpurun_out/r06/rehearse_world8.log

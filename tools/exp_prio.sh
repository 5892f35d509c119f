# A/B of the split-role kernels' consumer-wave issue priority (TA_CONV_PRIO = 0 / 1 / 2 / 3): kernels alone (tools/conv_bench.py) and the step
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
for P in 0 1 3; do
  echo "==== TA_CONV_PRIO=$P"
  TA_CONV_PRIO=$P timeout 300 python tools/conv_bench.py f16x3 2>&1 | grep -v wino | cut -c1-120
done
echo "## step level"
bash tools/exp_ab.sh "TA_CONV_PRIO=0" "TA_CONV_PRIO=1" 3

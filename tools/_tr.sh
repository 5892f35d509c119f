cd /root/repo
for i in 1 2 3; do
for V in "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"; do
  env $V timeout 300 python bench.py --single-mode --no-cpu-baseline --steps 100 --warmup 6 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$V',d['value'],d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"
done; done

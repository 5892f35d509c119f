#!/bin/bash
# A/B: plain per-image result lists (the default) against lazy ones (TERRAN_AMD_LAZY_RESULTS=1): pipelined headline + C2 wrapper rate
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do
  for l in "" 1; do
    TERRAN_AMD_LAZY_RESULTS=$l python bench.py --single-mode --no-cpu-baseline --steps 120 --warmup 8 --detail /tmp/ab_detail.json 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('lazy=$l', d['value'], d['value_k_steps'], d['config'].get('host_cpu_s_per_step'))"
  done
done
for l in "" 1; do echo "lazy=$l"; TERRAN_AMD_LAZY_RESULTS=$l python tools/model_bench.py --precisions f16x3 2>/dev/null | grep "C2 Retina"; done

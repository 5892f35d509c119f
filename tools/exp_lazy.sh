#!/bin/bash
# A/B: lazy per-image result lists (default) against eager ones (TERRAN_AMD_EAGER_RESULTS=1): pipelined headline + C2 wrapper rate
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do
  for e in "" 1; do
    TERRAN_AMD_EAGER_RESULTS=$e python bench.py --single-mode --no-cpu-baseline --steps 120 --warmup 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('eager=$e', d['value'], d['value_k_steps'], d['config'].get('host_per_rank'))"
  done
done
for e in "" 1; do echo "eager=$e"; TERRAN_AMD_EAGER_RESULTS=$e python tools/model_bench.py --precisions f16x2 2>/dev/null | grep "C2 Retina"; done

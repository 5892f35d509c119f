#!/bin/bash
# interleaved A/B of bench.py headline settings: tools/exp_ab.sh "<args A>" "<args B>" [rounds]
cd ${GRAFT_REPO_ROOT:-.}
R=${3:-3}
for i in $(seq $R); do
  for cfg in "$1" "$2"; do
    python bench.py --single-mode --no-cpu-baseline --steps 120 --warmup 8 $cfg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$cfg', '|', d['value'], d['value_k_steps'], d['power'] and (d['power'].get('sclk_mhz_mean'), d['power'].get('power_w_mean')))"
  done
done

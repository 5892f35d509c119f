#!/bin/bash
# embed batch-size sweep (pipelined headline, single mode)
cd $GRAFT_REPO_ROOT
for cfg in "256 512" "320 320" "192 192" "128 128" "320 383"; do
  set -- $cfg
  echo "== min $1 max $2"
  python bench.py --single-mode --full-line --no-cpu-baseline --steps 120 --warmup 8 --detail /tmp/ab_detail.json --embed-min-crops $1 --embed-max-crops $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['value_k_steps'], d['config']['step_overlap'][-60:], d['power'] and d['power'].get('sclk_mhz_mean'))"
done

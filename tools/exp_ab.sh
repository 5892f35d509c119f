#!/bin/bash
# interleaved A/B of bench.py headline settings at STEP level, with clock, power, joules per frame and PPT-throttle residency:
#   tools/exp_ab.sh "<args or ENV=.. assignments A>" "<... B>" [rounds] [extra args for both]
# a leading run of NAME=value words in a configuration is exported to that run's environment, the rest goes to bench.py
cd ${GRAFT_REPO_ROOT:-.}
R=${3:-3}
for i in $(seq $R); do
  for cfg in "$1" "$2"; do
    envs=(); args=()
    for w in $cfg; do
      if [[ ${#args[@]} -eq 0 && "$w" == *=* && "$w" != --* ]]; then envs+=("$w"); else args+=("$w"); fi
    done
    env "${envs[@]}" python bench.py --single-mode --no-cpu-baseline --steps 120 --warmup 8 --detail /tmp/ab_detail.json ${4:-} "${args[@]}" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
p=d.get('power') or {}
print('%-44s | %8.1f frames/s  k-steps %8.1f | %s W %s MHz | %s J/frame | ppt %s %% | host %s cpu-s/step' % ('$cfg', d['value'], d['value_k_steps'], p.get('power_w_mean'), p.get('sclk_mhz_mean'), p.get('energy_j_per_frame'), (p.get('throttle') or {}).get('ppt'), d['config'].get('host_cpu_s_per_step')))"
  done
done

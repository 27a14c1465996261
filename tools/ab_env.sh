#!/bin/bash
# same-box A/B of environment settings on the question stream: tools/ab_env.sh [-r REPS] [-s STEPS] "" "ZE_HOLD=256" "ZE_HOLD=512 ZE_BURST=4" ...
reps=2; steps=20
while [ "${1:0:1}" = "-" ]; do
  case "$1" in -r) reps=$2; shift 2;; -s) steps=$2; shift 2;; *) break;; esac
done
for rep in $(seq $reps); do
for t in "$@"; do
  env $t python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-batch64 --no-configs1 --no-reuse-sensitivity 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['scheduler']
        h = {k[9:]: v for k, v in sorted(s.items(), key=lambda kv: (len(kv[0]), kv[0])) if k.startswith('steps_')}
        print('[$t] value', round(d['value'], 2), 'steps', s['steps'], 'mean chains', round(d.get('mean_chains_per_step', 0), 1), 'held', s.get('held_steps', 0), h)
"
done; done

#!/bin/bash
# same-box A/B of ze_tune settings on the question stream: tools/ab_tune.sh [-r REPS] "" "22:1" "20:1,13:4" ...
# prints the line's value, the decode layer time and the per-phase milliseconds per question of every run
reps=2
if [ "$1" = "-r" ]; then reps=$2; shift 2; fi
for rep in $(seq $reps); do
for t in "$@"; do
  ZE_TUNE="$t" python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-batch64 --no-configs1 --no-reuse-sensitivity 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        ph = {k: round(v['ms_per_question'], 3) for k, v in d.get('roofline_phases', {}).items() if isinstance(v, dict) and 'ms_per_question' in v}
        print('tune[$t] value', round(d['value'], 2), 'attention_us', d['roofline'].get('avg_us'), 'independent', d['roofline'].get('independent_chains', {}).get('avg_us'), 'layer_us', d['roofline'].get('layer_us'), 'phases', ph)
"
done; done

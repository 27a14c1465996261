for rep in 1 2; do
for t in "" "20:1" "20:1,13:4,18:1,15:8"; do
  ZE_TUNE="$t" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-batch64 --no-configs1 --no-reuse-sensitivity 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('tune[$t] value', round(d['value'],2), 'layer_us', d['roofline'].get('layer_us'), 'decode frac', d.get('roofline_phases',{}).get('decode',{}).get('frac'))
"
done; done

#!/bin/bash
# Same-box A/B of the K / V LDS swizzle (ze_kv_swz, ze_kernels.h): the library built with -DZE_KV_SWZ_OLD for the three attention units
# (zoomearth_amd/libzoomearth_hip_oldswz.so, see DESIGN.md 7h for the build line) against the shipped one, on the question stream.
# Prints value, the decode attention alone (shared / independent chains) and the isolated ViT / prefill milliseconds per question.
cd "$(dirname "$0")/../.."
cp zoomearth_amd/libzoomearth_hip.so /tmp/swz_new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp zoomearth_amd/libzoomearth_hip_oldswz.so zoomearth_amd/libzoomearth_hip.so; else cp /tmp/swz_new.so zoomearth_amd/libzoomearth_hip.so; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch64 --no-configs1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; ph = d['roofline_phases']
        print('$v value', round(d['value'], 2), '| attention us', r['avg_us'], 'independent', r['independent_chains']['avg_us'], 'layer', r['layer_us'],
              '| isolated ms/q vit', ph['vit']['ms_per_question'], 'prefill', ph['prefill']['ms_per_question'], 'decode', ph['decode']['ms_per_question'])
"
  done
done
cp /tmp/swz_new.so zoomearth_amd/libzoomearth_hip.so

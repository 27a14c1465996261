#!/bin/bash
# Same-box A/B of two builds of the library under the driver's command (the finds of DESIGN.md 7h were each measured this way).
# Build the variant first, e.g. the integer bf16 conversion / the divided SiLU / the old LDS swizzle:
#   cp -r zoomearth_amd/csrc include /tmp/v/ (keeping the relative layout) && make -C /tmp/v/zoomearth_amd/csrc EXTRA="-DZE_SOFT_BF16"
#   (or -DZE_EXACT_SILU_DIV, -DZE_KV_SWZ_OLD, -DZE_FA_NO_FAST_PATH) && cp /tmp/v/zoomearth_amd/libzoomearth_hip.so zoomearth_amd/libzoomearth_hip_oldswz.so
# then run this script on the GPU box: it swaps the two files under bench.py, two repetitions each, and prints value, the decode
# attention alone (shared / independent chains), the decode layer and the isolated ViT / prefill / decode milliseconds per question.
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
        print('$v value', round(d['value'], 2), '| attention us', r['avg_us'], 'independent', r.get('independent_chains', {}).get('avg_us'), 'layer', r['layer_us'],
              '| isolated ms/q vit', ph['vit']['ms_per_question'], 'prefill', ph['prefill']['ms_per_question'], 'decode', ph['decode']['ms_per_question'])
"
  done
done
cp /tmp/swz_new.so zoomearth_amd/libzoomearth_hip.so

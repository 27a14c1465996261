#!/bin/bash
# round 6, eighth GPU call: did the attention unit's new (disabled) code cost the shipped form anything?  Same box, alternating: the
# tree before the change (_oldtree, commit 0e33787) against this one, tools/pmc_kernel.py wide490_shared / wide490 (HIP-event time)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for rep in 1 2 3; do
  for t in old new; do
    if [ $t = old ]; then d=_oldtree; else d=.; fi
    for m in wide490_shared wide490; do
      ( cd $d && timeout 300 python tools/pmc_kernel.py $m 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$t $m attention_us', d['attention']['us'])
" )
    done
  done
done | tee gpurun_out/r6/ab_attention_old_new.txt

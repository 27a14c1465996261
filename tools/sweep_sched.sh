#!/bin/bash
# scheduler-parameter sweep of the stream workload: tools/sweep_sched.sh "BURST MIN_ADMIT MAX_WAIT PREFILL_ROWS" ...
mkdir -p gpurun_out/sweep
for cfg in "$@"; do
  set -- $cfg
  tag=$(echo $cfg | tr ' ' '_')
  ZE_BURST=$1 ZE_MIN_ADMIT=$2 ZE_MAX_WAIT=$3 ZE_PREFILL_ROWS=$4 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch64 --no-configs1 > gpurun_out/sweep/$tag.json 2> gpurun_out/sweep/$tag.err
  python tools/show_line.py gpurun_out/sweep/$tag.json "burst/min_admit/max_wait/rows $cfg:" || tail -3 gpurun_out/sweep/$tag.err
done

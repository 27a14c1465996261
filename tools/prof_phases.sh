#!/bin/bash
# Per-kernel shares of an isolated ViT call and of isolated prefill passes (tools/prof_phases.py) under rocprofv3.
# usage (through gpurun, from the repo root): tools/prof_phases.sh <tag>  -> gpurun_out/<tag>_{vit,prefill}_kernel_stats.csv
set -u
tag=${1:-r04}
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for w in vit prefill; do
  rm -rf /tmp/pp_$w
  ( cd "$root" && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$w -- python3 tools/prof_phases.py $w 6 > "$out/${tag}_${w}_phase.log" 2>&1 )
  # (the prefill replay: one number per kernel AND pass size -- VERDICT r5 #2 -- from the raw trace, before it is deleted)
  [ $w = prefill ] && python3 "$root/tools/prefill_by_shape.py" /tmp/pp_$w "$out/${tag}_prefill_by_shape.csv"
  python3 "$root/tools/summarize_prof.py" /tmp/pp_$w "$out/${tag}_${w}_kernel_stats.csv" --delete-raw
  tail -1 "$out/${tag}_${w}_phase.log"
  python3 - "$out/${tag}_${w}_kernel_stats.csv" <<'PY'
import csv, sys
rows = [r for r in csv.reader(open(sys.argv[1])) if r and r[0] == "kernel_stats"]
for r in sorted(rows, key=lambda r: -float(r[7]))[:18]:
    print(f"{float(r[7]):5.1f}% {float(r[4]) / 1000:9.1f}us x{r[3]:>5}  {r[1][:120]}")
PY
done

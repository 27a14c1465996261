#!/bin/bash
# kernel statistics of the stream (the driver's command minus the annexes) under rocprofv3: tools/profile_stream.sh [tag]
set -u
tag=${1:-r05}
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/p_stream
( cd "$root" && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stream -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-batch64 --no-configs1 > "$out/${tag}_stream_line.json" 2> "$out/${tag}_stream.log" )
echo "exit $?"
python3 "$root/tools/summarize_prof.py" /tmp/p_stream "$out/${tag}_stream_kernel_stats.csv" --delete-raw

#!/bin/bash
# Round 5, second pass (the scheduler's hold raised the stream's mean live-chain count from ~400 to ~490): the decode step's kernels
# ALONE at 490 chains (kernel trace, then FETCH_SIZE / WRITE_SIZE in separate passes) and the kernel statistics of the stream.
# usage: tools/profile_round5b.sh [tag]  -> gpurun_out/<tag>_*.csv|json   (the 410- / 580-chain files of tools/profile_round5.sh stay)
set -u
tag=${1:-r05}
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for m in wide490 wide490_shared; do
  rm -rf /tmp/p_kt
  ( cd "$root" && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -- python3 tools/pmc_kernel.py $m > "$out/${tag}_${m}_launches.json" 2> "$out/${tag}_${m}.log" )
  python3 "$root/tools/summarize_prof.py" /tmp/p_kt "$out/${tag}_${m}_kernel_trace.csv" --delete-raw
  for c in FETCH_SIZE:fetch_size WRITE_SIZE:write_size; do
    rm -rf /tmp/p_pmc
    ( cd "$root" && rocprofv3 --pmc ${c%%:*} --kernel-trace --output-format csv -d /tmp/p_pmc -- python3 tools/pmc_kernel.py $m > "$out/${tag}_pmc_${m}_launches.json" 2> "$out/${tag}_pmc_${m}_${c##*:}.log" )
    python3 "$root/tools/summarize_prof.py" /tmp/p_pmc "$out/${tag}_pmc_${m}_${c##*:}.csv" --delete-raw
  done
done
rm -rf /tmp/p_stream
( cd "$root" && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stream -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-batch64 --no-configs1 > "$out/${tag}_stream_line.json" 2> "$out/${tag}_stream.log" )
python3 "$root/tools/summarize_prof.py" /tmp/p_stream "$out/${tag}_stream_kernel_stats.csv" --delete-raw
ls -la "$out" | grep "${tag}_" | tail -20

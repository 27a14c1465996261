#!/bin/bash
# Regenerates the rocprofv3 summaries of profiles/ on the GPU box (run from the repo root through gpurun).
# usage: tools/profile_round.sh <tag>     -> gpurun_out/<tag>_*.csv|json
set -u
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write
( cd "$root" && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch64 > "$out/${tag}_bench_line_under_rocprof.json" 2> "$out/${tag}_stats.log" )
python3 "$root/tools/summarize_prof.py" /tmp/p_stats "$out/${tag}_bench_kernel_stats.csv" --delete-raw
( cd "$root" && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch64 --no-graph > /dev/null 2> "$out/${tag}_fetch.log" )
python3 "$root/tools/summarize_prof.py" /tmp/p_fetch "$out/${tag}_pmc_fetch_size.csv" --delete-raw
( cd "$root" && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch64 --no-graph > /dev/null 2> "$out/${tag}_write.log" )
python3 "$root/tools/summarize_prof.py" /tmp/p_write "$out/${tag}_pmc_write_size.csv" --delete-raw
rm -rf /tmp/p_mfma
( cd "$root" && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_mfma -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch64 --no-graph > /dev/null 2> "$out/${tag}_mfma.log" )
python3 "$root/tools/summarize_prof.py" /tmp/p_mfma "$out/${tag}_pmc_mfma_busy.csv" --delete-raw
# BASELINE configs[2]: 64 chains through the scheduler (kernel stats, then HBM fetch bytes per kernel)
rm -rf /tmp/p_b64 /tmp/p_b64f
( cd "$root" && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_b64 -- python3 bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline > "$out/${tag}_batch64_line.json" 2> "$out/${tag}_b64.log" )
python3 "$root/tools/summarize_prof.py" /tmp/p_b64 "$out/${tag}_batch64_kernel_stats.csv" --delete-raw
( cd "$root" && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_b64f -- python3 bench.py --batch 64 --steps 1 --warmup 0 --no-cpu-baseline --no-graph > /dev/null 2> "$out/${tag}_b64_fetch.log" )
python3 "$root/tools/summarize_prof.py" /tmp/p_b64f "$out/${tag}_batch64_pmc_fetch_size.csv" --delete-raw
ls -la "$out" | tail -12

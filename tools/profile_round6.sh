#!/bin/bash
# Round-6 rocprofv3 evidence (run from the repo root through gpurun): kernel statistics of the benchmark workloads, the isolated
# ViT call / prefill passes, the kernel trace + PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, --kernel-trace only beside
# them) of the decode step's kernels at the line's own chain count (490 since the hold; 410 before, live-like contexts) and at 580 chains, and the MFMA-busy table.
# usage: tools/profile_round6.sh <tag> [fast]  -> gpurun_out/<tag>_*.csv|json
set -u
tag=${1:-r06}
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
stats() {  # name, bench args...
  local name=$1; shift
  rm -rf /tmp/p_$name
  ( cd "$root" && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- python3 bench.py "$@" > "$out/${tag}_${name}_line.json" 2> "$out/${tag}_${name}.log" )
  python3 "$root/tools/summarize_prof.py" /tmp/p_$name "$out/${tag}_${name}_kernel_stats.csv" --delete-raw
}
pmc() {  # counter list, file tag, mode
  local ctr=$1 name=$2 mode=$3
  rm -rf /tmp/p_pmc
  ( cd "$root" && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/p_pmc -- python3 tools/pmc_kernel.py $mode > "$out/${tag}_pmc_${mode}_launches.json" 2> "$out/${tag}_pmc_${mode}_${name}.log" )
  python3 "$root/tools/summarize_prof.py" /tmp/p_pmc "$out/${tag}_pmc_${mode}_${name}.csv" --delete-raw
}
# the isolated decode step at the line's mean chain count (490) and in the bucket the stream runs most chain-steps in (580):
# kernel trace only (durations reproducible from profiles/), then the PMC passes
for n in 490 580; do
  for m in wide$n wide${n}_shared; do
    rm -rf /tmp/p_kt
    ( cd "$root" && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -- python3 tools/pmc_kernel.py $m > "$out/${tag}_${m}_launches.json" 2> "$out/${tag}_${m}.log" )
    python3 "$root/tools/summarize_prof.py" /tmp/p_kt "$out/${tag}_${m}_kernel_trace.csv" --delete-raw
  done
done
for mode in wide490 wide490_shared wide580 wide580_shared batch64 configs1; do
  pmc FETCH_SIZE fetch_size $mode
  pmc WRITE_SIZE write_size $mode
done
# the default line's workload (BASELINE configs[3], the question stream), configs[1] and configs[2]
stats stream --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-batch64 --no-configs1 --no-reuse-sensitivity   # (the driver's command, minus the annexes)
if [ "${2:-}" != "fast" ]; then
  stats configs1 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline
  stats batch64 --batch 64 --steps 1 --warmup 1 --no-cpu-baseline
fi
# MFMA utilisation of the stream's kernels (eager launches: counters are per dispatch)
rm -rf /tmp/p_mfma
( cd "$root" && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_mfma -- python3 bench.py --steps 2 --warmup 0 --lanes 1 --no-cpu-baseline --no-batch64 --no-configs1 --no-reuse-sensitivity --no-graph > /dev/null 2> "$out/${tag}_mfma.log" )
python3 "$root/tools/summarize_prof.py" /tmp/p_mfma "$out/${tag}_stream_pmc_mfma_busy.csv" --delete-raw
# per-kernel shares of an isolated ViT call and of isolated prefill passes
cd "$root" && bash tools/prof_phases.sh "$tag" > "$out/${tag}_phases.txt" 2>&1
ls -la "$out" | grep "${tag}_" | tail -40

#!/bin/bash
# round 6, third GPU call: the prefill split-K (three K slices for the down projection) -- the whole GPU suite on it, the per-shape
# table of the replayed passes, and the same-box A/B against knob 20 = 1 (unsplit) on the question stream
mkdir -p gpurun_out/r6
cd $GRAFT_REPO_ROOT
export ZE_PARITY_LEDGER=gpurun_out/r6/parity_ledger_splitk.json
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6/gpu_suite_splitk.txt 2>&1; echo "suite rc $?" >> gpurun_out/r6/gpu_suite_splitk.txt
tail -6 gpurun_out/r6/gpu_suite_splitk.txt
unset ZE_PARITY_LEDGER
timeout 600 bash tools/prof_phases.sh r06b > gpurun_out/r6/prof_phases_splitk.txt 2>&1
grep -E "^(down|gate_up|o |qkv)" gpurun_out/r6/prof_phases_splitk.txt
ZE_TUNE=20:1 timeout 300 python tools/prof_phases.py prefill 4 2>&1 | tail -1
timeout 300 python tools/prof_phases.py prefill 4 2>&1 | tail -1
timeout 1500 bash tools/ab_tune.sh -r 2 "" "20:1" > gpurun_out/r6/ab_splitk.txt 2>&1
cat gpurun_out/r6/ab_splitk.txt

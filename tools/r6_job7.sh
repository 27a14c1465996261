#!/bin/bash
# round 6, seventh GPU call: the final tree -- the whole GPU suite with the parity ledger, the tokenizer counters of the e2e tests,
# the round's rocprofv3 profiles (counter passes on THIS tree's kernel sources)
mkdir -p gpurun_out/r6
cd $GRAFT_REPO_ROOT
export ZE_PARITY_LEDGER=gpurun_out/r6/parity_ledger_final2.json
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6/gpu_suite_final2.txt 2>&1; echo "suite rc $?" >> gpurun_out/r6/gpu_suite_final2.txt
tail -4 gpurun_out/r6/gpu_suite_final2.txt
unset ZE_PARITY_LEDGER
timeout 600 python -m pytest tests/test_gpu_infer_e2e.py -m gpu -q -s -k "batched_equals" 2>&1 | grep -E "generated rows kept|passed|failed" 
timeout 2400 bash tools/profile_round6.sh r06 fast > gpurun_out/r6/profile_round6b.txt 2>&1; tail -3 gpurun_out/r6/profile_round6b.txt

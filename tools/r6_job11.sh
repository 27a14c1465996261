#!/bin/bash
# round 6, last GPU call: the whole GPU suite on the final tree (as the driver runs it), with the parity ledger
mkdir -p gpurun_out/r6
cd $GRAFT_REPO_ROOT
export ZE_PARITY_LEDGER=gpurun_out/r6/parity_ledger_final3.json
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6/gpu_suite_final3.txt 2>&1; echo "suite rc $?" >> gpurun_out/r6/gpu_suite_final3.txt
tail -4 gpurun_out/r6/gpu_suite_final3.txt

#!/bin/bash
# round 6, second GPU call: the whole GPU suite with the parity ledger, the ledger of the exact-division SiLU build on the 3B-shape /
# full-depth tests, the driver-form line with the new annexes, the per-shape prefill table
mkdir -p gpurun_out/r6
cd $GRAFT_REPO_ROOT
export ZE_PARITY_LEDGER=gpurun_out/r6/parity_ledger.json
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6/gpu_suite.txt 2>&1; echo "suite rc $?" >> gpurun_out/r6/gpu_suite.txt
tail -4 gpurun_out/r6/gpu_suite.txt
ZE_LIB_PATH=$PWD/zoomearth_amd/libzoomearth_hip_exactsilu.so ZE_PARITY_LEDGER=gpurun_out/r6/parity_ledger_exactsilu.json timeout 900 python -m pytest tests/test_gpu_3b_shape.py tests/test_gpu_smoke.py tests/test_gpu_full_depth.py tests/test_gpu_ops_kernels.py tests/test_gpu_batch.py -m gpu -q -s > gpurun_out/r6/gpu_exactsilu.txt 2>&1
tail -2 gpurun_out/r6/gpu_exactsilu.txt
unset ZE_PARITY_LEDGER
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/line_a.json 2> gpurun_out/r6/line_a.err
python tools/show_line.py gpurun_out/r6/line_a.json 2>/dev/null | head -40
timeout 600 bash tools/prof_phases.sh r06 > gpurun_out/r6/prof_phases.txt 2>&1
tail -25 gpurun_out/r6/prof_phases.txt

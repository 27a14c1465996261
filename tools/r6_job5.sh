#!/bin/bash
# round 6, fifth GPU call: the one-lane form of VERDICT r5 #1b on its tuned tiles (pair of launches for qkv, 256 x 256 split-K tiles for down beyond 768 rows)
mkdir -p gpurun_out/r6
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_ops.py -m gpu -q -k "above_768 or (wide_decode and 1152) or split_k" > gpurun_out/r6/tests_job5.txt 2>&1; tail -3 gpurun_out/r6/tests_job5.txt
( ZE_COUNTS=978,1152,1280,1408 timeout 900 python tools/bench_wide.py 1408 "" > gpurun_out/r6/wide_1408_tuned2.txt 2>&1 ); tail -5 gpurun_out/r6/wide_1408_tuned2.txt
( timeout 1500 bash tools/ab_env.sh -r 2 "" "ZE_LANES=1 ZE_STREAM_SLOTS=1536" "ZE_LANES=1 ZE_STREAM_SLOTS=1536 ZE_HOLD=768" > gpurun_out/r6/ab_lanes_tuned2.txt 2>&1 ); cat gpurun_out/r6/ab_lanes_tuned2.txt

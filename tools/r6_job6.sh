#!/bin/bash
# round 6, sixth GPU call: the decode attention with split rows and paired prefix parts -- its tests, the attention tests around it, the
# launch alone (shared / independent, paired / unpaired / round-5 partition), the stream A/B
mkdir -p gpurun_out/r6
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_ops_kernels.py tests/test_gpu_3b_shape.py tests/test_gpu_model.py -m gpu -q -x -s > gpurun_out/r6/tests_job6.txt 2>&1; tail -4 gpurun_out/r6/tests_job6.txt; grep -E "split|partition" gpurun_out/r6/tests_job6.txt | head
( ZE_GROUP=10 ZE_COUNTS=490,576 timeout 900 python tools/bench_wide.py 768 "" "23:2" "23:1" > gpurun_out/r6/wide_attn_split.txt 2>&1 ); grep -E "n=490|n=576" gpurun_out/r6/wide_attn_split.txt | sed 's/|.*attention/| attention/'
( timeout 1500 bash tools/ab_tune.sh -r 2 "" "23:2" "23:1" > gpurun_out/r6/ab_attn_split.txt 2>&1 ); cat gpurun_out/r6/ab_attn_split.txt

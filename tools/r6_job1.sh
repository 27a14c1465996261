#!/bin/bash
# round 6, first GPU call: the A/B data of VERDICT r5 #1 on the code as it stands
mkdir -p gpurun_out/r6
cd $GRAFT_REPO_ROOT
( timeout 600 python tools/price_mixed_pass.py > gpurun_out/r6/mixed_pricing.txt 2>&1; cp gpurun_out/mixed_pass_pricing.json gpurun_out/r6/ )
( ZE_COUNTS=489,576,768,978,1152,1408 timeout 900 python tools/bench_wide.py 1408 "" > gpurun_out/r6/wide_1408.txt 2>&1 )
( timeout 1500 bash tools/ab_env.sh -r 2 "" "ZE_LANES=1 ZE_STREAM_SLOTS=1536" > gpurun_out/r6/ab_lanes.txt 2>&1 )
tail -3 gpurun_out/r6/mixed_pricing.txt gpurun_out/r6/wide_1408.txt gpurun_out/r6/ab_lanes.txt

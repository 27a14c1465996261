#!/bin/bash
# round 6, fourth GPU call: the final state -- the whole GPU suite with the parity ledger, the one-lane A/B of VERDICT r5 #1b on the
# tuned > 768-row tiles, the driver-form line, the real entry point, the round's rocprofv3 profiles
mkdir -p gpurun_out/r6
cd $GRAFT_REPO_ROOT
export ZE_PARITY_LEDGER=gpurun_out/r6/parity_ledger_final.json
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6/gpu_suite_final.txt 2>&1; echo "suite rc $?" >> gpurun_out/r6/gpu_suite_final.txt
tail -4 gpurun_out/r6/gpu_suite_final.txt
unset ZE_PARITY_LEDGER
( ZE_COUNTS=489,576,768,978,1152,1408 timeout 900 python tools/bench_wide.py 1408 "" > gpurun_out/r6/wide_1408_tuned.txt 2>&1 ); tail -7 gpurun_out/r6/wide_1408_tuned.txt
( timeout 1200 bash tools/ab_env.sh -r 2 "" "ZE_LANES=1 ZE_STREAM_SLOTS=1536" > gpurun_out/r6/ab_lanes_tuned.txt 2>&1 ); cat gpurun_out/r6/ab_lanes_tuned.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/line_final.json 2> gpurun_out/r6/line_final.err
python tools/show_line.py gpurun_out/r6/line_final.json
timeout 900 python tools/bench_infer_e2e.py --questions 1024 --batch_size 512 --max_new_tokens 144 --lanes 2 --hold 384 > gpurun_out/r6/infer_e2e.json 2> gpurun_out/r6/infer_e2e.err; tail -c 600 gpurun_out/r6/infer_e2e.json
timeout 2400 bash tools/profile_round6.sh r06 fast > gpurun_out/r6/profile_round6.txt 2>&1; tail -5 gpurun_out/r6/profile_round6.txt

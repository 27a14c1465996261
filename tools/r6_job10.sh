#!/bin/bash
# round 6, tenth GPU call: the driver-form line and the real entry point on the final tree (profiles committed: traffic_source says "the same")
mkdir -p gpurun_out/r6
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/line_final2.json 2> gpurun_out/r6/line_final2.err
python tools/show_line.py gpurun_out/r6/line_final2.json
timeout 900 python tools/bench_infer_e2e.py --questions 1024 --batch_size 512 --max_new_tokens 144 --lanes 2 --hold 384 > gpurun_out/r6/infer_e2e2.json 2> gpurun_out/r6/infer_e2e2.err; tail -c 300 gpurun_out/r6/infer_e2e2.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()"
timeout 600 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q 2>&1 | tail -3

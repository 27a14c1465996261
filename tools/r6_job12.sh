#!/bin/bash
# round 6, the final measurement chain in ONE call: quick tests of the last host-side change, the round's profiles on the final kernel
# sources, the digest (so that the line's traffic_source reads "the same"), the driver-form line, the real entry point
mkdir -p gpurun_out/r6
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_smoke.py -m gpu -q -x > gpurun_out/r6/tests_job12.txt 2>&1; tail -2 gpurun_out/r6/tests_job12.txt
timeout 2400 bash tools/profile_round6.sh r06 fast > gpurun_out/r6/profile_round6d.txt 2>&1; tail -2 gpurun_out/r6/profile_round6d.txt
for f in gpurun_out/r06_*; do case "$f" in *.log) ;; *) cp "$f" profiles/ ;; esac; done
python tools/profile_digest.py profiles 6 > profiles/r06_digest.txt 2>&1
cp profiles/traffic_latest.json profiles/r06_digest.txt gpurun_out/r6/
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/line_final3.json 2> gpurun_out/r6/line_final3.err
python tools/show_line.py gpurun_out/r6/line_final3.json
timeout 900 python tools/bench_infer_e2e.py --questions 1024 --batch_size 512 --max_new_tokens 144 --lanes 2 --hold 384 > gpurun_out/r6/infer_e2e3.json 2> gpurun_out/r6/infer_e2e3.err; tail -c 200 gpurun_out/r6/infer_e2e3.json

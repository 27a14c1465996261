#!/bin/bash
# round 6, ninth GPU call: the final kernel sources -- the attention tests on the re-instantiated kernel, then the round's profiles
mkdir -p gpurun_out/r6
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_ops_kernels.py tests/test_gpu_3b_shape.py -m gpu -q -x > gpurun_out/r6/tests_job9.txt 2>&1; tail -3 gpurun_out/r6/tests_job9.txt
timeout 2400 bash tools/profile_round6.sh r06 fast > gpurun_out/r6/profile_round6c.txt 2>&1; tail -3 gpurun_out/r6/profile_round6c.txt

#!/bin/bash
# Ablation of the BK = 32 ring K loop (the decode step's gate/up and down at 513 .. 768 rows): alternative libraries with the MFMAs
# compiled out (ABLATE = 1) and with the fragment reads compiled out too (ABLATE = 2: only waits, barriers and LDS-DMA remain),
# timed by tools/bench_wide.py with ze_tune 18:1 (the plain loop).  The results are WRONG by construction: measurement only.
# build (CPU):  tools/probes/ring_ablate.sh build      run (GPU box):  tools/probes/ring_ablate.sh run
set -u
cd "$(dirname "$0")/../.."
CS=zoomearth_amd/csrc
case "${1:-build}" in
build)
    for a in 1 2; do
        ( /opt/rocm/bin/hipcc -DZE_RING_ABLATE=$a -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-strict-aliasing -fno-slp-vectorize -Iinclude -c $CS/ze_gemm.hip -o /tmp/ze_gemm_ablate$a.o 2>/dev/null &&
          /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o zoomearth_amd/libze_ablate$a.so $(ls $CS/*.o | grep -v '/ze_gemm\.o$') /tmp/ze_gemm_ablate$a.o && echo built $a ) &
    done
    wait
    ;;
run)
    mkdir -p gpurun_out
    for a in 1 2; do
        ZE_LIB_PATH=$PWD/zoomearth_amd/libze_ablate$a.so ZE_COUNTS=${ZE_COUNTS:-576} timeout 300 python tools/bench_wide.py 768 18:1 2>&1 | grep tune | sed "s/^/ablate$a /" | cut -c1-150
    done
    ;;
esac

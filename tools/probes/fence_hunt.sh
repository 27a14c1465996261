#!/bin/bash
# Root-cause hunt for the mis-scheduled instantiations of k_attn_decode_wave_long.
# build: cross-compiles one probe binary (+ the device assembly, /tmp/fence_hunt/*.s) per variant of the fence / asm-statement
#        hooks of ze_attn_batch.hip (tools/probes/fence_variants.h), for 192- and 256-key parts (ROUNDS = 3, 4: the two
#        instantiations that were wrong unfenced).
# run:   (on the GPU box) runs every binary on dense random rows against float64 and writes gpurun_out/fence_hunt/*.txt
set -u
cd "$(dirname "$0")/../.."
BIN=tools/probes/bin
FL="--offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -fno-slp-vectorize -Iinclude -Izoomearth_amd/csrc -include tools/probes/fence_variants.h"
NAMES=(fenced nofence onlyA onlyB post64 pre64 prelgk postlgk cvtpost cvtpre cvtbuiltin fenced_asmcvt)
case "${1:-build}" in
build)
    mkdir -p $BIN /tmp/fence_hunt
    for fh in ${FH_LIST:-0 1 2 3 4 5 6 7 8 9 10}; do
        for R in ${R_LIST:-3 4}; do
            (
                n=${NAMES[$fh]}
                /opt/rocm/bin/hipcc $FL -DFH=$fh -DAW_LONG_ROUNDS=$R -o $BIN/fh_${n}_r$R tools/probes/attn_wave_probe.hip zoomearth_amd/csrc/ze_attn_batch.hip 2>/tmp/fence_hunt/${n}_r$R.log
                rc=$?
                /opt/rocm/bin/hipcc $FL -DFH=$fh -DAW_LONG_ROUNDS=$R --cuda-device-only -S -o /tmp/fence_hunt/${n}_r$R.s zoomearth_amd/csrc/ze_attn_batch.hip 2>/dev/null
                echo "built $n r$R: $rc $?"
            ) &
        done
        wait
    done
    ;;
run)
    mkdir -p gpurun_out/fence_hunt
    for b in $BIN/fh_*; do
        n=$(basename $b)
        timeout 60 $b > gpurun_out/fence_hunt/$n.txt 2>&1
        echo "== $n: $(grep -c 'knob 0' gpurun_out/fence_hunt/$n.txt) lines; worst knob-0 error vs float64: $(grep 'knob 0' gpurun_out/fence_hunt/$n.txt | sed 's/.*float64| \([0-9.]*\).*/\1/' | sort -g | tail -1)"
    done
    ;;
esac

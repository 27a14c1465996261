#!/bin/bash
# SQ-side counters of the decode step's kernels at N chains (tools/pmc_kernel.py wide<N>): where the waves of a kernel spend
# their cycles.  Two passes (8 SQ slots each).  usage (GPU box): tools/pmc_sq.sh 580 [tag]
set -u
cd "$(dirname "$0")/.."
N=${1:-580}; TAG=${2:-sq}
OUT=gpurun_out/pmc_sq; mkdir -p $OUT
export TMPDIR=/tmp
pass() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/p_sq
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/p_sq -- python3 tools/pmc_kernel.py wide$N > $OUT/${TAG}_${N}_${name}.json 2> $OUT/${TAG}_${N}_${name}.log
  python3 tools/summarize_prof.py /tmp/p_sq $OUT/${TAG}_${N}_${name}.csv --delete-raw
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $OUT/sq_counter_names.txt

#!/bin/bash
# tools/gpu_r2h.sh — instruction-cache / instruction-mix counter passes over the Poseidon throughput kernels.
set -o pipefail
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${1:-r2h}
OUT=$PWD/gpurun_out
pmc() {  # name, counters, what
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pmc_$1_$R -- python3 $OUT/../tools/kern_once.py $3 2 > $OUT/pmc_$1_$R.log 2>&1; echo "pmc $1 exit $?")
  f=$(find /tmp/pmc_$1_$R -name "*counter_collection*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/pmc_$1_$R.csv && wc -l $OUT/pmc_$1_$R.csv
}
pmc ica "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" all &&
pmc icb "SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM" all
echo "session done"

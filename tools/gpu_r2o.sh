#!/bin/bash
# tools/gpu_r2o.sh — SQ counters of the nine-limb NTT kernels (one --pmc pass each)
set -o pipefail
mkdir -p gpurun_out; cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; R=${1:-r2o}; OUT=$PWD/gpurun_out
pmc() { (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pmc_$1_$R -- python3 $OUT/../tools/kern_once.py $3 2 > $OUT/pmc_$1_$R.log 2>&1; echo "pmc $1 exit $?")
  f=$(find /tmp/pmc_$1_$R -name "*counter_collection*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/pmc_$1_$R.csv && wc -l $OUT/pmc_$1_$R.csv; }
pmc sqa "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" ntt &&
pmc sqb "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" ntt &&
pmc fetch "FETCH_SIZE" ntt && pmc write "WRITE_SIZE" ntt
echo "session done"

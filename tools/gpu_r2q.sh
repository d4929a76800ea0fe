#!/bin/bash
# tools/gpu_r2q.sh — round-2 closing session: the whole GPU suite, the bench line (+ reference-schema CSV), rocprofv3 kernel stats of the
# timed steps, then the counter passes (HBM bytes of the 2^23 coset NTT; SQ counters of the hot kernels), each --pmc pass on its own.
set -o pipefail
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${1:-r2q}
OUT=$PWD/gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/gpu_tests_$R.log 2>&1; rc=$?; echo "tests exit $rc"; tail -14 $OUT/gpu_tests_$R.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --steps 10 --warmup 2 --csv $OUT/benchmarkdata_$R.csv > $OUT/bench_$R.json 2> $OUT/bench_$R.err; rc=$?; echo "bench exit $rc"; cut -c1-400 $OUT/bench_$R.json; tail -3 $OUT/bench_$R.err
[ $rc -eq 0 ] || exit $rc
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$R -- python3 $OUT/../bench.py --steps-only --steps 3 --warmup 1 > $OUT/rocprof_bench_$R.log 2>&1; echo "rocprof stats exit $?") &&
(mkdir -p $OUT/prof_$R && find /tmp/prof_$R -name "*stats*.csv" -exec cp {} $OUT/prof_$R/ \; ; ls $OUT/prof_$R)
pmc() {  # name, counters, what
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pmc_$1_$R -- python3 $OUT/../tools/kern_once.py $3 2 > $OUT/pmc_$1_$R.log 2>&1; echo "pmc $1 exit $?")
  f=$(find /tmp/pmc_$1_$R -name "*counter_collection*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/pmc_$1_$R.csv && wc -l $OUT/pmc_$1_$R.csv
}
pmc fetch "FETCH_SIZE" ntt &&
pmc write "WRITE_SIZE" ntt &&
pmc sqa "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" all &&
pmc sqb "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" all
echo "session done"

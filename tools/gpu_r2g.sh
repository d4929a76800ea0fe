#!/bin/bash
# tools/gpu_r2g.sh — round-2 session G: the whole GPU suite, then the bench line with the reference-bench CSV.
set -o pipefail
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${1:-r2g}
OUT=$PWD/gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/gpu_tests_$R.log 2>&1; rc=$?; echo "tests exit $rc"; tail -14 $OUT/gpu_tests_$R.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --steps 10 --warmup 2 --csv $OUT/benchmarkdata_$R.csv > $OUT/bench_$R.json 2> $OUT/bench_$R.err; echo "bench exit $?"; cut -c1-400 $OUT/bench_$R.json; cat $OUT/benchmarkdata_$R.csv; tail -3 $OUT/bench_$R.err

#!/bin/bash
# tools/gpu_r2a.sh — round-2 session A: new config tests first, then the whole GPU suite, step breakdown, one bench line.
set -o pipefail
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${1:-r2a}
timeout -k 10 600 python -m pytest tests/test_gpu_r2_configs.py -m gpu -x -q --durations=15 > gpurun_out/gpu_tests_new_$R.log 2>&1; rc=$?; echo "new tests exit $rc"; tail -25 gpurun_out/gpu_tests_new_$R.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/gpu_tests_$R.log 2>&1; rc=$?; echo "pytest exit $rc" | tee -a gpurun_out/gpu_tests_$R.log
tail -15 gpurun_out/gpu_tests_$R.log
[ $rc -eq 0 ] &&
timeout -k 10 200 python tools/step_breakdown.py > gpurun_out/step_breakdown_$R.txt 2>&1 && cat gpurun_out/step_breakdown_$R.txt &&
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err && echo "bench done" && cat gpurun_out/bench_$R.json

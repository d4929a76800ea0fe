#!/bin/bash
# tools/gpu_quick.sh — short gpurun call: parity tests, kernel timings, one bench line.
set -o pipefail
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${1:-q}
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_$R.log 2>&1; rc=$?; echo "pytest exit $rc" | tee -a gpurun_out/gpu_tests_$R.log
tail -3 gpurun_out/gpu_tests_$R.log
[ $rc -eq 0 ] &&
timeout -k 10 300 python tests/gpu_microbench.py > gpurun_out/microbench_$R.jsonl 2>&1 && echo "microbench done" && cat gpurun_out/microbench_$R.jsonl &&
timeout -k 10 400 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err && echo "bench done" && cat gpurun_out/bench_$R.json

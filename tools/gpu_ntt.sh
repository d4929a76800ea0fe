#!/bin/bash
# tools/gpu_ntt.sh — NTT parity subset + NTT timings.
set -o pipefail
mkdir -p gpurun_out; cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
R=${1:-n}
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or lde or six_step or bench_step or fold_and" > gpurun_out/gpu_tests_$R.log 2>&1; rc=$?; tail -2 gpurun_out/gpu_tests_$R.log
[ $rc -eq 0 ] && timeout -k 10 300 python tests/gpu_microbench.py > gpurun_out/microbench_$R.jsonl 2>&1 && grep -E "\"ntt\"|six_step" gpurun_out/microbench_$R.jsonl

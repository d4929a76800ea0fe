#!/bin/bash
# tools/gpu_r2d.sh — round-2 session D: sum-check (N4) tests, compiled C++ ABI client, LDE zero-skip regression, bench line.
set -o pipefail
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${1:-r2d}
OUT=$PWD/gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_r2_sumcheck.py tests/test_capi_client.py tests/test_gpu_r2_configs.py -m gpu -x -q -s --durations=8 > $OUT/gpu_tests_$R.log 2>&1; rc=$?; echo "tests exit $rc"; grep -E "prove_plain|prove_mf" $OUT/gpu_tests_$R.log | head -20; tail -14 $OUT/gpu_tests_$R.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or lde or fold or bench_step" > $OUT/gpu_tests_ntt_$R.log 2>&1; rc=$?; echo "ntt tests exit $rc"; tail -4 $OUT/gpu_tests_ntt_$R.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --e2e-log 0 > $OUT/bench_$R.json 2> $OUT/bench_$R.err; echo "bench exit $?"; cut -c1-300 $OUT/bench_$R.json; python3 -c "
import json; d=json.load(open('$OUT/bench_$R.json')); print(json.dumps(d['roofline'])); print(d['ms_per_step'])"

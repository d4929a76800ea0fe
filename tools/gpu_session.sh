#!/bin/bash
# tools/gpu_session.sh — one gpurun call: parity tests, instruction rates, kernel timings, bench, rocprof stats.
set -o pipefail
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${1:-r01}
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_$R.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/gpu_tests_$R.log
tail -3 gpurun_out/gpu_tests_$R.log
timeout -k 10 120 ./tools/bin/instr_rates > gpurun_out/instr_rates_$R.jsonl 2>&1 && echo "instr_rates done" &&
timeout -k 10 300 python tests/gpu_microbench.py > gpurun_out/microbench_$R.jsonl 2>&1 && echo "microbench done" && cat gpurun_out/microbench_$R.jsonl &&
timeout -k 10 400 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err && echo "bench done" && cat gpurun_out/bench_$R.json &&
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$R -- python3 $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OLDPWD/gpurun_out/rocprof_bench_$R.log 2>&1; echo "rocprof exit $?") &&
(mkdir -p gpurun_out/prof_$R && find /tmp/prof_$R -name "*stats*.csv" -exec cp {} gpurun_out/prof_$R/ \; ; ls gpurun_out/prof_$R)

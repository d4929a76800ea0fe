#!/bin/bash
# tools/gpu_session.sh — one gpurun call: parity tests, instruction rates, kernel timings, bench, rocprof stats.
set -o pipefail
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${1:-r01}
mkdir -p tools/bin
[ -x tools/bin/instr_rates ] || hipcc --offload-arch=gfx950 -O3 tools/instr_rates.hip -o tools/bin/instr_rates
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_$R.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/gpu_tests_$R.log
tail -3 gpurun_out/gpu_tests_$R.log
timeout -k 10 120 ./tools/bin/instr_rates > gpurun_out/instr_rates_$R.jsonl 2>&1 && echo "instr_rates done" &&
timeout -k 10 300 python tests/gpu_microbench.py > gpurun_out/microbench_$R.jsonl 2>&1 && echo "microbench done" && cat gpurun_out/microbench_$R.jsonl &&
timeout -k 10 400 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err && echo "bench done" && cat gpurun_out/bench_$R.json &&
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$R -- python3 $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OLDPWD/gpurun_out/rocprof_bench_$R.log 2>&1; echo "rocprof exit $?") &&
(mkdir -p gpurun_out/prof_$R && find /tmp/prof_$R -name "*stats*.csv" -exec cp {} gpurun_out/prof_$R/ \; ; ls gpurun_out/prof_$R) &&
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f_$R -- python3 $OLDPWD/tools/ntt_once.py 23 2 > $OLDPWD/gpurun_out/pmc_fetch_$R.log 2>&1; echo "pmc fetch exit $?") &&
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w_$R -- python3 $OLDPWD/tools/ntt_once.py 23 2 > $OLDPWD/gpurun_out/pmc_write_$R.log 2>&1; echo "pmc write exit $?") &&
(mkdir -p gpurun_out/pmc_$R && find /tmp/pmc_f_$R /tmp/pmc_w_$R -name "*counter_collection*.csv" -exec sh -c 'cp "$1" gpurun_out/pmc_'$R'/$(echo "$1" | tr / _ | tail -c 60)' _ {} \; ; ls gpurun_out/pmc_$R) &&
(STARK_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 1 --warmup 1 --log-trace 16 > gpurun_out/bench2_gloo_$R.json 2> gpurun_out/bench2_gloo_$R.err; echo "bench2 exit $?"; cat gpurun_out/bench2_gloo_$R.json | cut -c1-300)

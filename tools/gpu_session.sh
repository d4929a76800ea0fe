#!/bin/bash
# tools/gpu_session.sh — THE reproducible GPU session of a round (one gpurun call, about 14 GPU-minutes):
#   /usr/local/graft/bin/gpurun --timeout 1190 -- 'bash tools/gpu_session.sh r03'
# 1. the whole `-m gpu` suite; 2. the bench line (+ the reference-schema CSV); 3. rocprofv3 kernel stats of the timed steps only;
# 4. the counter passes, each --pmc set in its own run (HBM bytes of the 2^23 coset NTT; SQ counters of the hot kernels);
# 5. the serial-sponge timings (three-wave against one-wave kernel, where a permutation's time goes) and the product-chain micro-benchmarks;
# 6. the small-launch latencies (five-wave against one-wave / wave-pair kernels), the launch trace of a prove given f0, the coset NTT with merged against separate tables;
# 7. a two-rank rehearsal of `bench.py --gpus 2` on the one GPU (gloo) at the size whose step roots have an oracle golden.
# Everything lands in gpurun_out/ with the round tag; tools/pmc_summary.py turns the counter CSVs into the tracked summaries under profiles/.
set -o pipefail
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${1:-rXX}
OUT=$PWD/gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/gpu_tests_$R.log 2>&1; rc=$?; echo "tests exit $rc"; tail -14 $OUT/gpu_tests_$R.log
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
timeout -k 10 200 python tools/sponge_timing.py 2>/dev/null | grep log_n0 > $OUT/sponge_timing_$R.jsonl; cat $OUT/sponge_timing_$R.jsonl
timeout -k 10 200 python tools/sponge_debug_timing.py 2>/dev/null | grep dbg > $OUT/sponge_breakdown_$R.jsonl
[ -x tools/bin/chain_row ] && (cd tools && ./bin/chain_row 2000 > $OUT/chain_row_$R.txt && python3 chain_row_check.py $OUT/chain_row_$R.txt | tail -1; ./bin/chain_bench > $OUT/chain_bench_$R.txt 2>&1)
timeout -k 10 300 python tools/latency_timing.py 2>/dev/null | grep one_wave > $OUT/latency_timing_$R.jsonl; cat $OUT/latency_timing_$R.jsonl
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ptrace_$R -- python3 $OUT/../tools/prove_trace.py 16 3 > $OUT/ptrace_$R.log 2>&1; echo "prove trace exit $?") &&
(f=$(find /tmp/ptrace_$R -name "*kernel_trace.csv" | head -1); python tools/prove_trace.py summarize "$f" > $OUT/prove_trace_k16_$R.txt; tail -2 $OUT/prove_trace_k16_$R.txt)
for m in 1 0 1 0; do timeout -k 10 120 python tools/variant_bench.py stark_mlwe_amd/libstark_mlwe_hip.so merged$m ntt_merged_coset=$m ntt 2>/dev/null | grep variant | cut -c1-330; done > $OUT/ntt_merged_ab_$R.jsonl; cat $OUT/ntt_merged_ab_$R.jsonl
(STARK_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 1 --warmup 1 --log-trace 19 --no-cpu-baseline > $OUT/bench2_gloo_$R.json 2> $OUT/bench2_gloo_$R.err; echo "bench2 exit $?"; cut -c1-300 $OUT/bench2_gloo_$R.json)
echo "session done"

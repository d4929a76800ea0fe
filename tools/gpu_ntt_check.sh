#!/bin/bash
# NTT / LDE parity tests, then the HBM-traffic counters of the 2^23 coset NTT (separate --pmc passes) and its timing
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$PWD/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r2_configs.py tests/test_gpu_r3_step.py -m gpu -q -k "ntt or lde or fft or six_step or bench_step or sharded" 2>&1 | tail -6
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $OUT/../tools/kern_once.py ntt 2 > $OUT/pmc_ntt_$c.log 2>&1; echo "pmc $c exit $?")
  f=$(find /tmp/pmc_$c -name "*counter_collection*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/pmc_ntt_$c.csv
done
python tools/pmc_summary.py traffic $OUT/pmc_ntt_FETCH_SIZE.csv $OUT/pmc_ntt_WRITE_SIZE.csv 23 | cut -c1-600
for m in 1 0 1 0; do python tools/variant_bench.py stark_mlwe_amd/libstark_mlwe_hip.so merged$m ntt_merged_coset=$m ntt 2>/dev/null | grep variant | cut -c1-300; done
python bench.py --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_ntt_check.json; python -c "import json;d=json.load(open('$OUT/bench_ntt_check.json'));print(d['ms_per_step'], d['roots_match_golden'], d['roofline']['avg_ms'], d['roofline']['lde_2^20_to_2^23_ms_per_column'])"

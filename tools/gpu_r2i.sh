#!/bin/bash
# tools/gpu_r2i.sh — NTT engine check: the NTT/LDE/six-step parity tests, then old-vs-new timings of the 2^23 coset transform.
set -o pipefail
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
R=${1:-r2i}
OUT=$PWD/gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r2_configs.py tests/test_gpu_zz_dist.py -m gpu -x -q -k "ntt or lde or six_step or sharded or dist or domain or powers" --durations=6 > $OUT/gpu_tests_$R.log 2>&1; rc=$?; echo "tests exit $rc"; tail -12 $OUT/gpu_tests_$R.log
[ $rc -eq 0 ] || exit $rc
for v in tools/bin/libvar_carry.so stark_mlwe_amd/libstark_mlwe_hip.so; do timeout -k 10 150 python tools/variant_bench.py $v >> $OUT/variants_$R.jsonl || exit 1; done; cat $OUT/variants_$R.jsonl

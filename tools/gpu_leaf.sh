#!/bin/bash
# tools/gpu_leaf.sh — Poseidon parity subset + kernel timings (short iteration loop for the hash kernels).
set -o pipefail
mkdir -p gpurun_out; cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
R=${1:-l}
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "permute or hash or leaf or merkle or prove_bytes or fingerprint" > gpurun_out/gpu_tests_$R.log 2>&1; rc=$?; tail -2 gpurun_out/gpu_tests_$R.log
[ $rc -eq 0 ] && timeout -k 10 300 python tests/gpu_microbench.py > gpurun_out/microbench_$R.jsonl 2>&1 && grep -E "leaf_pair|merkle_level|column_sponge" gpurun_out/microbench_$R.jsonl

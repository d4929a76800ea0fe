#!/bin/bash
# tools/gpu_r2c.sh — round-2 session C: configs[2] golden at 2^22, the multi-GPU split rehearsed on one GPU, 2-rank bench rehearsal (gloo).
set -o pipefail
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${1:-r2c}
OUT=$PWD/gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_r2_configs.py tests/test_gpu_zz_dist.py -m gpu -x -q --durations=8 > $OUT/gpu_tests_$R.log 2>&1; rc=$?; echo "tests exit $rc"; tail -16 $OUT/gpu_tests_$R.log
[ $rc -eq 0 ] || exit $rc
STARK_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --log-trace 16 --e2e-log 0 > $OUT/bench2_gloo_$R.json 2> $OUT/bench2_gloo_$R.err; echo "bench2 exit $?"; cut -c1-1200 $OUT/bench2_gloo_$R.json; tail -5 $OUT/bench2_gloo_$R.err

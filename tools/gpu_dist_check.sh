#!/bin/bash
# the multi-process tests, then the two-rank rehearsal of `bench.py --gpus 2` on the one GPU (gloo) at the size whose step roots have an oracle golden
set -o pipefail
mkdir -p gpurun_out; OUT=$PWD/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_zz_dist.py -m gpu -q 2>&1 | tail -5
(STARK_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 1 --warmup 1 --log-trace 19 --no-cpu-baseline > $OUT/bench2_gloo.json 2> $OUT/bench2_gloo.err; echo "bench2 exit $?"; cut -c1-1400 $OUT/bench2_gloo.json; tail -3 $OUT/bench2_gloo.err)

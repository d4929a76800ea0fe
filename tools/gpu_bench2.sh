#!/bin/bash
# tools/gpu_bench2.sh — two-rank rehearsal of bench.py on one GPU (gloo), repeated; prints the dist_prove entry of each run.
set -o pipefail
mkdir -p gpurun_out; cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8; do
  STARK_DIST_CHECK=1 STARK_BENCH_BACKEND=gloo timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29650 + i)) bench.py --gpus 2 --steps 1 --warmup 1 --log-trace 16 > gpurun_out/b2_$i.json 2> gpurun_out/b2_$i.err
  echo "run $i exit $?"
  python3 - <<PY
import json
try:
    l=[x for x in open('gpurun_out/b2_$i.json') if x.startswith('{')][-1]; d=json.loads(l); print(d['dist_prove_given_f0'])
except Exception as e: print('no line', e)
PY
  grep -n "what()" gpurun_out/b2_$i.err | head -3
done

#!/usr/bin/env python3
"""tools/batch_scaling.py — stark_deep_fri_prove_batch_dev: B independent 2^14-row traces per call, B = 1 .. 128.  The 4 * B serial column sponges
run concurrently (one five-wave workgroup each); the B tails (merge, commit, queries) run one after another."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from stark_mlwe_amd.api import Context, DeepFriParams
ctx = Context(0); lib = ctx.lib
k = 14; n0 = 1 << k; prm = DeepFriParams([16, 16, 8], 32, 0xDEEFBAAD)
traces, keep = [], []
for p in range(128):
    cols = [torch.empty((n0, 4), dtype=torch.int64, device="cuda") for _ in range(4)]
    for c in range(4): ctx._chk(lib.stark_synth_column_dev(ctx.h, 0xBA7C0000 + p, c, 0, n0, C.c_void_p(cols[c].data_ptr())))
    keep.append(cols); traces.append([c.data_ptr() for c in cols])
torch.cuda.synchronize()
ctx.deep_fri_prove_batch_dev(traces[:8], n0, prm)      # warm-up: creates all four worker contexts (a one-time cost of ~85 ms)
for B in (1, 2, 4, 8, 16, 32, 64, 128):
    t0 = time.perf_counter(); got = ctx.deep_fri_prove_batch_dev(traces[:B], n0, prm); dt = time.perf_counter() - t0
    sponge_ms = got[0][2][0] - 0  # stage 0 of proof 0 = the shared sponge stage + its own merge
    print(json.dumps({"log_n0": k, "batch": B, "seconds": dt, "proves_per_s": B / dt, "trace_rows_per_s": B * n0 / dt, "shared_sponge_stage_ms": sponge_ms}), flush=True)
ctx.close()

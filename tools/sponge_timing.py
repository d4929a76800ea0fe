#!/usr/bin/env python3
"""tools/sponge_timing.py — us per dependent permutation of the serial column sponge (build_f0, crates/deep_ali/src/fri.rs:548-557):
five-wave kernel (poseidon_chain.hpp) against the round-2 one-wave kernel (option sponge_one_wave), same digests."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from stark_mlwe_amd.api import Context

ctx = Context(0); lib = ctx.lib
out = []
for k in (12, 16):
    n0 = 1 << k
    cols = [torch.empty((n0, 4), dtype=torch.int64, device="cuda") for _ in range(4)]
    for c in range(4):
        ctx._chk(lib.stark_synth_column_dev(ctx.h, 0x5EED0000 + k, c, 0, n0, C.c_void_p(cols[c].data_ptr())))
    f0 = torch.empty((n0, 4), dtype=torch.int64, device="cuda")
    res = {}
    for mode in (0, 1):
        ctx._chk(lib.stark_ctx_set_option(ctx.h, b"sponge_one_wave", mode))
        aux = np.zeros(28, np.uint64)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx._chk(lib.stark_build_f0_dev(ctx.h, *[C.c_void_p(c.data_ptr()) for c in cols], n0, C.c_void_p(f0.data_ptr()), aux.ctypes.data_as(C.c_void_p)))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[mode] = (dt, aux.copy(), f0.cpu().numpy().copy())
    same = bool((res[0][1] == res[1][1]).all() and (res[0][2] == res[1][2]).all())
    nperm = n0 // 16 + 2
    out.append({"log_n0": k, "five_wave_us_per_permutation": res[0][0] * 1e6 / nperm, "one_wave_us_per_permutation": res[1][0] * 1e6 / nperm,
                "build_f0_ms_five_wave": res[0][0] * 1e3, "build_f0_ms_one_wave": res[1][0] * 1e3, "same_digests_and_f0": same})
    print(json.dumps(out[-1]), flush=True)
ctx._chk(lib.stark_ctx_set_option(ctx.h, b"sponge_one_wave", 0))
ctx.close()
sys.exit(0 if all(o["same_digests_and_f0"] for o in out) else 1)

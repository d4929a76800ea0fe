#!/usr/bin/env python3
"""timing experiments on the five-wave sponge (option sponge_debug; digests are wrong in these modes) — where a permutation's time goes"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from stark_mlwe_amd.api import Context
ctx = Context(0); lib = ctx.lib
k = 14; n0 = 1 << k
cols = [torch.empty((n0, 4), dtype=torch.int64, device="cuda") for _ in range(4)]
for c in range(4): ctx._chk(lib.stark_synth_column_dev(ctx.h, 0x5EED0000 + k, c, 0, n0, C.c_void_p(cols[c].data_ptr())))
out = torch.empty((4, 4), dtype=torch.int64, device="cuda")
nperm = n0 // 16 + 2
for dbg, what in [(0, "full kernel"), (1, "A does not wait for E/H"), (2, "B without its product"), (4, "C without its lane products"), (3, "A free-running, B idle"), (7, "A free-running, B and C idle"),
                  (8, "no full rounds"), (16, "no partial rounds (full rounds + absorb only)"), (24, "neither (absorb + loop only)")]:
    ctx._chk(lib.stark_ctx_set_option(ctx.h, b"sponge_debug", dbg))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx._chk(lib.stark_tr_hash_fields_tagged_dev(ctx.h, None, b"ALI/A", C.c_void_p(cols[0].data_ptr()), n0, 1, C.c_void_p(out.data_ptr())))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"dbg": dbg, "what": what, "us_per_permutation": dt * 1e6 / nperm}), flush=True)
ctx._chk(lib.stark_ctx_set_option(ctx.h, b"sponge_debug", 0)); ctx.close()

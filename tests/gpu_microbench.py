"""Not a test: kernel-level timings printed as JSON lines (run on the GPU box by hand / gpurun)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stark_mlwe_amd.api import Context, PALLAS_FR, _ptr


def main():
    dev = torch.device("cuda", 0)
    ctx = Context(0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)); lib = ctx.lib
    def dbuf(rows): return torch.empty((rows, 4), dtype=torch.int64, device=dev)
    def timed(fn, reps=3):
        fn(); ms = C.c_float(); ctx._chk(lib.stark_timer_start(ctx.h))
        for _ in range(reps): fn()
        ctx._chk(lib.stark_timer_stop_ms(ctx.h, C.byref(ms))); return ms.value / reps
    for lg in (16, 20, 22, 23, 24):
        x = dbuf(1 << lg); ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 7, 0, 1 << lg, C.c_void_p(x.data_ptr())))
        ms = timed(lambda: ctx._chk(lib.stark_ntt_dev(ctx.h, PALLAS_FR, C.c_void_p(x.data_ptr()), lg, 0, None)))
        print(json.dumps({"kernel": "ntt", "log_n": lg, "ms": ms, "GBps": 64.0 * (1 << lg) / ms / 1e6}), flush=True)
        del x
    tp = ctx.transcript_params()
    for lg in (16, 20, 22):
        n = 1 << lg; f, fn, h = dbuf(n), dbuf(n // 16), dbuf(n)
        ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 0, 0, n, C.c_void_p(f.data_ptr()))); ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 1, 0, n // 16, C.c_void_p(fn.data_ptr())))
        ms = timed(lambda: ctx._chk(lib.stark_leaf_pair_hash_dev(ctx.h, tp.h, C.c_void_p(f.data_ptr()), C.c_void_p(fn.data_ptr()), n, 16, C.c_void_p(h.data_ptr()))), reps=2)
        print(json.dumps({"kernel": "leaf_pair", "log_n": lg, "ms": ms, "leaves_per_s": n / ms * 1e3, "us_per_leaf_per_cu": ms * 1e3 / n * 256}), flush=True)
        mp = ctx.poseidon_params_for_width(17); out = dbuf(n // 16)
        ms = timed(lambda: ctx._chk(lib.stark_poseidon_hash_ds_batch_dev(ctx.h, mp.h, 16, 0, 0, 0, C.c_void_p(h.data_ptr()), n, C.c_void_p(out.data_ptr()))), reps=2)
        print(json.dumps({"kernel": "merkle_level_a16", "log_n_in": lg, "ms": ms, "nodes_per_s": n / 16 / ms * 1e3}), flush=True)
        z = np.array([3, 5, 7, 9], np.uint64)
        ms = timed(lambda: ctx._chk(lib.stark_fri_fold_dev(ctx.h, C.c_void_p(f.data_ptr()), n, _ptr(z), 16, C.c_void_p(out.data_ptr()))))
        print(json.dumps({"kernel": "fri_fold_m16", "log_n": lg, "ms": ms, "GBps": 32.0 * n * (1 + 1 / 16) / ms / 1e6}), flush=True)
        del f, fn, h, out
    # serial column sponge (tr_hash_fields_tagged over one column): latency per dependent permutation
    for lg in (12, 16):
        n = 1 << lg; f, o = dbuf(n), dbuf(1)
        ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 0, 0, n, C.c_void_p(f.data_ptr())))
        ms = timed(lambda: ctx._chk(lib.stark_tr_hash_fields_tagged_dev(ctx.h, None, b"ALI/A", C.c_void_p(f.data_ptr()), n, 1, C.c_void_p(o.data_ptr()))), reps=1)
        print(json.dumps({"kernel": "column_sponge", "log_n": lg, "ms": ms, "us_per_permutation": ms * 1e3 / (n / 16)}), flush=True)
    # six-step building blocks on one GPU (world size 1): column pass + (identity exchange) + row pass
    from stark_mlwe_amd import dist as sd
    for lg, lr in ((20, 10), (24, 10)):
        slab = dbuf(1 << lg); ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 7, 0, 1 << lg, C.c_void_p(slab.data_ptr())))
        plan = sd.DistNtt(sd.HipProvider(ctx), lg, lr)
        def run():
            plan.p.ntt_columns(slab, plan.log_rows, plan.ncl, 0, lg, False); plan.p.ntt_rows(slab, plan.nrl, lg - lr, False, None)
        ms = timed(run)
        print(json.dumps({"kernel": "six_step_local_phases", "log_n": lg, "log_rows": lr, "ms": ms, "GBps": 64.0 * (1 << lg) / ms / 1e6}), flush=True)
        del slab
    ctx.close()


if __name__ == "__main__":
    main()

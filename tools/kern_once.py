"""tools/kern_once.py — runs the hot kernels a fixed number of times each, nothing else of weight, for counter profiling
(rocprofv3 --pmc ... -- python3 tools/kern_once.py):  k_ntt_strided x2 + k_ntt_last (2^23 coset NTT), k_leaf_pair2 (2^22 leaves),
k_hash_ds2<17> (one arity-16 level over 2^22 digests), k_tr_hash_chain (one serial sponge over 2^14 fields: the five-wave kernel).
Usage: kern_once.py [ntt|leaf|tree|sponge|all] [reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stark_mlwe_amd.api import Context, PALLAS_FR, _ptr
import bench
what = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
ctx = Context(0); lib = ctx.lib
P = lambda t: C.c_void_p(t.data_ptr())
if what in ("ntt", "all"):
    lg = 23
    x = torch.empty((1 << lg, 4), dtype=torch.int64, device=dev)
    ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 7, 0, 1 << lg, P(x)))
    coset = bench._mont_small(5)
    for _ in range(reps + 1):
        ctx._chk(lib.stark_ntt_dev(ctx.h, PALLAS_FR, P(x), lg, 0, _ptr(coset)))
    del x
if what in ("leaf", "tree", "all"):
    n = 1 << 22
    f = torch.empty((n, 4), dtype=torch.int64, device=dev); fn = torch.empty((n // 16, 4), dtype=torch.int64, device=dev); h = torch.empty((n, 4), dtype=torch.int64, device=dev)
    ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 0, 0, n, P(f))); ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 1, 0, n // 16, P(fn)))
    if what in ("leaf", "all"):
        for _ in range(reps):
            ctx._chk(lib.stark_leaf_pair_hash_dev(ctx.h, None, P(f), P(fn), n, 16, P(h)))
    if what in ("tree", "all"):
        out = torch.empty((n // 16, 4), dtype=torch.int64, device=dev)
        p17 = ctx.poseidon_params_for_width(17)
        for _ in range(reps):
            ctx._chk(lib.stark_poseidon_hash_ds_batch_dev(ctx.h, p17.h, 16, 0, 0, 0, P(f), n, P(out)))

if what in ("sponge", "all"):
    n = 1 << 14
    col = torch.empty((n, 4), dtype=torch.int64, device=dev); out = torch.empty((1, 4), dtype=torch.int64, device=dev)
    ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 2, 0, n, P(col)))
    for _ in range(reps):
        ctx._chk(lib.stark_tr_hash_fields_tagged_dev(ctx.h, None, b"ALI/A", P(col), n, 1, P(out)))
ctx.sync(); ctx.close()

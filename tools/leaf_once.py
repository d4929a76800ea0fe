"""One leaf-hash launch (2^20 leaves) for counter profiling."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stark_mlwe_amd.api import Context
dev = torch.device("cuda", 0); ctx = Context(0); lib = ctx.lib
n = 1 << 20
f = torch.empty((n, 4), dtype=torch.int64, device=dev); fn = torch.empty((n // 16, 4), dtype=torch.int64, device=dev); h = torch.empty((n, 4), dtype=torch.int64, device=dev)
ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 0, 0, n, C.c_void_p(f.data_ptr()))); ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 1, 0, n // 16, C.c_void_p(fn.data_ptr())))
tp = ctx.transcript_params()
for _ in range(2):
    ctx._chk(lib.stark_leaf_pair_hash_dev(ctx.h, tp.h, C.c_void_p(f.data_ptr()), C.c_void_p(fn.data_ptr()), n, 16, C.c_void_p(h.data_ptr())))
ctx.sync(); ctx.close()

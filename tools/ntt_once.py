"""Runs the bench's dominant HBM kernel group once per call site for profiling: a 2^23 coset NTT (3 launches)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stark_mlwe_amd.api import Context, PALLAS_FR, _ptr
import bench
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 23
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
ctx = Context(0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
x = torch.empty((1 << lg, 4), dtype=torch.int64, device=dev)
ctx._chk(ctx.lib.stark_synth_column_dev(ctx.h, 1, 7, 0, 1 << lg, C.c_void_p(x.data_ptr())))
coset = bench._mont_small(5)
for _ in range(reps):
    ctx._chk(ctx.lib.stark_ntt_dev(ctx.h, PALLAS_FR, C.c_void_p(x.data_ptr()), lg, 0, _ptr(coset)))
ctx.sync(); ctx.close()

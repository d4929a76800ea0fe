"""tools/step_breakdown.py — wall time of the pieces of one bench step (LDE x4, merge, fri_build) on one GPU (diagnostic)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from stark_mlwe_amd.api import Context, _ptr, PALLAS_FR
import bench
dev = torch.device("cuda", 0); ts = torch.cuda.Stream(dev); torch.cuda.set_stream(ts)
ctx = Context(0, C.c_void_p(ts.cuda_stream)); lib = ctx.lib
lg = 20; n, N = 1 << lg, 1 << (lg + 3)
db = lambda r: torch.empty((r, 4), dtype=torch.int64, device=dev)
cols = [db(n) for _ in range(4)]
for c in range(4): ctx._chk(lib.stark_synth_column_dev(ctx.h, 0x5EED0000 + lg, c, 0, n, C.c_void_p(cols[c].data_ptr())))
ext = [db(N) for _ in range(4)]; f0 = db(N)
coset = bench._mont_small(5); omega = bench._root_of_unity_pallas(lg + 3); z = bench._mont_small(0xC0FFEE)
sched = np.ascontiguousarray([16, 16, 8], dtype=np.uint64)
def T(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def lde():
    for c in range(4): ctx._chk(lib.stark_lde_dev(ctx.h, PALLAS_FR, C.c_void_p(cols[c].data_ptr()), lg, 3, _ptr(coset), C.c_void_p(ext[c].data_ptr())))
def merge(): ctx._chk(lib.stark_ali_merge_dev(ctx.h, *[C.c_void_p(e.data_ptr()) for e in ext], None, None, _ptr(omega), _ptr(z), N, C.c_void_p(f0.data_ptr()), None))
def build():
    st = C.c_void_p(); ctx._chk(lib.stark_fri_build_dev(ctx.h, C.c_void_p(f0.data_ptr()), N, _ptr(sched), 3, 0xDEEFBAAD, C.byref(st))); ctx._chk(lib.stark_fri_state_free(st))
print({"lde_x4_ms": T(lde), "merge_ms": T(merge), "fri_build_ms": T(build)})
h = db(N); fn = db(N // 16)
def leaf(): ctx._chk(lib.stark_leaf_pair_hash_dev(ctx.h, ctx.transcript_params().h, C.c_void_p(f0.data_ptr()), C.c_void_p(fn.data_ptr()), N, 16, C.c_void_p(h.data_ptr())))
def tree():
    t = C.c_void_p(); ctx._chk(lib.stark_merkle_build_dev(ctx.h, ctx.poseidon_params_for_width(17).h, 16, 0, C.c_void_p(h.data_ptr()), N, 0, None, 0, 0, 0, C.byref(t))); ctx.sync(); lib.stark_merkle_free(t)
print({"leaf_2^23_ms": T(leaf), "tree_over_2^23_ms": T(tree)})
ctx.close()

"""tools/lde_timing.py — stark_lde_dev 2^20 -> 2^23 (Pallas, coset 5) of the library given by path: ms per column and a checksum of the output."""
import ctypes as C, hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stark_mlwe_amd._abi as abi
path = os.path.abspath(sys.argv[1]); abi.lib_path = lambda: path
from stark_mlwe_amd.api import Context, PALLAS_FR, _ptr
import bench
dev = torch.device("cuda", 0)
ctx = Context(0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)); lib = ctx.lib
res = {"lib": os.path.basename(path)}
for lg, lb in ((20, 3), (16, 4), (18, 2)):
    x = torch.empty((1 << lg, 4), dtype=torch.int64, device=dev); y = torch.empty((1 << (lg + lb), 4), dtype=torch.int64, device=dev)
    ctx._chk(lib.stark_synth_column_dev(ctx.h, 3, 1, 0, 1 << lg, C.c_void_p(x.data_ptr())))
    coset = bench._mont_small(5)
    fn = lambda: ctx._chk(lib.stark_lde_dev(ctx.h, PALLAS_FR, C.c_void_p(x.data_ptr()), lg, lb, _ptr(coset), C.c_void_p(y.data_ptr())))
    fn(); ms = C.c_float(); ctx._chk(lib.stark_timer_start(ctx.h))
    for _ in range(10): fn()
    ctx._chk(lib.stark_timer_stop_ms(ctx.h, C.byref(ms)))
    res[f"lde_2^{lg}_x{1 << lb}_ms"] = round(ms.value / 10, 4); res[f"digest_{lg}_{lb}"] = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]
print(json.dumps(res)); ctx.close()

"""tools/variant_bench.py — times the Poseidon throughput kernels (and the 2^23 coset NTT) of ONE build of the library given by path,
and prints a checksum of the outputs so that variants can be compared for equal results.  Not product code.
Usage: python tools/variant_bench.py <path/to/libstark_variant.so> [tag]"""
import ctypes as C, hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stark_mlwe_amd._abi as abi
path = os.path.abspath(sys.argv[1]); tag = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(path)
opts = dict(kv.split("=") for kv in sys.argv[3].split(",")) if len(sys.argv) > 3 and sys.argv[3] else {}
only_ntt = len(sys.argv) > 4 and sys.argv[4] == "ntt"
abi.lib_path = lambda: path
from stark_mlwe_amd.api import Context, PALLAS_FR, _ptr
import bench
dev = torch.device("cuda", 0)
ctx = Context(0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)); lib = ctx.lib
P = lambda t: C.c_void_p(t.data_ptr())
def dbuf(rows): return torch.empty((rows, 4), dtype=torch.int64, device=dev)
def timed(fn, reps=3):
    fn(); ms = C.c_float(); ctx._chk(lib.stark_timer_start(ctx.h))
    for _ in range(reps): fn()
    ctx._chk(lib.stark_timer_stop_ms(ctx.h, C.byref(ms))); return ms.value / reps
def digest(t): return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]
for k, v in opts.items(): ctx.set_option(k, int(v))
res = {"variant": tag, "options": opts}
n = 1 << 22 if not only_ntt else 1 << 8
f, fn, h, out = dbuf(n), dbuf(n // 16), dbuf(n), dbuf(n // 16)
ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 0, 0, n, P(f))); ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 1, 0, n // 16, P(fn)))
tp = ctx.transcript_params(); mp = ctx.poseidon_params_for_width(17)
res["leaf_pair_2^22_ms"] = timed(lambda: ctx._chk(lib.stark_leaf_pair_hash_dev(ctx.h, tp.h, P(f), P(fn), n, 16, P(h))), reps=2)
res["leaf_digest"] = digest(h)
res["merkle_level_a16_2^22_ms"] = timed(lambda: ctx._chk(lib.stark_poseidon_hash_ds_batch_dev(ctx.h, mp.h, 16, 0, 0, 0, P(h), n, P(out))), reps=2)
res["level_digest"] = digest(out)
del f, fn, h, out
lg = 23; x = dbuf(1 << lg); ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 7, 0, 1 << lg, P(x)))
coset = bench._mont_small(5)
res["ntt_2^23_coset_ms"] = timed(lambda: ctx._chk(lib.stark_ntt_dev(ctx.h, PALLAS_FR, P(x), lg, 0, _ptr(coset))), reps=5)
ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 7, 0, 1 << lg, P(x))); ctx._chk(lib.stark_ntt_dev(ctx.h, PALLAS_FR, P(x), lg, 0, _ptr(coset)))
res["ntt_digest"] = digest(x)
print(json.dumps(res), flush=True)
ctx.close()

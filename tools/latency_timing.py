"""tools/latency_timing.py — the small-size launches of a prove (one Merkle level / one leaf layer of n nodes, prove-given-f0 at 2^12 .. 2^16), five-wave latency
kernels (default) against option "sponge_one_wave" (the one-wave / wave-pair kernels).  Prints JSON lines.  Not product code."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stark_mlwe_amd.api import Context, _ptr
import bench
dev = torch.device("cuda", 0)
ctx = Context(0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)); lib = ctx.lib
P = lambda t: C.c_void_p(t.data_ptr())
def timed(fn, reps=20):
    fn(); ms = C.c_float(); ctx._chk(lib.stark_timer_start(ctx.h))
    for _ in range(reps): fn()
    ctx._chk(lib.stark_timer_stop_ms(ctx.h, C.byref(ms))); return ms.value / reps
p17 = ctx.poseidon_params_for_width(17)
for one_wave in (0, 1):
    ctx.set_option("sponge_one_wave", one_wave)
    for n in (1, 16, 256, 512, 2048, 4096, 8192):
        f = torch.empty((n * 16, 4), dtype=torch.int64, device=dev); out = torch.empty((n, 4), dtype=torch.int64, device=dev)
        ctx._chk(lib.stark_synth_column_dev(ctx.h, 1, 0, 0, n * 16, P(f)))
        lvl = timed(lambda: ctx._chk(lib.stark_poseidon_hash_ds_batch_dev(ctx.h, p17.h, 16, 0, 0, 0, P(f), n * 16, P(out))))
        leaf = timed(lambda: ctx._chk(lib.stark_leaf_pair_hash_dev(ctx.h, None, P(f), None, n, 1, P(out))))
        print(json.dumps({"one_wave_option": one_wave, "nodes": n, "merkle_level_a16_us": round(lvl * 1e3, 1), "leaf_layer_us": round(leaf * 1e3, 1)}), flush=True)
    for k in (12, 14, 16):
        n0 = 1 << k
        f0 = torch.empty((n0, 4), dtype=torch.int64, device=dev); ctx._chk(lib.stark_synth_column_dev(ctx.h, 5, 0, 0, n0, P(f0)))
        best = 1e9
        sched = (C.c_size_t * len(bench.SCHEDULE))(*bench.SCHEDULE)
        for _ in range(5):
            ph = C.c_void_p(); t0 = time.perf_counter()
            ctx._chk(lib.stark_deep_fri_prove_dev(ctx.h, None, None, None, None, P(f0), n0, sched, len(bench.SCHEDULE), 32, bench.SEED_Z, C.byref(ph)))
            best = min(best, time.perf_counter() - t0); lib.stark_proof_free(ph)
        print(json.dumps({"one_wave_option": one_wave, "prove_given_f0_log_n0": k, "ms": round(best * 1e3, 3)}), flush=True)
ctx.set_option("sponge_one_wave", 0)
ctx.sync(); ctx.close()

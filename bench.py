#!/usr/bin/env python3
"""bench.py — the hot path of saholmes/stark-mlwe on MI355X, BASELINE.json config[1]:
"2^20 trace, blowup 8, single MI355X: NTT/LDE + Poseidon-Merkle kernels only".

One step (per GPU) = one pass of the hot path over one synthetic trace of 2^20 rows x 4 columns
(a, s, e, t), inputs already resident in HBM:
  1. LDE of the 4 columns, 2^20 -> 2^23 evaluations on the coset 5*<w>  (4 iNTT + 4 coset NTT);
  2. DEEP-ALI merge of the extended columns into f0 (fixed out-of-domain point z);
  3. fri_build_transcript(f0, [16,16,8]): 3 folds, leaf-pair Poseidon hashes of all layers and the
     4 Poseidon-Merkle trees (arity 16,16,8,2).
`value` = trace rows per second over all ranks (weak scaling: every rank proves its own 2^20-row
trace shard; the path partitions by trace, no data-path collective).  `roofline` is the Fr-NTT
(one 2^23 coset NTT = 3 kernel launches, algorithmic bytes 64*n); `cpu_baseline` is the C++ oracle
(a port of the reference's algorithm; the Rust reference cannot be built in this image) on a bounded
sample of the same workload, rank 0, N=1 only.

Usage: python bench.py --gpus N --steps K --warmup W   (N>1: launched by torch.distributed.run)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_TRACE = 20
LOG_BLOWUP = 3
SCHEDULE = [16, 16, 8]
SEED_Z = 0xDEEFBAAD
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FR_MULTS_T17, FR_MULTS_T9 = 21408, 5904   # reference-dense Fr-mults per permutation


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-trace", type=int, default=LOG_TRACE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("STARK_BENCH_BACKEND", "nccl")      # "gloo" only to rehearse N>1 on a one-GPU box
    ndev = torch.cuda.device_count()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    local_rank = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from stark_mlwe_amd.api import Context, _ptr, PALLAS_FR
    # ONE real stream for torch and the library: the default stream's handle is 0, which the C-ABI reads as "make your own"
    # (a non-blocking stream that is not ordered against torch's work), so a dedicated stream is made current for torch
    # and handed to the context.
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    ctx = Context(local_rank, C.c_void_p(tstream.cuda_stream))
    lib = ctx.lib

    log_n = args.log_trace
    n, N = 1 << log_n, 1 << (log_n + LOG_BLOWUP)
    L = len(SCHEDULE)
    sched = np.ascontiguousarray(SCHEDULE, dtype=np.uint64)

    def dbuf(rows):
        return torch.empty((rows, 4), dtype=torch.int64, device=dev)

    # synthetic trace shard of this rank: columns 0..3, rows [rank*n, (rank+1)*n)  (DESIGN.md "Synthetic inputs")
    seed = 0x5EED0000 + log_n
    cols = [dbuf(n) for _ in range(4)]
    for c in range(4):
        ctx._chk(lib.stark_synth_column_dev(ctx.h, seed, c, rank * n, n, C.c_void_p(cols[c].data_ptr())))
    ext = [dbuf(N) for _ in range(4)]
    f0 = dbuf(N)
    coset = _mont_small(5)           # multiplicative generator of Pallas Fr as the LDE coset shift
    omega = _root_of_unity_pallas(log_n + LOG_BLOWUP)
    z = _mont_small(0xC0FFEE)        # fixed out-of-domain point for the kernels-only step (z^N != 1 checked by the library)

    def step():
        for c in range(4):   # LDE = iNTT(2^20) + zero-pad + coset NTT(2^23)
            ctx._chk(lib.stark_lde_dev(ctx.h, PALLAS_FR, C.c_void_p(cols[c].data_ptr()), log_n, LOG_BLOWUP, _ptr(coset), C.c_void_p(ext[c].data_ptr())))
        ctx._chk(lib.stark_ali_merge_dev(ctx.h, *[C.c_void_p(e.data_ptr()) for e in ext], None, None, _ptr(omega), _ptr(z), N, C.c_void_p(f0.data_ptr()), None))
        st = C.c_void_p()
        ctx._chk(lib.stark_fri_build_dev(ctx.h, C.c_void_p(f0.data_ptr()), N, _ptr(sched), L, SEED_Z, C.byref(st)))
        roots = []
        for l in range(L + 1):
            r = np.zeros(4, np.uint64); ctx._chk(lib.stark_fri_layer_root(st, l, _ptr(r))); roots.append(r)
        ctx._chk(lib.stark_fri_state_free(st))
        return roots

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        roots = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed * 1e3 / args.steps

    # ---- kernel-level measurement (HIP events on the context's stream, outside the timed region) -------
    # dominant HBM-streaming kernel group of the headline metric: one 2^23 coset NTT (3 launches)
    scratch = dbuf(N)
    reps = 5
    scratch.copy_(ext[0])
    ctx._chk(lib.stark_ntt_dev(ctx.h, PALLAS_FR, C.c_void_p(scratch.data_ptr()), log_n + LOG_BLOWUP, 0, _ptr(coset)))
    ctx._chk(lib.stark_timer_start(ctx.h))
    for _ in range(reps):
        ctx._chk(lib.stark_ntt_dev(ctx.h, PALLAS_FR, C.c_void_p(scratch.data_ptr()), log_n + LOG_BLOWUP, 0, _ptr(coset)))
    ms = C.c_float(); ctx._chk(lib.stark_timer_stop_ms(ctx.h, C.byref(ms)))
    ntt_ms = ms.value / reps
    ntt_bytes = 64.0 * N
    ntt_gbps = ntt_bytes / (ntt_ms * 1e-3) / 1e9
    # Poseidon leaf kernel: one launch over 2^23 leaves
    h = dbuf(N); fnext = dbuf(N // 16)
    ctx._chk(lib.stark_leaf_pair_hash_dev(ctx.h, ctx.transcript_params().h, C.c_void_p(f0.data_ptr()), C.c_void_p(fnext.data_ptr()), N, 16, C.c_void_p(h.data_ptr())))
    ctx._chk(lib.stark_timer_start(ctx.h))
    ctx._chk(lib.stark_leaf_pair_hash_dev(ctx.h, ctx.transcript_params().h, C.c_void_p(f0.data_ptr()), C.c_void_p(fnext.data_ptr()), N, 16, C.c_void_p(h.data_ptr())))
    ctx._chk(lib.stark_timer_stop_ms(ctx.h, C.byref(ms)))
    leaf_ms = ms.value
    del scratch, h, fnext

    # ---- the reference's own bench shape (deep_fri_prove, schedule [16,16,8], r = 32; end_to_end.rs:187-270) ----
    def prove(log_k, given_f0):
        nk = 1 << log_k
        cs = [dbuf(nk) for _ in range(4)]
        for c in range(4):
            ctx._chk(lib.stark_synth_column_dev(ctx.h, 0x5EED0000 + log_k, c, 0, nk, C.c_void_p(cs[c].data_ptr())))
        ph = C.c_void_p()
        args = [C.c_void_p(x.data_ptr()) for x in cs]
        t0 = time.perf_counter()
        ctx._chk(lib.stark_deep_fri_prove_dev(ctx.h, *( [None] * 4 + [args[0]] if given_f0 else args + [None]), nk, _ptr(sched), L, 32, SEED_Z, C.byref(ph)))
        wall = (time.perf_counter() - t0) * 1e3
        res = {"log_n0": log_k, "r": 32, "wall_ms": wall, "proof_bytes": int(lib.stark_proof_len(ph)), "size_estimate": int(lib.stark_proof_size_estimate(ph)),
               "build_f0_ms": lib.stark_proof_stage_ms(ph, 0), "fri_build_ms": lib.stark_proof_stage_ms(ph, 1), "queries_encode_ms": lib.stark_proof_stage_ms(ph, 2)}
        lib.stark_proof_free(ph)
        return res
    prove(12, False)                                   # warm the constants / plans
    prove_e2e = prove(16, False)                       # DeepAliRealBuilder incl. the four serial column sponges
    prove_f0 = prove(20, True)                         # "prove given f0": everything after build_f0

    # ---- the one real exchange of the path: six-step NTT of ONE 2^24 vector sharded over all ranks (config[3]) ----
    # Reported beside the main line, never part of `value`; a failure here must not take the bench line down.
    dist_ntt = None
    try:
        from stark_mlwe_amd import dist as sd
        lgd = 24
        plan = sd.DistNtt(sd.HipProvider(ctx, device=dev), lgd, 10)
        slab = dbuf((1 << lgd) // world)
        ctx._chk(lib.stark_synth_column_dev(ctx.h, seed, 7, rank * ((1 << lgd) // world), (1 << lgd) // world, C.c_void_p(slab.data_ptr())))
        plan.forward(slab); barrier()
        t1 = time.perf_counter()
        for _ in range(3):
            plan.forward(slab)
        barrier()
        dt = (time.perf_counter() - t1) / 3
        if world > 1:
            tm = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX); dt = float(tm.item())
        dist_ntt = {"log_n": lgd, "ranks": world, "ms": dt * 1e3, "GBps_algorithmic": 64.0 * (1 << lgd) / dt / 1e9,
                    "all_to_all_bytes_per_rank": (1 << lgd) * 32 // world * (world - 1) // world,
                    "note": "column NTTs + twiddle, one all_to_all_single over RCCL, row NTTs; output in transposed block order"}
        del slab
    except Exception as ex:   # noqa: BLE001
        dist_ntt = {"error": repr(ex)[:300]}

    # ---- ONE trace of world * 2^20 rows block-sharded over the ranks, one proof (configs[3]/[4] shape): sharded folds,
    # leaf hashes and lower Merkle levels, all-gather of the tree tops, query values collected with one all-reduce.
    dist_prove = None
    try:
        from stark_mlwe_amd import dist as sd
        n_tot = world << log_n
        f0_blk = dbuf(n)
        ctx._chk(lib.stark_synth_column_dev(ctx.h, seed, 5, rank * n, n, C.c_void_p(f0_blk.data_ptr())))
        dp = sd.DistProver(sd.HipProvider(ctx, device=dev), n_tot, SCHEDULE, 32, 0xDEEFBAAD)
        dp.prove(None, None, None, None, f0_local=f0_blk); barrier()      # warm-up (plans, parameter tables)
        t1 = time.perf_counter()
        proof, est = dp.prove(None, None, None, None, f0_local=f0_blk)
        barrier()
        dt = time.perf_counter() - t1
        if world > 1:
            tm = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX); dt = float(tm.item())
        import hashlib
        dist_prove = {"log_n0": log_n + (world.bit_length() - 1), "ranks": world, "r": 32, "wall_ms": dt * 1e3, "rows_per_s": n_tot / dt,
                      "proof_bytes": len(proof), "size_estimate": est, "proof_sha256": hashlib.sha256(proof).hexdigest()[:16], **{k: round(v, 3) for k, v in dp.timings.items()},
                      "note": "deep_fri_prove stages after build_f0 on ONE trace sharded by contiguous blocks over the ranks (stark_mlwe_amd.dist.DistProver); every rank ends with the same proof bytes"}
        del f0_blk
    except Exception as ex:   # noqa: BLE001
        dist_prove = {"error": repr(ex)[:300]}

    out = None
    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "ntt_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"ntt_2^{log_n + LOG_BLOWUP}_bytes_per_transform")
            except Exception:
                traffic = None
        out = {
            "metric": "prove ms + Fr-NTT achieved GB/s, 2^20/2^24 trace at 1/2/4/8 GPUs",
            "value": world * n / elapsed * args.steps,
            "unit": "trace rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u256 (4x u64 Montgomery limbs, Pallas Fr)",
            "data": "synthetic",
            "config": {"workload": f"2^{log_n} trace x 4 columns per GPU, blowup 8: LDE (iNTT+coset NTT) + DEEP-ALI merge + FRI folds + Poseidon leaf hashes + Poseidon-Merkle trees (schedule [16,16,8]), kernels only",
                       "log_trace": log_n, "log_blowup": LOG_BLOWUP, "schedule": SCHEDULE, "field": "pallas_fr", "sharding": "one trace shard per GPU, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": f"Fr-NTT 2^{log_n + LOG_BLOWUP} coset forward (k_ntt_strided x2 + k_ntt_last)", "achieved": ntt_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ntt_gbps / HBM_PEAK_GBPS, "traffic": traffic, "algorithmic_bytes": ntt_bytes, "avg_ms": ntt_ms},
            "poseidon": {"kernel": "k_leaf_pair (t=17), 2^%d leaves" % (log_n + LOG_BLOWUP), "ms": leaf_ms, "leaves_per_s": N / (leaf_ms * 1e-3),
                         "reference_dense_fr_mults_per_s": FR_MULTS_T17 * N / (leaf_ms * 1e-3), "bound": "integer VALU (not HBM, not MFMA)"},
            "prove_end_to_end": dict(prove_e2e, note="deep_fri_prove with DeepAliRealBuilder on a 2^16-row trace (device-resident columns); build_f0 is the serial column sponge (fri.rs:548-557)"),
            "prove_given_f0": dict(prove_f0, note="deep_fri_prove stages after build_f0 on n0 = 2^20"),
            "dist_ntt": dist_ntt,
            "dist_prove_given_f0": dist_prove,
            "roots": ["".join(f"{int(x):016x}" for x in r[::-1]) for r in roots],
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(np, log_n)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def _mont_small(x):
    """Montgomery limbs of a small integer in Pallas Fr (host-side scalar; mirrors F::from(u64))."""
    import numpy as np
    p = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
    m = (x << 256) % p
    return np.array([(m >> (64 * i)) & (2**64 - 1) for i in range(4)], np.uint64)


def _root_of_unity_pallas(log_n):
    import numpy as np
    p = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
    w = pow(5, (p - 1) >> 32, p)
    for _ in range(32 - log_n):
        w = w * w % p
    m = (w << 256) % p
    return np.array([(m >> (64 * i)) & (2**64 - 1) for i in range(4)], np.uint64)


def cpu_baseline(np, log_n_gpu):
    """The oracle (CPU port of the reference's algorithm) on a bounded sample of the same workload:
    a 2^10-row trace (LDE to 2^13, merge, FRI build).  Single thread, like the reference."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    o = oracle_lib.Oracle()
    o.l.oracle_set_threads(1)        # single thread, like the reference (no rayon in its dependency tree)
    lg = 11
    n, N = 1 << lg, 1 << (lg + LOG_BLOWUP)
    seed = 0x5EED0000 + lg
    cols = [o.synth_column(seed, c, 0, n) for c in range(4)]
    coset, z, omega = _mont_small(5), _mont_small(0xC0FFEE), _root_of_unity_pallas(lg + LOG_BLOWUP)
    t0 = time.perf_counter()
    ext = [o.lde(0, c, LOG_BLOWUP, coset) for c in cols]
    f0, _ = o.ali_merge(ext[0], ext[1], ext[2], ext[3], omega, z, want_c_star=False)
    pr = o.deep_fri_prove(None, None, None, None, N, SCHEDULE, 1, SEED_Z, f0=f0)   # r = 1: the reference panics on an empty query set
    dt = time.perf_counter() - t0
    pr.free()
    return {"value": n / dt, "unit": "trace rows/s", "cores": 1, "kind": "port",
            "sample": f"2^{lg}-row trace x 4 columns (LDE to 2^{lg + LOG_BLOWUP}, merge, FRI build), {dt:.1f} s on one host core; C++ oracle, dense MDS as in the reference, constants hoisted"}


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — the hot path of saholmes/stark-mlwe on MI355X, BASELINE.json configs[1]:
"2^20 trace, blowup 8, single MI355X: NTT/LDE + Poseidon-Merkle kernels only".

One step = one pass of the hot path over one synthetic trace of 2^20 rows x 4 columns PER GPU
(a, s, e, t), inputs already resident in HBM:
  1. LDE of the 4 columns, 2^20 -> 2^23 evaluations on the coset 5*<w>;
  2. DEEP-ALI merge of the extended columns into f0 (fixed out-of-domain point z);
  3. fri_build_transcript(f0, [16,16,8]): 3 folds, leaf-pair Poseidon hashes of all layers and the
     4 Poseidon-Merkle trees (arity 16,16,8,2).
N = 1: the trace lives on the one GPU.  N > 1: ONE trace of N*2^20 rows is block-sharded over the ranks
(north_star's split): every LDE is a six-step NTT across the ranks (all-to-all transposes over RCCL), the
merge, folds, leaf hashes and lower Merkle levels are block-local with global indices, the tree tops are
all-gathered (stark_mlwe_amd.dist.ShardedTrace).  `value` = trace rows per second of that one job, work per
GPU fixed => "weak".

Side sections (outside the timed region, rank 0): HIP-event kernel timings for the roofline blocks (Fr-NTT
against HBM, the Poseidon leaf kernel against the live-measured integer MAC issue rate), end-to-end
`deep_fri_prove` at the reference's bench shape and at 2^20, and the CPU baseline (the C++ oracle — a port of
the reference's algorithm, the Rust reference cannot be built in this image — on a bounded sample).

Usage: python bench.py --gpus N --steps K --warmup W   (N>1: launched by torch.distributed.run)
       --steps-only   only the timed steps (what `rocprofv3 --kernel-trace --stats` should see)
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
STEP_COSET, STEP_Z = 5, 0xC0FFEE   # LDE coset shift (generator of Pallas Fr), fixed DEEP point of the kernels-only step
MFMA_I8_PEAK_MACS = 2.5e15      # dense int8 MAC/s: twice the bf16 rate (MI355X_MICROARCH.md: about 2.5 PFLOP/s bf16 dense = 1.25 PMAC/s)
FR_MULTS_T17, FR_MULTS_T9 = 21408, 5904   # reference-dense Fr-mults per permutation
P_PALLAS = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001


def leaf_kernel_macs():
    """Algorithmic multiply-accumulates of ONE leaf hash in kernel form.  Returns (valu, mfma):
    valu = 32x32+64 MACs on the vector ALU (v_mad_u64_u32), radix-2^29 arithmetic: a dot-product term is 9x9 = 81 MACs, a square 45, a
    Montgomery step 36 (nine digits x four non-trivial limbs of r);
    mfma = int8 MACs on the matrix cores: the six dense full-round products of a leaf hash run as v_mfma_i32_32x32x32_i8 over signed
    radix-256 digits — per product and sponge 17 outputs x 64 digit positions x (17 elements x 32 digits) MACs (half of them on the
    structural zeros of the Toeplitz operand: counted, they occupy the pipe).
    t = 17, RF = 8, RP = 64.  VALU share of a full round: 17 S-boxes + 17 Montgomery steps of the product's outputs; partial rounds in blocks of 4:
    2t-1 = 33 terms + 1.5 cross terms per round, 5 reductions per round, one S-box; the last full round squeezes row 0 only (17 terms, 2
    reductions); the leaf kernel's round 0 is a closed form: 2 S-boxes and 2 terms per lane instead of 17 S-boxes and a dense product."""
    t, rf, rp = 17, 8, 64
    sbox = 2 * 45 + 81 + 3 * 36
    full_valu = t * sbox + t * 36                       # rounds 1..6: the product itself is on the matrix cores
    last = t * sbox + t * 81 + 2 * 36                   # round 7: row 0 only
    part = sbox + (2 * t - 1 + 1.5) * 81 + 5 * 36
    round0 = 2 * sbox + 2 * t * 81 + t * 36
    valu = (rf - 2) * full_valu + last + round0 + rp * part
    mfma = (rf - 2) * t * 64 * (t * 32)
    return valu, mfma


def make_single_gpu_step(ctx, cols, log_n, dev):
    """The N = 1 step of this bench as a closure over device-resident trace columns `cols` (four [2^log_n, 4] int64 tensors):
    LDE of the four columns to 2^(log_n+3) points on the coset 5*<w>, DEEP-ALI merge at the fixed point z = 0xC0FFEE
    (crates/deep_ali/src/lib.rs:48-105), fri_build_transcript with [16,16,8] (crates/deep_ali/src/fri.rs:231-312).  Returns the L+1
    layer roots (Montgomery limbs).  tests/test_gpu_r3_step.py runs exactly this function against the oracle-generated golden roots."""
    import numpy as np
    import torch
    from stark_mlwe_amd.api import _ptr, PALLAS_FR
    lib = ctx.lib
    N = 1 << (log_n + LOG_BLOWUP)
    L = len(SCHEDULE)
    sched = np.ascontiguousarray(SCHEDULE, dtype=np.uint64)
    ext = [torch.empty((N, 4), dtype=torch.int64, device=dev) for _ in range(4)]
    f0 = torch.empty((N, 4), dtype=torch.int64, device=dev)
    omega = _root_of_unity_pallas(log_n + LOG_BLOWUP)
    coset = _mont_small(STEP_COSET)
    z = _mont_small(STEP_Z)         # fixed out-of-domain point for the kernels-only step (z^N != 1 checked by the library)

    def step():
        for c in range(4):   # LDE: interpolate on <w_n>, evaluate on 5*<w_N>
            ctx._chk(lib.stark_lde_dev(ctx.h, PALLAS_FR, C.c_void_p(cols[c].data_ptr()), log_n, LOG_BLOWUP, _ptr(coset), C.c_void_p(ext[c].data_ptr())))
        ctx._chk(lib.stark_ali_merge_dev(ctx.h, *[C.c_void_p(e.data_ptr()) for e in ext], None, None, _ptr(omega), _ptr(z), N, C.c_void_p(f0.data_ptr()), None))
        st = C.c_void_p()
        ctx._chk(lib.stark_fri_build_dev(ctx.h, C.c_void_p(f0.data_ptr()), N, _ptr(sched), L, SEED_Z, C.byref(st)))
        roots = []
        for l in range(L + 1):
            r = np.zeros(4, np.uint64); ctx._chk(lib.stark_fri_layer_root(st, l, _ptr(r))); roots.append(r)
        ctx._chk(lib.stark_fri_state_free(st))
        return roots
    return step


def roots_hex(roots):
    return ["".join(f"{int(x):016x}" for x in r[::-1]) for r in roots]


def golden_step_roots(log_n, seed):
    """Roots the CPU oracle produced ONCE for this step (tools/gen_golden.py step:K -> tests/golden/step_roots_kK.json), or None when
    no golden exists for this size / seed.  A committed data file: the oracle itself is not touched here."""
    path = os.path.join(ROOT, "tests", "golden", f"step_roots_k{log_n}.json")
    if not os.path.exists(path):
        return None
    g = json.load(open(path))
    if g.get("synth_seed") != seed or g.get("schedule") != SCHEDULE or g.get("seed_z") != SEED_Z or g.get("log_blowup") != LOG_BLOWUP or g.get("coset") != STEP_COSET or g.get("z") != STEP_Z:
        return None
    return g["roots"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-trace", type=int, default=LOG_TRACE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--steps-only", action="store_true", help="only the timed steps and the headline fields (for rocprofv3 kernel stats)")
    ap.add_argument("--e2e-log", type=int, default=20, help="log2 rows of the end-to-end deep_fri_prove section (0 disables)")
    ap.add_argument("--synth-seed", type=lambda x: int(x, 0), default=None, help="seed of the synthetic trace (default 0x5EED0000 + log-trace); lets a 1-GPU run reproduce the trace of an N-GPU run")
    ap.add_argument("--save-roots", default=None, help="write the step's roots (with the parameters that define the step) to this JSON file: a ONE-GPU run of a trace size that has no oracle golden, kept under tests/golden/step_roots_k<K>_one_gpu.json for the N > 1 runs of the same trace to compare with")
    ap.add_argument("--csv", default=None, help="also write the reference's benchmarkdata.csv schema (end_to_end.rs:42-44) for the reference-input proves")
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
    # ONE stream for torch and the library (include/stark_mlwe.h "Stream rule"): a dedicated torch stream is made current
    # and handed to the context, so torch-side slices / copies / collectives and library kernels are ordered without host syncs.
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

    # synthetic trace: columns 0..3; this rank holds rows [rank*n, (rank+1)*n) of the world*n-row trace (DESIGN.md "Synthetic inputs")
    log_total = log_n + (world.bit_length() - 1)     # the whole trace: 2^log_n rows per GPU
    seed = args.synth_seed if args.synth_seed is not None else 0x5EED0000 + log_total    # at N = 1 the seed of tests/golden/step_roots_k{log_n}.json
    cols = [dbuf(n) for _ in range(4)]
    for c in range(4):
        ctx._chk(lib.stark_synth_column_dev(ctx.h, seed, c, rank * n, n, C.c_void_p(cols[c].data_ptr())))
    coset = _mont_small(STEP_COSET)  # multiplicative generator of Pallas Fr as the LDE coset shift
    z = _mont_small(STEP_Z)          # fixed out-of-domain point for the kernels-only step (z^N != 1 checked by the library)

    if world == 1:
        step = make_single_gpu_step(ctx, cols, log_n, dev)
        sharding = "one GPU holds the whole trace"
    else:
        from stark_mlwe_amd import dist as sd
        if backend == "nccl" and os.environ.get("STARK_COMM", "lib") == "lib":
            # data-path collectives through the library's own RCCL communicator, on the shared stream (include/stark_mlwe.h
            # stark_comm_*); torch.distributed only carried the unique id and brackets the timed region.
            # Agree FIRST on whether every rank can bind RCCL (a local probe), then enter the collective init together: a rank that
            # failed inside LibComm() while its peers already sat in ncclCommInitRank would hang the job.
            flag = torch.tensor([1 if lib.stark_comm_available(None) == 0 else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            lc, comm_used = sd.checked_lib_comm(ctx, rank, world, dev, int(flag.item()) == 1)
            if lc is not None:
                sd.set_comm(lc)
            if rank == 0:
                sys.stderr.write(f"[bench] collectives: {comm_used}\n")
        job = sd.ShardedTrace(sd.HipProvider(ctx, device=dev), log_total, LOG_BLOWUP, SCHEDULE, SEED_Z, coset, z)

        def step():
            return job.step(cols)
        sharding = job.describe()

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
        "config": {"workload": f"one 2^{log_total}-row trace x 4 columns (2^{log_n} rows per GPU), blowup 8: LDE (iNTT + coset NTT) + DEEP-ALI merge + FRI folds + Poseidon leaf hashes + Poseidon-Merkle trees (schedule [16,16,8]), kernels only",
                   "log_trace_per_gpu": log_n, "log_blowup": LOG_BLOWUP, "schedule": SCHEDULE, "field": "pallas_fr", "sharding": sharding},
        "roots": roots_hex(roots),
    }
    gold = golden_step_roots(log_total, seed)      # N > 1: the sharded step commits to the same 2^log_total-row trace as one GPU would (same roots)
    # True / False against the committed oracle golden of exactly this step; null when no golden exists for this size, seed or N
    out["roots_match_golden"] = None if gold is None else (out["roots"] == gold)
    # product against product, where the CPU oracle is out of reach (2^23 rows = the N = 8 trace: about 6 h of the build container): the roots a ONE-GPU run
    # of this very trace committed to (tests/golden/step_roots_k<K>_one_gpu.json, written by --save-roots); null when there is none or N = 1
    one = None
    if world > 1:
        pth = os.path.join(ROOT, "tests", "golden", f"step_roots_k{log_total}_one_gpu.json")
        if os.path.exists(pth):
            g1 = json.load(open(pth))
            if g1.get("synth_seed") == seed and g1.get("schedule") == SCHEDULE and g1.get("seed_z") == SEED_Z and g1.get("log_blowup") == LOG_BLOWUP and g1.get("coset") == STEP_COSET and g1.get("z") == STEP_Z:
                one = out["roots"] == g1["roots"]
    out["roots_match_one_gpu_run"] = one
    if args.save_roots and rank == 0 and world == 1:
        with open(args.save_roots, "w") as f:
            json.dump({"log_trace": log_total, "log_blowup": LOG_BLOWUP, "schedule": SCHEDULE, "seed_z": SEED_Z, "synth_seed": seed, "coset": STEP_COSET, "z": STEP_Z,
                       "roots": out["roots"], "generator": "bench.py --save-roots on ONE MI355X (the product, not the oracle)"}, f, indent=1)
            f.write("\n")
    if args.steps_only:
        if rank == 0:
            print(json.dumps(out), flush=True)
        ctx.close()
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- kernel-level measurement (HIP events on the context's stream, outside the timed region) ---------------------------
    ms = C.c_float()

    def timed(fn, reps):
        fn()
        ctx._chk(lib.stark_timer_start(ctx.h))
        for _ in range(reps):
            fn()
        ctx._chk(lib.stark_timer_stop_ms(ctx.h, C.byref(ms)))
        return ms.value / reps

    # (1) the HBM-streaming kernel group of the headline metric: one 2^23 coset NTT (3 launches over the whole vector)
    scratch = dbuf(N)
    ctx._chk(lib.stark_synth_column_dev(ctx.h, seed, 7, 0, N, C.c_void_p(scratch.data_ptr())))
    ntt_ms = timed(lambda: ctx._chk(lib.stark_ntt_dev(ctx.h, PALLAS_FR, C.c_void_p(scratch.data_ptr()), log_n + LOG_BLOWUP, 0, _ptr(coset))), 5)
    ntt_bytes = 64.0 * N
    ntt_gbps = ntt_bytes / (ntt_ms * 1e-3) / 1e9
    lde_out = dbuf(N)
    lde_ms = timed(lambda: ctx._chk(lib.stark_lde_dev(ctx.h, PALLAS_FR, C.c_void_p(cols[0].data_ptr()), log_n, LOG_BLOWUP, _ptr(coset), C.c_void_p(lde_out.data_ptr()))), 5)
    # (2) the dominant kernel of the step: the Poseidon leaf hash, one launch over 2^23 leaves
    h = dbuf(N); fnext = dbuf(N // 16)
    ctx._chk(lib.stark_synth_column_dev(ctx.h, seed, 6, 0, N // 16, C.c_void_p(fnext.data_ptr())))
    leaf_ms = timed(lambda: ctx._chk(lib.stark_leaf_pair_hash_dev(ctx.h, None, C.c_void_p(scratch.data_ptr()), C.c_void_p(fnext.data_ptr()), N, 16, C.c_void_p(h.data_ptr()))), 1)
    mac_rate = C.c_double()
    ctx._chk(lib.stark_diag_mac_rate(ctx.h, C.byref(mac_rate)))
    del scratch, h, fnext, lde_out

    # ---- the reference's own bench shape (deep_fri_prove, schedule [16,16,8], r = 32; end_to_end.rs:187-270) ----
    def prove(log_k, given_f0):
        nk = 1 << log_k
        cs = [dbuf(nk) for _ in range(4)]
        for c in range(4):
            ctx._chk(lib.stark_synth_column_dev(ctx.h, 0x5EED0000 + log_k, c, 0, nk, C.c_void_p(cs[c].data_ptr())))
        ph = C.c_void_p()
        a_ = [C.c_void_p(x.data_ptr()) for x in cs]
        t0 = time.perf_counter()
        ctx._chk(lib.stark_deep_fri_prove_dev(ctx.h, *([None] * 4 + [a_[0]] if given_f0 else a_ + [None]), nk, _ptr(sched), L, 32, SEED_Z, C.byref(ph)))
        wall = (time.perf_counter() - t0) * 1e3
        res = {"log_n0": log_k, "r": 32, "wall_ms": wall, "proof_bytes": int(lib.stark_proof_len(ph)), "size_estimate": int(lib.stark_proof_size_estimate(ph)),
               "build_f0_ms": lib.stark_proof_stage_ms(ph, 0), "fri_build_ms": lib.stark_proof_stage_ms(ph, 1), "queries_encode_ms": lib.stark_proof_stage_ms(ph, 2)}
        if not given_f0:
            res["sponge_share"] = res["build_f0_ms"] / wall
            res["us_per_dependent_permutation"] = res["build_f0_ms"] * 1e3 / (nk / 16 + 2)
        lib.stark_proof_free(ph)
        return res
    sections = {}
    if rank == 0:
        prove(12, False)                                   # warm the constants / plans
        sections["prove_end_to_end_2^16"] = dict(prove(16, False), note="deep_fri_prove with DeepAliRealBuilder, device-resident columns; the reference published 57 143 ms for this size (benchmarkdata.csv:7)")
        if args.e2e_log:
            sections[f"prove_end_to_end_2^{args.e2e_log}"] = dict(prove(args.e2e_log, False), note="build_f0 = the four serial column sponges (fri.rs:548-557), n0/16 dependent permutations each, one wave per column")
        sections["prove_given_f0_2^20"] = dict(prove(20, True), note="deep_fri_prove stages after build_f0 on n0 = 2^20")
        sections["serial_sponge"] = sponge_section(ctx, torch, dev)
        sections["reference_bench"] = reference_bench(ctx, np, args.csv, world)
        sections["reference_bench_presets"] = preset_bench(ctx, np, torch, dev, args.csv, world)
        sections["reference_bench_sumcheck"] = sumcheck_bench(ctx, np, torch, dev)

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "ntt_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"ntt_2^{log_n + LOG_BLOWUP}_bytes_per_transform")
            except Exception:
                traffic = None
        macs_leaf, mfma_leaf = leaf_kernel_macs()
        leaves_per_s = N / (leaf_ms * 1e-3)
        out.update({
            "roofline": {"bound": "hbm", "kernel": f"Fr-NTT 2^{log_n + LOG_BLOWUP} coset forward (k_ntt_strided x2 + k_ntt_last)", "achieved": ntt_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ntt_gbps / HBM_PEAK_GBPS, "traffic": traffic, "algorithmic_bytes": ntt_bytes, "avg_ms": ntt_ms,
                         "lde_2^%d_to_2^%d_ms_per_column" % (log_n, log_n + LOG_BLOWUP): lde_ms,
                         # what actually binds a 255-bit NTT on this chip: integer multiply-accumulate issue, not HBM (VALU busy 90 %, counters in
                         # profiles/*_sq_counters.json).  Algorithmic MACs per element: (log2 n)/2 butterflies + 3 table products (coset pre-scale, two inter-pass
                         # twiddles), each one carry-free 9x9 product + one Montgomery step by 2^261 = 81 + 36 v_mad_u64_u32.
                         "valu": {"unit": "lane-MAC/s (v_mad_u64_u32)", "macs_per_element": ((log_n + LOG_BLOWUP) / 2 + 3) * 117, "achieved": ((log_n + LOG_BLOWUP) / 2 + 3) * 117 * N / (ntt_ms * 1e-3),
                                  "peak": mac_rate.value, "frac": ((log_n + LOG_BLOWUP) / 2 + 3) * 117 * N / (ntt_ms * 1e-3) / mac_rate.value, "peak_source": "stark_diag_mac_rate, measured live on this device"}},
            "poseidon": {"kernel": "k_leaf_pair2 (t=17), 2^%d leaves" % (log_n + LOG_BLOWUP), "ms": leaf_ms, "leaves_per_s": leaves_per_s,
                         "reference_dense_fr_mults_per_s": FR_MULTS_T17 * leaves_per_s,
                         "roofline": {"bound": "int-valu", "unit": "lane-MAC/s (v_mad_u64_u32)", "peak": mac_rate.value, "peak_source": "stark_diag_mac_rate, measured live on this device",
                                      "macs_per_leaf": macs_leaf, "achieved": macs_leaf * leaves_per_s, "frac": macs_leaf * leaves_per_s / mac_rate.value,
                                      "share_of_step": leaf_ms / ms_per_step if world == 1 else None,
                                      "note": "the vector-ALU part of the kernel; its six dense full-round products run on the matrix cores (mfma_i8 below) and overlap with it",
                                      "mfma_i8": {"unit": "int8 MAC/s (v_mfma_i32_32x32x32_i8)", "macs_per_leaf": mfma_leaf, "achieved": mfma_leaf * leaves_per_s,
                                                  "peak": MFMA_I8_PEAK_MACS, "frac": mfma_leaf * leaves_per_s / MFMA_I8_PEAK_MACS,
                                                  "peak_source": "MI355X_MICROARCH.md: dense int8 = 2x the bf16 rate (about 2.5 PFLOP/s bf16 = 1.25 PMAC/s)"}}},
        })
        out.update(sections)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(np, log_n)
        print(json.dumps(out), flush=True)
    barrier()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def sponge_section(ctx, torch, dev):
    """The serial column sponge of build_f0 (crates/deep_ali/src/fri.rs:548-557) on its own: one tr_hash_fields_tagged over 2^14 fields (1 026
    dependent t = 17 permutations), five-wave kernel (poseidon_chain.hpp) and the round-2 one-wave kernel (option sponge_one_wave); equal digests."""
    lib = ctx.lib
    n = 1 << 14
    col = torch.empty((n, 4), dtype=torch.int64, device=dev)
    ctx._chk(lib.stark_synth_column_dev(ctx.h, 0x5EED0000 + 14, 0, 0, n, C.c_void_p(col.data_ptr())))
    res, dig = {}, {}
    for name, opt in (("five_wave", 0), ("one_wave", 1)):
        ctx._chk(lib.stark_ctx_set_option(ctx.h, b"sponge_one_wave", opt))
        out = torch.zeros((1, 4), dtype=torch.int64, device=dev)
        for _ in range(2):
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            ctx._chk(lib.stark_tr_hash_fields_tagged_dev(ctx.h, None, b"ALI/A", C.c_void_p(col.data_ptr()), n, 1, C.c_void_p(out.data_ptr())))
            torch.cuda.synchronize(dev); dt = time.perf_counter() - t0
        res[name + "_us_per_permutation"] = dt * 1e6 / (n // 16 + 2); dig[name] = out.cpu().numpy().tobytes()
    ctx._chk(lib.stark_ctx_set_option(ctx.h, b"sponge_one_wave", 0))
    res["equal_digests"] = dig["five_wave"] == dig["one_wave"]
    res["note"] = "one workgroup of five waves per chain: the dependent products in row form (16 lanes per product), the accumulators on two helper waves, the full rounds' S-boxes in row form on all five; floor of the chain 64 x 3 x 233 ns = 45 us"
    return res


PUBLISHED_PROOF_BYTES = {11: 39592, 12: 52000, 13: 60968, 14: 72936, 15: 87736, 16: 101976, 17: 119952, 18: 140032}   # crates/channel/benchmarkdata.csv:2-9


def reference_bench(ctx, np, csv_path, gpus):
    """The reference's own bench (channel/benches/end_to_end.rs:187-374, preset "paper" = [16,16,8], r = 32, seed_z = 0xDEEFBAAD) on its
    own inputs (seed chain from 1337, one LCG step per k from 11; StdRng + Fp::rand through stark_ref_bench_inputs): prove and verify
    through the C-ABI; `proof_bytes` is deep_fri_proof_size_bytes and must equal what the reference published."""
    from stark_mlwe_amd.api import DeepFriParams, ref_bench_inputs
    prm = DeepFriParams(SCHEDULE, 32, SEED_Z)
    rows, seed = [], 1337
    for k in range(11, 17):
        seed = (seed * 1103515245 + 12345) % 2**64
        cols = ref_bench_inputs(seed, 1 << k, 4)
        t0 = time.perf_counter()
        proof, est, ms = ctx.deep_fri_prove(cols[0], cols[1], cols[2], cols[3], 1 << k, prm)
        prove_s = time.perf_counter() - t0
        t0 = time.perf_counter(); ok = ctx.deep_fri_verify(prm, proof); verify_ms = (time.perf_counter() - t0) * 1e3
        rows.append({"label": "paper", "k": k, "schedule": SCHEDULE, "proof_bytes": est, "published_proof_bytes": PUBLISHED_PROOF_BYTES[k], "matches_published": est == PUBLISHED_PROOF_BYTES[k],
                     "prove_s": prove_s, "verify_ms": verify_ms, "verified": bool(ok), "prove_elems_per_s": (1 << k) / prove_s, "encoded_bytes": len(proof),
                     "build_f0_ms": ms[0], "fri_build_ms": ms[1], "queries_encode_ms": ms[2]})
    if csv_path:
        with open(csv_path, "w") as f:
            f.write("csv,label,k,schedule,proof_bytes,prove_s,verify_ms,prove_elems_per_s,delta_size_pct_vs_paper,delta_prove_pct_vs_paper,delta_verify_pct_vs_paper,delta_throughput_pct_vs_paper,gpus,build_f0_ms,fri_build_ms,queries_encode_ms,published_proof_bytes\n")
            for r in rows:
                f.write("csv,%s,%d,[%s],%d,%.6f,%.3f,%.6f,0.00,0.00,0.00,0.00,%d,%.3f,%.3f,%.3f,%d\n" % (r["label"], r["k"], ",".join(str(m) for m in r["schedule"]), r["proof_bytes"], r["prove_s"], r["verify_ms"],
                                                                                                   r["prove_elems_per_s"], gpus, r["build_f0_ms"], r["fri_build_ms"], r["queries_encode_ms"], r["published_proof_bytes"]))
    return {"rows": rows, "all_match_published": all(r["matches_published"] and r["verified"] for r in rows),
            "note": "reference published prove 1.85 s / verify 103 ms at k = 11 ... 57.1 s / 212 ms at k = 16 (Apple arm64, 1 thread; benchmarkdata.csv:2-7)"}


# The reference bench's other schedules (channel/benches/end_to_end.rs:195-201) and the Criterion means its authors left under
# target/criterion/e2e_mf_fri/{prove,verify}-<label>/<k>/new/estimates.json (Apple arm64, one thread), ms.
PRESETS = [("mod16", [16, 16, 16, 16]), ("uni32x3", [32, 32, 32]), ("uni64x2x8", [64, 64, 8]), ("hi64_32_8", [64, 32, 8]), ("hi32_32_16", [32, 32, 16]),
           # the 128-fold schedules (end_to_end.rs:202-210; layers of arity 128 = Poseidon width 129): no published timings, measured at k = 16
           ("uni128", [128]), ("uni128x2", [128, 128]), ("hi128_64", [128, 64]), ("hi128_32", [128, 32]), ("hi128_16", [128, 16]), ("hi128_64_8", [128, 64, 8]), ("hi128_32_8", [128, 32, 8])]
PUBLISHED_PRESET_MS = {("mod16", 16): (55906.8, 240.5), ("uni32x3", 15): (27146.9, 162.1), ("uni32x3", 16): (54165.2, 195.1), ("uni64x2x8", 15): (26999.7, 135.5),
                       ("uni64x2x8", 16): (53552.6, 166.4), ("hi64_32_8", 14): (13454.9, 130.8), ("hi64_32_8", 15): (27069.2, 150.0), ("hi64_32_8", 16): (53520.7, 183.3),
                       ("hi32_32_16", 14): (13647.6, 148.4), ("hi32_32_16", 15): (27093.2, 169.0), ("hi32_32_16", 16): (54904.8, 196.2)}


# The reference's sum-check benches (`e2e_plain` prove / verify at k = 12, 14, 16; `e2e_mf` at k = 12, 14, two queries per round;
# Criterion means, Apple arm64, 1 thread) — N4, the second consumer of the Merkle kernels (crates/channel/src/lib.rs:1045-1240).
PUBLISHED_SUMCHECK_MS = {("plain", 12): (148.3, 4.05), ("plain", 14): (585.9, 4.59), ("plain", 16): (2363.3, 5.13), ("mf", 12): (323.4, 25.8), ("mf", 14): (1207.0, 34.1)}


def sumcheck_bench(ctx, np, torch, dev):
    """prove_plain / prove_mf / verify_* through the C-ABI on a synthetic 2^k-entry witness (host buffers, as the reference's API takes them)."""
    lib = ctx.lib; rows = []
    for (variant, k), (pub_p, pub_v) in sorted(PUBLISHED_SUMCHECK_MS.items()):
        wdev = torch.empty((1 << k, 4), dtype=torch.int64, device=dev)
        ctx._chk(lib.stark_synth_column_dev(ctx.h, 0x5C0000 + k, 0, 0, 1 << k, C.c_void_p(wdev.data_ptr())))
        w = wdev.cpu().numpy().view(np.uint64)
        prove = (lambda: ctx.prove_plain(k, 2025, w)) if variant == "plain" else (lambda: ctx.prove_mf(k, 2025, 2, w))
        verify = (lambda pr: ctx.verify_plain(k, 2025, pr)) if variant == "plain" else (lambda pr: ctx.verify_mf(k, 2025, 2, pr))
        prove()                                                        # warm: parameters, plans
        best_p, best_v, ok, proof = 1e9, 1e9, True, b""
        for _ in range(3):
            t0 = time.perf_counter(); proof = prove(); best_p = min(best_p, time.perf_counter() - t0)
            t0 = time.perf_counter(); ok = verify(proof) and ok; best_v = min(best_v, time.perf_counter() - t0)
        bad = bytearray(proof); bad[len(bad) // 2] ^= 1
        rows.append({"variant": variant, "k": k, "prove_ms": best_p * 1e3, "verify_ms": best_v * 1e3, "proof_bytes": len(proof), "verified": bool(ok),
                     "tampered_rejected": not verify(bytes(bad)), "published_prove_ms": pub_p, "published_verify_ms": pub_v})
    return {"rows": rows, "all_verified": all(r["verified"] and r["tampered_rejected"] for r in rows),
            "note": "synthetic witness from host memory (upload included); published_* = the reference authors' Criterion means (Apple arm64, 1 thread; SURVEY.md §6(b)); "
                    "proof bytes are compared with the oracle's in tests/test_gpu_r2_sumcheck.py (parity unpinned: the reference holds no vectors for this path)"}


def preset_bench(ctx, np, torch, dev, csv_path, gpus):
    """deep_fri_prove / deep_fri_verify for the reference bench's presets other than "paper" at the sizes its authors measured (k_min .. 16; the 128-fold
    schedules, for which they left no timings, at k = 16): layers of arity 32 / 64 / 128 (Poseidon widths t = 33 / 65 / 129, one wave per node).  Inputs: the synthetic trace (seed 0x5EED0000 + k) — the reference's seed chain for
    these presets depends on how many ks its "paper" loop ran, which the repo does not record, and it published no proof sizes for them.
    Per row: end-to-end prove from (a, s, e, t), the stages after build_f0 alone (`prove_given_f0_ms`), verify."""
    from stark_mlwe_amd.api import _ptr
    lib = ctx.lib
    rows = []
    for label, sched_l in PRESETS:
        sch = np.ascontiguousarray(sched_l, dtype=np.uint64)
        kmin = sum(int(m).bit_length() - 1 for m in sched_l)
        for k in range(max(14, kmin) if 128 not in sched_l else 16, 17):
            nk = 1 << k
            cs = [torch.empty((nk, 4), dtype=torch.int64, device=dev) for _ in range(4)]
            for c in range(4):
                ctx._chk(lib.stark_synth_column_dev(ctx.h, 0x5EED0000 + k, c, 0, nk, C.c_void_p(cs[c].data_ptr())))
            ptr = [C.c_void_p(x.data_ptr()) for x in cs]

            def run(given_f0):
                ph = C.c_void_p()
                t0 = time.perf_counter()
                ctx._chk(lib.stark_deep_fri_prove_dev(ctx.h, *([None] * 4 + [ptr[0]] if given_f0 else ptr + [None]), nk, _ptr(sch), len(sched_l), 32, SEED_Z, C.byref(ph)))
                wall = (time.perf_counter() - t0) * 1e3
                ms = [lib.stark_proof_stage_ms(ph, i) for i in range(3)]
                proof, est = ctx._proof_out(ph)
                return wall, ms, proof, est
            run(True)                                    # warm this schedule's parameter sets (t = 33 / 65) and plans
            wall_f0, ms_f0, _, _ = run(True)
            wall, ms, proof, est = run(False)
            prm = __import__("stark_mlwe_amd.api", fromlist=["DeepFriParams"]).DeepFriParams(sched_l, 32, SEED_Z)
            t0 = time.perf_counter(); ok = ctx.deep_fri_verify(prm, proof); verify_ms = (time.perf_counter() - t0) * 1e3
            pub = PUBLISHED_PRESET_MS.get((label, k), (None, None))
            rows.append({"label": label, "k": k, "schedule": sched_l, "proof_bytes": est, "encoded_bytes": len(proof), "prove_s": wall / 1e3, "verify_ms": verify_ms, "verified": bool(ok),
                         "build_f0_ms": ms[0], "fri_build_ms": ms[1], "queries_encode_ms": ms[2], "prove_given_f0_ms": wall_f0, "given_f0_fri_build_ms": ms_f0[1], "given_f0_queries_encode_ms": ms_f0[2],
                         "prove_elems_per_s": nk / (wall / 1e3), "published_prove_ms": pub[0], "published_verify_ms": pub[1]})
            del cs
    if csv_path:
        with open(csv_path, "a") as f:
            for r in rows:
                f.write("csv,%s,%d,[%s],%d,%.6f,%.3f,%.6f,,,,,%d,%.3f,%.3f,%.3f,\n" % (r["label"], r["k"], ",".join(str(m) for m in r["schedule"]), r["proof_bytes"], r["prove_s"], r["verify_ms"],
                                                                                     r["prove_elems_per_s"], gpus, r["build_f0_ms"], r["fri_build_ms"], r["queries_encode_ms"]))
    return {"rows": rows, "all_verified": all(r["verified"] for r in rows),
            "note": "synthetic inputs; published_* = the reference authors' Criterion means (Apple arm64, 1 thread; SURVEY.md §6(b)); prove is bounded by the serial column sponges whatever the schedule, prove_given_f0_ms is what the schedule changes"}


def _mont_small(x):
    """Montgomery limbs of a small integer in Pallas Fr (host-side scalar; mirrors F::from(u64))."""
    import numpy as np
    m = (x << 256) % P_PALLAS
    return np.array([(m >> (64 * i)) & (2**64 - 1) for i in range(4)], np.uint64)


def _root_of_unity_pallas(log_n):
    import numpy as np
    w = pow(5, (P_PALLAS - 1) >> 32, P_PALLAS)
    for _ in range(32 - log_n):
        w = w * w % P_PALLAS
    m = (w << 256) % P_PALLAS
    return np.array([(m >> (64 * i)) & (2**64 - 1) for i in range(4)], np.uint64)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _usable_cores():
    """Host threads this process may really use: the affinity mask, capped by the cgroup CPU quota of the box's share."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(np, log_n_gpu):
    """The oracle (CPU port of the reference's algorithm, dense MDS as in the reference, constants hoisted) on bounded samples
    of the same work, on this box's host cores: single thread (the reference is single-threaded) and all cores (OpenMP over
    leaves / nodes / butterflies).  `value` = the bench workload (LDE + merge + FRI build) at 2^11 rows, one thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    o = oracle_lib.Oracle()
    ncores = _usable_cores()
    res = {"unit": "trace rows/s", "kind": "port", "cpu_model": _cpu_model(), "nproc": os.cpu_count(), "threads_all_cores_mode": ncores, "compiler": "g++ -O2 -fopenmp (oracle/Makefile)"}

    def workload(lg, threads):
        o.l.oracle_set_threads(threads)
        n, N = 1 << lg, 1 << (lg + LOG_BLOWUP)
        cols = [o.synth_column(0x5EED0000 + lg, c, 0, n) for c in range(4)]
        coset, z, omega = _mont_small(5), _mont_small(0xC0FFEE), _root_of_unity_pallas(lg + LOG_BLOWUP)
        t0 = time.perf_counter()
        ext = [o.lde(0, c, LOG_BLOWUP, coset) for c in cols]
        t1 = time.perf_counter()
        f0, _ = o.ali_merge(ext[0], ext[1], ext[2], ext[3], omega, z, want_c_star=False)
        t2 = time.perf_counter()
        pr = o.deep_fri_prove(None, None, None, None, N, SCHEDULE, 1, SEED_Z, f0=f0)   # r = 1: the reference panics on an empty query set
        t3 = time.perf_counter()
        pr.free()
        return {"rows_per_s": n / (t3 - t0), "seconds": t3 - t0, "lde_s": t1 - t0, "merge_s": t2 - t1, "fri_build_s": t3 - t2}

    def prove(lg, threads):
        o.l.oracle_set_threads(threads)
        n0 = 1 << lg
        cols = [o.synth_column(0x5EED0000 + lg, c, 0, n0) for c in range(4)]
        t0 = time.perf_counter()
        pr = o.deep_fri_prove(cols[0], cols[1], cols[2], cols[3], n0, SCHEDULE, 32, SEED_Z)
        dt = time.perf_counter() - t0
        r = {"log_n0": lg, "threads": threads, "prove_s": dt, "build_f0_s": pr.secs(0), "fri_build_and_queries_s": pr.secs(1), "rows_per_s": n0 / dt}
        pr.free()
        return r

    def ntt(lg, threads):
        o.l.oracle_set_threads(threads)
        x = o.synth_column(1, 7, 0, 1 << lg)
        t0 = time.perf_counter(); o.ntt(0, x); dt = time.perf_counter() - t0
        return {"log_n": lg, "threads": threads, "ms": dt * 1e3, "GBps_algorithmic": 64.0 * (1 << lg) / dt / 1e9}

    one = workload(11, 1)
    res.update({"value": one["rows_per_s"], "cores": 1,
                "sample": f"2^11-row trace x 4 columns (LDE to 2^14, merge, FRI build), {one['seconds']:.1f} s on one host core",
                "workload_1t": one, "workload_all_cores": dict(workload(11, ncores), cores=ncores)})
    if ncores >= 8:   # a larger sample of the same workload on all cores (about 8x the work of the one-thread sample): the CPU side at a size nearer the GPU's
        res["workload_all_cores_2^14"] = dict(workload(14, ncores), cores=ncores, rows=1 << 14)
    res["prove"] = [prove(12, 1), prove(12, ncores), prove(16, ncores)]
    res["ntt"] = [ntt(20, 1), ntt(20, ncores), ntt(24, ncores)]
    # the serial column sponge on ONE host core (it does not parallelise): dense rounds as in the reference (oracle) and the
    # sparse kernel-form rounds the GPU runs, instantiated on the host (libstark_mlwe_hostcheck.so: a diagnostic build of the
    # product's own inline code, not the oracle)
    o.l.oracle_set_threads(1)
    col = o.synth_column(3, 0, 0, 1 << 12)
    t0 = time.perf_counter(); o.tr_hash_fields_tagged(b"ALI/A", col); dt = time.perf_counter() - t0
    sponge = {"permutations": (1 << 12) // 16 + 2, "oracle_dense_us_per_permutation": dt * 1e6 / ((1 << 12) // 16 + 2)}
    try:
        import hostcheck_lib
        hc = hostcheck_lib.HostCheck(); tp = hc.params(1)
        t0 = time.perf_counter(); hc.tr_hash(tp, b"ALI/A", col); dt = time.perf_counter() - t0
        sponge["host_sparse_kernel_form_us_per_permutation"] = dt * 1e6 / ((1 << 12) // 16 + 2)
        hc.params_free(tp)
    except Exception as ex:   # noqa: BLE001
        sponge["host_sparse_error"] = repr(ex)[:200]
    res["sponge"] = sponge
    o.l.oracle_set_threads(ncores)
    return res


if __name__ == "__main__":
    main()

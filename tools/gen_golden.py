#!/usr/bin/env python3
"""tools/gen_golden.py — golden proof hashes for the end-to-end GPU tests (TEST INFRASTRUCTURE).

Runs the CPU oracle's `deep_fri_prove` (DeepAliRealBuilder: the four serial column sponges, the
FS challenges, the DEEP-ALI merge, all folds / leaf hashes / trees, the query phase and the canonical
encoding — oracle/fri.hpp restating crates/deep_ali/src/fri.rs:535-641) on the synthetic trace of
DESIGN.md "Synthetic inputs" (seed 0x5EED0000 + k, columns 0..3) and records

    k, r, schedule, seed_z, proof length, deep_fri_proof_size_bytes, sha256(proof bytes)

as one JSON file per case under tests/golden/.  The GPU tests (`tests/test_gpu_parity.py`) run the
same inputs through `stark_deep_fri_prove_dev` and compare the digest: BASELINE.json configs[2]
("2^22 trace full FRI (40 queries) ... end-to-end proof bytes bit-exact vs CPU") at its stated size.

Run ONCE in the build container (all host cores; k = 22 takes about a quarter of an hour):
    python tools/gen_golden.py 16:32 20:32 22:40

`step:K` records the roots of bench.py's kernels-only step (BASELINE.json configs[1]) for a 2^K-row trace: LDE of the four
synthetic columns to 2^(K+3) points on the coset 5*<w> (oracle/ntt.hpp), DEEP-ALI merge at the fixed point z = 0xC0FFEE
(oracle/fri.hpp, crates/deep_ali/src/lib.rs:48-105), fri_build_transcript with [16,16,8] (crates/deep_ali/src/fri.rs:231-312)
-> tests/golden/step_roots_k{K}.json.  `step:20` is the bench size (about half an hour of all host cores).

`preset:LABEL:K` records the proof of one of the reference bench's other schedules (channel/benches/end_to_end.rs:195-201) at 2^K rows, r = 32
-> tests/golden/proof_k{K}_r32_{LABEL}.json (trees of arity 32 / 64 = Poseidon widths 33 / 65).
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCHEDULE = [16, 16, 8]
SEED_Z = 0xDEEFBAAD


PRESETS = {"paper": [16, 16, 8], "mod16": [16, 16, 16, 16], "uni32x3": [32, 32, 32], "uni64x2x8": [64, 64, 8], "hi64_32_8": [64, 32, 8], "hi32_32_16": [32, 32, 16]}
P_PALLAS = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
LOG_BLOWUP = 3
STEP_Z = 0xC0FFEE
STEP_COSET = 5


def _mont(x):
    import numpy as np
    m = (x << 256) % P_PALLAS
    return np.array([(m >> (64 * i)) & (2**64 - 1) for i in range(4)], np.uint64)


def step_roots(o, k):
    """The oracle's roots of bench.py's step on the synthetic 2^k-row trace (seed 0x5EED0000 + k)."""
    n, N = 1 << k, 1 << (k + LOG_BLOWUP)
    t0 = time.time()
    cols = [o.synth_column(0x5EED0000 + k, c, 0, n) for c in range(4)]
    ext = [o.lde(0, c, LOG_BLOWUP, _mont(STEP_COSET)) for c in cols]
    omega = o.root_of_unity(k + LOG_BLOWUP)
    f0, _ = o.ali_merge(ext[0], ext[1], ext[2], ext[3], omega, _mont(STEP_Z), want_c_star=False)
    f0_sha = hashlib.sha256(f0.tobytes()).hexdigest()
    del ext
    pr = o.deep_fri_prove(None, None, None, None, N, SCHEDULE, 1, SEED_Z, f0=f0)   # r = 1: the reference panics on an empty query set
    rec = {"log_trace": k, "log_blowup": LOG_BLOWUP, "schedule": SCHEDULE, "seed_z": SEED_Z, "synth_seed": 0x5EED0000 + k,
           "coset": STEP_COSET, "z": STEP_Z, "f0_sha256": f0_sha,
           "roots": ["".join(f"{int(x):016x}" for x in pr.root(l)[::-1]) for l in range(pr.num_layers())],
           "generator": "tools/gen_golden.py step:%d (oracle lde + ali_merge + fri_build_transcript, all host cores)" % k,
           "oracle_seconds": round(time.time() - t0, 1)}
    pr.free()
    return rec


def main():
    import oracle_lib
    o = oracle_lib.Oracle()
    o.l.oracle_set_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count() or 1)))
    for spec in sys.argv[1:]:
        if spec.startswith("step:"):
            k = int(spec.split(":")[1])
            rec = step_roots(o, k)
            path = os.path.join(ROOT, "tests", "golden", f"step_roots_k{k}.json")
            with open(path, "w") as f:
                json.dump(rec, f, indent=1)
                f.write("\n")
            print(path, rec["roots"], f"{rec['oracle_seconds']} s", flush=True)
            continue
        schedule, label = SCHEDULE, None
        if spec.startswith("preset:"):
            _, label, kk = spec.split(":"); schedule = PRESETS[label]; spec = f"{kk}:32"
        k, r = (int(x) for x in spec.split(":"))
        n0 = 1 << k
        t0 = time.time()
        cols = [o.synth_column(0x5EED0000 + k, c, 0, n0) for c in range(4)]
        pr = o.deep_fri_prove(cols[0], cols[1], cols[2], cols[3], n0, schedule, r, SEED_Z)
        b = pr.bytes()
        rec = {"log_n0": k, "r": r, "schedule": schedule, "seed_z": SEED_Z, "synth_seed": 0x5EED0000 + k,
               "proof_len": len(b), "size_estimate": pr.size_estimate(), "sha256": hashlib.sha256(b).hexdigest(),
               "roots": ["".join(f"{int(x):016x}" for x in pr.root(l)[::-1]) for l in range(pr.num_layers())],
               "generator": "tools/gen_golden.py (oracle/fri.hpp deep_fri_prove, all host cores)", "oracle_seconds": round(time.time() - t0, 1)}
        pr.free()
        path = os.path.join(ROOT, "tests", "golden", f"proof_k{k}_r{r}.json" if label is None else f"proof_k{k}_r{r}_{label}.json")
        with open(path, "w") as f:
            json.dump(rec, f, indent=1)
            f.write("\n")
        print(path, rec["sha256"], rec["proof_len"], rec["size_estimate"], f"{rec['oracle_seconds']} s", flush=True)


if __name__ == "__main__":
    main()

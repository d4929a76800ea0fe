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


def main():
    import oracle_lib
    o = oracle_lib.Oracle()
    o.l.oracle_set_threads(os.cpu_count() or 1)
    for spec in sys.argv[1:]:
        k, r = (int(x) for x in spec.split(":"))
        n0 = 1 << k
        t0 = time.time()
        cols = [o.synth_column(0x5EED0000 + k, c, 0, n0) for c in range(4)]
        pr = o.deep_fri_prove(cols[0], cols[1], cols[2], cols[3], n0, SCHEDULE, r, SEED_Z)
        b = pr.bytes()
        rec = {"log_n0": k, "r": r, "schedule": SCHEDULE, "seed_z": SEED_Z, "synth_seed": 0x5EED0000 + k,
               "proof_len": len(b), "size_estimate": pr.size_estimate(), "sha256": hashlib.sha256(b).hexdigest(),
               "roots": ["".join(f"{int(x):016x}" for x in pr.root(l)[::-1]) for l in range(pr.num_layers())],
               "generator": "tools/gen_golden.py (oracle/fri.hpp deep_fri_prove, all host cores)", "oracle_seconds": round(time.time() - t0, 1)}
        pr.free()
        path = os.path.join(ROOT, "tests", "golden", f"proof_k{k}_r{r}.json")
        with open(path, "w") as f:
            json.dump(rec, f, indent=1)
            f.write("\n")
        print(path, rec["sha256"], rec["proof_len"], rec["size_estimate"], f"{rec['oracle_seconds']} s", flush=True)


if __name__ == "__main__":
    main()

"""The PRODUCT's verifier logic (stark_mlwe_amd/csrc/fri_verify.hpp: decoder, deep_fri_verify, verify_many_ds / verify_pairs_ds —
written from crates/deep_ali/src/fri.rs:643-762 and crates/merkle/src/lib.rs:587-773) on the CPU, with the hashes computed by the
host instantiation of the kernel bodies (libstark_mlwe_hostcheck.so).  Checked against the oracle's independent restatement of
the same functions: accept on honest proofs, and the SAME accept/reject decision on hundreds of tampered byte strings.
The GPU build of the same logic (capi_verify.hip: hashes on the device) is tested in tests/test_gpu_r2_verify.py."""
import random
import struct

import numpy as np
import pytest


@pytest.fixture(scope="module")
def tparams(hostcheck):
    h = hostcheck.params(1)
    yield h
    hostcheck.params_free(h)


@pytest.mark.parametrize("n0,sched,r", [(1 << 10, [16, 8], 8), (1 << 9, [8, 4, 2], 5), (1 << 11, [16, 16, 8], 6), (64, [8, 8], 4), (2, [2], 1)])
def test_deep_fri_verify_agrees_with_oracle_on_honest_and_tampered_proofs(oracle, hostcheck, tparams, n0, sched, r):
    cols = oracle.rand_fr_columns(77 + n0, n0, 4)
    ref = oracle.deep_fri_prove(cols[0], cols[1], cols[2], cols[3], n0, sched, r, 0xDEEFBAAD)
    proof = ref.bytes(); ref.free()
    assert oracle.deep_fri_verify(proof, sched, r, 0xDEEFBAAD) == 1
    assert hostcheck.deep_fri_verify(tparams, proof, sched, r) == 1
    # wrong parameters
    assert hostcheck.deep_fri_verify(tparams, proof, sched, r + 1) == 0
    assert hostcheck.deep_fri_verify(tparams, proof, sched[:-1], r) == 0
    # truncated / extended / empty byte strings never decode
    for bad in (b"", proof[:-1], proof + b"\0", proof[: len(proof) // 2]):
        assert hostcheck.deep_fri_verify(tparams, bad, sched, r) == 0
    # single-bit flips everywhere: the product verifier and the oracle's restatement must take the same decision
    rng = random.Random(n0 + r)
    positions = sorted(set([0, 7, 8, 8 + 31, 40, len(proof) - 1, len(proof) - 33, len(proof) - 41] + [rng.randrange(len(proof)) for _ in range(48 if n0 <= 1 << 10 else 24)]))
    accepted = 0
    for pos in positions:
        if pos < 0: continue
        bad = bytearray(proof); bad[pos] ^= 1 << rng.randrange(8)
        want = oracle.deep_fri_verify(bytes(bad), sched, r, 0xDEEFBAAD)
        got = hostcheck.deep_fri_verify(tparams, bytes(bad), sched, r)
        assert got == (1 if want == 1 else 0), f"byte {pos}: product {got}, oracle {want}"
        accepted += got
    # most of the proof is load-bearing; what the reference's verifier never reads (child_pos / parent_pos, the trailing omega) may flip freely
    assert accepted < len(positions) // 2


def test_verifier_first_payload_wins_and_local_check(oracle, hostcheck, tparams):
    """fri.rs:663-664 (`entry().or_insert`): for a repeated child index the FIRST query's payload is the one that is hashed, and
    fri.rs:168-176: s_i must equal f_parent_b for every query — so corrupting a LATER duplicate's f_i goes unnoticed by the Merkle
    check (both verifiers agree), while corrupting its s_i alone trips the local check."""
    n0, sched, r = 64, [8, 8], 24          # 24 queries over 64 leaves: repeated indices are certain
    cols = oracle.rand_fr_columns(5, n0, 4)
    ref = oracle.deep_fri_prove(cols[0], cols[1], cols[2], cols[3], n0, sched, r, 1); proof = ref.bytes(); ref.free()
    assert hostcheck.deep_fri_verify(tparams, proof, sched, r) == 1 == oracle.deep_fri_verify(proof, sched, r, 1)
    # walk the encoding to the per-query section and find two queries with the same layer-0 index
    tail = 8 + 32                                   # n0, omega0
    per_q = 8 + 2 * 32 + 8 + len(sched) * 32 + 8 + len(sched) * 128
    qbase = len(proof) - tail - r * per_q
    assert struct.unpack_from("<Q", proof, qbase - 8)[0] == r
    seen = {}
    for q in range(r):
        off = qbase + q * per_q
        i0 = struct.unpack_from("<Q", proof, off + 8)[0]
        pay = off + 8 + len(sched) * 32 + 8         # payload of layer 0: f_i, s_i, f_parent, s_parent
        if i0 in seen:
            bad = bytearray(proof); bad[pay + 3] ^= 4          # f_i of the later duplicate
            assert hostcheck.deep_fri_verify(tparams, bytes(bad), sched, r) == oracle.deep_fri_verify(bytes(bad), sched, r, 1) == 1
            bad = bytearray(proof); bad[pay + 32 + 3] ^= 4     # its s_i: local check s_i == f_parent_b
            assert hostcheck.deep_fri_verify(tparams, bytes(bad), sched, r) == oracle.deep_fri_verify(bytes(bad), sched, r, 1) == 0
            return
        seen[i0] = q
    pytest.fail("no repeated index among the queries")


@pytest.mark.parametrize("arity,n,label", [(16, 4096, 0), (16, 55, 9), (8, 19, 3), (2, 8, 1), (4, 64, 7), (32, 100, 4)])
def test_merkle_verify_single_roundtrip_and_tamper(oracle, hostcheck, tparams, arity, n, label):
    """merkle/src/lib.rs:1053-1136 shapes: commit -> open -> verify is true; a changed leaf, sibling, root, label or arity is false."""
    leaves = oracle.synth_column(31, arity, 0, n)
    t = oracle.merkle_build(arity, label, leaves)
    rng = random.Random(n); idx = sorted(set(rng.randrange(n) for _ in range(7)))
    pr = t.open_bytes(idx); root = t.root(); t.free()
    vals = leaves[idx]
    assert hostcheck.merkle_verify(tparams, False, arity, label, root, idx, vals, None, pr) == 1
    # caller order is free (indices are sorted inside, :598-600)
    perm = list(range(len(idx))); rng.shuffle(perm)
    assert hostcheck.merkle_verify(tparams, False, arity, label, root, [idx[i] for i in perm], vals[perm], None, pr) == 1
    bad = vals.copy(); bad[0, 0] ^= np.uint64(1)
    assert hostcheck.merkle_verify(tparams, False, arity, label, root, idx, bad, None, pr) == 0
    assert hostcheck.merkle_verify(tparams, False, arity, label + 1, root, idx, vals, None, pr) == 0           # DS hygiene (:1013-1050)
    r2 = root.copy(); r2[1] ^= np.uint64(2)
    assert hostcheck.merkle_verify(tparams, False, arity, label, r2, idx, vals, None, pr) == 0
    assert hostcheck.merkle_verify(tparams, False, arity, label, root, idx[:-1], vals[:-1], None, pr) == 0    # index set differs from the proof's
    assert hostcheck.merkle_verify(tparams, False, arity, label, root, [], vals[:0], None, pr) == 0
    if len(pr) > 200:
        b = bytearray(pr); b[len(pr) // 2] ^= 1
        assert hostcheck.merkle_verify(tparams, False, arity, label, root, idx, vals, None, bytes(b)) == 0
    wrong_width = 8 if arity > 8 else 16
    assert hostcheck.merkle_verify(tparams, False, wrong_width, label, root, idx, vals, None, pr) == 0          # ok_width (:610-618)


@pytest.mark.parametrize("arity,n", [(2, 8), (2, 2), (16, 64), (8, 32)])
def test_merkle_verify_pairs_roundtrip_and_tamper(oracle, hostcheck, tparams, arity, n):
    """merkle/src/lib.rs:1138-1168: pair-leaf trees (leaf DS level 2^32-1)."""
    f = oracle.synth_column(41, 0, 0, n); cp = oracle.synth_column(41, 1, 0, n)
    t = oracle.merkle_build(arity, 7, f, cp=cp)
    idx = sorted({0, n - 1, n // 2})
    pr = t.open_bytes(idx); root = t.root(); t.free()
    assert hostcheck.merkle_verify(tparams, True, arity, 7, root, idx, f[idx], cp[idx], pr) == 1
    bad = cp[idx].copy(); bad[-1, 2] ^= np.uint64(8)
    assert hostcheck.merkle_verify(tparams, True, arity, 7, root, idx, f[idx], bad, pr) == 0
    assert hostcheck.merkle_verify(tparams, True, arity, 8, root, idx, f[idx], cp[idx], pr) == 0
    assert hostcheck.merkle_verify(tparams, False, arity, 7, root, idx, f[idx], None, pr) == 0                # single-column verification of a pair tree

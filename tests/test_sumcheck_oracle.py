"""N4: the oracle's restatement of the sum-check consumer (oracle/channel.hpp <- crates/channel/src/lib.rs:7-1240) against the
reference's own round-trip tests (:1246-1452) — the reference holds no fixed values for this path, so these are properties: the
verifier accepts what the prover made, rejects tampering, and the messages have the sizes the bincode layout dictates.  CPU only."""
import random
import struct

import numpy as np
import pytest


def _fb(b, off):
    assert struct.unpack_from("<Q", b, off)[0] == 32
    return b[off + 8: off + 40], off + 40


@pytest.mark.parametrize("k,ds,seed", [(6, 5050, 42), (5, 2025, 7), (1, 1, 1), (10, 3030, 999)])
def test_plain_sumcheck_roundtrip_and_tamper(oracle, k, ds, seed):
    """e2e_sumcheck_roundtrip (:1338-1382): k = 6, ds_tag 5050, StdRng(42)."""
    w = oracle.rand_fr_columns(seed, 1 << k, 1)[0]
    p = oracle.sumcheck_prove(0, k, ds, w)
    assert len(p) == 40 + 8 + k * 80 + 1 + 40                      # root | rounds | None | final_eval (bincode, :935-941)
    assert oracle.sumcheck_verify(0, k, ds, p) == 1
    assert oracle.sumcheck_prove(0, k, ds, w) == p                 # deterministic
    # the first round's coefficients satisfy 2 c0 + c1 = sum of the table (send_claim, :434-446)
    c0b, off = _fb(p, 48); c1b, _ = _fb(p, off)
    c0 = oracle.from_le_bytes_mod_order(c0b); c1 = oracle.from_le_bytes_mod_order(c1b)
    s = oracle.from_u64(0)
    for x in w:
        s = oracle.add(s, x)
    assert (oracle.add(oracle.add(c0, c0), c1) == s).all()
    rng = random.Random(k)
    for pos in [8 + 3, 48 + 8 + 5, len(p) - 5] + [rng.randrange(48, len(p)) for _ in range(12)]:
        bad = bytearray(p); bad[pos] ^= 1 << rng.randrange(8)
        assert oracle.sumcheck_verify(0, k, ds, bytes(bad)) != 1, pos      # the root alone is only bound into the transcript; everything else is checked
    assert oracle.sumcheck_verify(0, k, ds, p[:-1]) == -1


@pytest.mark.parametrize("k,ds,q,seed", [(5, 6060, 3, 1337), (6, 11, 2, 5), (3, 9, 8, 2), (1, 4, 1, 3)])
def test_merkle_folded_sumcheck_roundtrip_and_tamper(oracle, k, ds, q, seed):
    """e2e_sumcheck_merkle_folded_roundtrip (:1384-1451): k = 5, ds_tag 6060, 3 queries per round, StdRng(1337)."""
    w = oracle.rand_fr_columns(seed, 1 << k, 1)[0]
    p = oracle.sumcheck_prove(1, k, ds, w, q=q)
    assert oracle.sumcheck_verify(1, k, ds, p, q=q) == 1
    assert oracle.sumcheck_prove(1, k, ds, w, q=q) == p
    # initial root = MerkleCommitment::commit(witness) (:608)
    rb, _ = _fb(p, 0)
    assert rb == oracle.to_bytes_le(oracle.commitment_root(ds, w))
    assert struct.unpack_from("<Q", p, 40)[0] == k                 # one RoundMF per variable
    rng = random.Random(k * 7 + q); rejected = 0
    positions = [8 + 3, len(p) - 5] + [rng.randrange(48, len(p)) for _ in range(40)]
    for pos in positions:
        bad = bytearray(p); bad[pos] ^= 1 << rng.randrange(8)
        rejected += oracle.sumcheck_verify(1, k, ds, bytes(bad), q=q) != 1
    assert rejected >= len(positions) - 2                          # (values the verifier never reads are few)
    assert oracle.sumcheck_verify(1, k, ds + 1, p, q=q) != 1      # another tree label: the openings no longer verify

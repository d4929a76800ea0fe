"""N4 on the GPU: prove_plain / prove_mf / verify_* through the C-ABI (commits on the Merkle kernels, device-resident transcript)
against the oracle's restatement of crates/channel/src/lib.rs — byte-identical proofs, the same accept / reject decisions.
Needs an MI355X: `pytest -m gpu`."""
import random
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,ds,seed", [(6, 5050, 42), (5, 2025, 7), (1, 1, 1), (12, 2025, 7), (14, 77, 3)])
def test_prove_plain_bytes_equal_oracle(gpu_ctx, oracle, k, ds, seed):
    w = oracle.rand_fr_columns(seed, 1 << k, 1)[0]
    t0 = time.time(); got = gpu_ctx.prove_plain(k, ds, w); dt = time.time() - t0
    want = oracle.sumcheck_prove(0, k, ds, w)
    assert got == want
    assert gpu_ctx.verify_plain(k, ds, got) is True
    rng = random.Random(k)
    for pos in [48 + 8 + 5, len(got) - 5] + [rng.randrange(48, len(got)) for _ in range(8)]:
        bad = bytearray(got); bad[pos] ^= 1 << rng.randrange(8)
        assert gpu_ctx.verify_plain(k, ds, bytes(bad)) == (oracle.sumcheck_verify(0, k, ds, bytes(bad)) == 1)
    assert gpu_ctx.verify_plain(k, ds, got[:-1]) is False
    print(f"prove_plain k={k}: {dt * 1e3:.1f} ms, {len(got)} bytes")


@pytest.mark.parametrize("k,ds,q,seed", [(5, 6060, 3, 1337), (6, 11, 2, 5), (3, 9, 8, 2), (1, 4, 1, 3), (12, 2025, 2, 7), (14, 5, 2, 9)])
def test_prove_mf_bytes_equal_oracle(gpu_ctx, oracle, k, ds, q, seed):
    """Every folded layer is committed on the GPU (MerkleCommitment: arity 16, "POSEIDON-T17-X5-SEED"): all k+1 roots, the query
    indices drawn from the device transcript, the openings and the final evaluation must equal the oracle's, byte for byte."""
    w = oracle.rand_fr_columns(seed, 1 << k, 1)[0]
    t0 = time.time(); got = gpu_ctx.prove_mf(k, ds, q, w); dt = time.time() - t0
    want = oracle.sumcheck_prove(1, k, ds, w, q=q)
    assert got == want
    assert gpu_ctx.verify_mf(k, ds, q, got) is True
    rng = random.Random(k * 5 + q)
    for pos in [8 + 3, len(got) - 5] + [rng.randrange(48, len(got)) for _ in range(10)]:
        bad = bytearray(got); bad[pos] ^= 1 << rng.randrange(8)
        assert gpu_ctx.verify_mf(k, ds, q, bytes(bad)) == (oracle.sumcheck_verify(1, k, ds, bytes(bad), q=q) == 1), pos
    assert gpu_ctx.verify_mf(k, ds + 1, q, got) is False
    print(f"prove_mf k={k} q={q}: {dt * 1e3:.1f} ms, {len(got)} bytes")


def test_sumcheck_argument_errors(gpu_ctx, oracle):
    from stark_mlwe_amd.api import StarkError
    w = oracle.rand_fr_columns(1, 48, 1)[0]
    with pytest.raises(StarkError):
        gpu_ctx.prove_plain(6, 1, w)            # Mle::new: "MLE length must be 2^k" (:259)


def test_transcript_object_matches_reference_flow(gpu_ctx, oracle):
    """transcript/src/lib.rs:119-152 shapes: new(label); absorb_bytes(msg); challenge(label) is deterministic and equals the oracle's
    transcript; a second challenge, long byte strings (several 31-byte words, several permutations) and `challenges` (label || le64(i))."""
    from stark_mlwe_amd.api import Transcript
    for label, msg, ch in ((b"test", b"hello", b"alpha"), (b"FSv1", b"x" * 100, b"c"), (b"L", b"", b"z")):
        t = Transcript(gpu_ctx, label); t.absorb_bytes(msg); got = t.challenge(ch); t.free()
        assert (got == oracle.transcript_vec(label, msg, ch)).all()
        t2 = Transcript(gpu_ctx, label); t2.absorb_bytes(msg); assert (t2.challenge(ch) == got).all(); t2.free()      # determinism (:124-136)
    # absorb_fields over several rate blocks + two challenges = tr_hash_fields_tagged's framing built by hand (fri.rs:28-35)
    xs = oracle.synth_column(9, 9, 0, 40)
    t = Transcript(gpu_ctx, b"FRI/FS"); t.absorb_bytes(b"ALI/A"); t.absorb_fields(xs)
    assert (t.challenge(b"out") == oracle.tr_hash_fields_tagged(b"ALI/A", xs)).all()
    c2 = t.challenge(b"out"); assert not (c2 == oracle.tr_hash_fields_tagged(b"ALI/A", xs)).all()                        # the state moved on
    t.free()
    t = Transcript(gpu_ctx, b"chs"); many = t.challenges(b"r", 3); t.free()
    t = Transcript(gpu_ctx, b"chs"); one = [t.challenge(b"r" + i.to_bytes(8, "little")) for i in range(3)]; t.free()
    assert all((many[i] == one[i]).all() for i in range(3))                                                          # :103-112


def test_field_crate_helpers(gpu_ctx, oracle):
    """crates/field tests (field/src/lib.rs:217-280): Domain::new(log_n) has omega^n = 1 and omega^(n/2) != 1; compute_powers(base, n)
    = [1, base, base^2, ...]; precomputed elements equal the powers of omega."""
    from stark_mlwe_amd.api import BLS12_381_FR, PALLAS_FR
    for field in (PALLAS_FR, BLS12_381_FR):
        size, lg, w, el = gpu_ctx.domain(11, field, precompute=True)
        assert size == 2048 and (oracle.pow(w, size, field) == oracle.from_u64(1, field)).all() and not (oracle.pow(w, size // 2, field) == oracle.from_u64(1, field)).all()
        assert (el[0] == oracle.from_u64(1, field)).all() and (el[1] == w).all() and (el[2047] == oracle.pow(w, 2047, field)).all()
        acc = oracle.from_u64(1, field)
        for i in range(40):
            assert (el[i] == acc).all(); acc = oracle.mul(acc, w, field)
    b = oracle.from_u64(3)
    p = gpu_ctx.compute_powers(b, 1000)
    assert (p[999] == oracle.pow(b, 999)).all() and (p[0] == oracle.from_u64(1)).all()
    assert gpu_ctx.compute_powers(b, 0).shape == (0, 4)


def test_channel_roundtrips_composed_from_the_abi(gpu_ctx, oracle):
    """e2e_merkle_channel_roundtrip and e2e_mle_commit_eval_roundtrip (channel/src/lib.rs:1253-1336) composed from the exported pieces:
    prover and verifier transcripts fed the same messages draw the same challenges; the commitment is MerkleCommitment's; the MLE
    evaluation at the drawn point equals the oracle's."""
    from stark_mlwe_amd.api import MerkleChannelCfg, Transcript
    table = oracle.rand_fr_columns(999, 32, 1)[0]
    seedp = gpu_ctx.generate_params_t17_x5(b"POSEIDON-T17-X5-SEED")
    cfg = MerkleChannelCfg(16, seedp, 3030)                                   # MerkleCommitment::tree_cfg (commitment/src/lib.rs:65-73)
    tree = gpu_ctx.merkle_new(table, cfg); root = tree.root()
    assert (root == oracle.commitment_root(3030, table)).all()
    tp, tv = Transcript(gpu_ctx, b"MLE-CHAN-E2E"), Transcript(gpu_ctx, b"MLE-CHAN-E2E")
    for t in (tp, tv):                                                        # send_digest / recv_digest (:22-26, :77-81)
        t.absorb_bytes(b"CHAN/SEND/DIGEST"); t.absorb_bytes(b"commit/root"); t.absorb_fields(root)
    r_p = np.stack([tp.challenge(b"r" + j.to_bytes(8, "little")) for j in range(5)])      # MleProver::draw_point (:315-324)
    r_v = np.stack([tv.challenge(b"r" + j.to_bytes(8, "little")) for j in range(5)])
    assert (r_p == r_v).all()
    val = gpu_ctx.mle_evaluate(table, r_p)
    assert (val == oracle.mle_evaluate(table, r_p)).all()
    idx = [0, 1, 2, 31]
    pr = tree.open_many(idx)
    # the generic verifier derives poseidon_params_for_arity(16) — other constants than MerkleCommitment's — and must refuse this tree
    assert gpu_ctx.merkle_verify_single(gpu_ctx.merkle_cfg(16, 3030), root, idx, table[idx], pr) is False
    tree.free(); tp.free(); tv.free(); seedp.free()


def test_commitment_scheme_roundtrip(gpu_ctx, oracle):
    """commitment/src/lib.rs:121-136 (merkle_commit_open_verify_roundtrip): n = 64 leaves, ds_tag 123, indices [0,15,16,31,47,63];
    plus the channel test's ragged n = 55 (channel/src/lib.rs:1264-1281)."""
    for n, ds, idx in ((64, 123, [0, 15, 16, 31, 47, 63]), (55, 2025, [0, 3, 7, 11, 54])):
        leaves = oracle.rand_fr_columns(42, n, 1)[0]
        root, tree = gpu_ctx.commitment_commit(ds, leaves)
        assert (root == oracle.commitment_root(ds, leaves)).all()
        pr = tree.open_many(idx); tree.free()
        assert gpu_ctx.commitment_verify(ds, root, idx, leaves[idx], pr) is True
        bad = leaves[idx].copy(); bad[2, 1] ^= np.uint64(1)
        assert gpu_ctx.commitment_verify(ds, root, idx, bad, pr) is False
        assert gpu_ctx.commitment_verify(ds + 1, root, idx, leaves[idx], pr) is False

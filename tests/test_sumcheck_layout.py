"""N4 (sum-check) is "parity unpinned": the reference holds no fixed vector for prove_plain / prove_mf, so product and oracle are only compared
with each other.  This test ties what CAN be tied to the reference TEXT rather than to the oracle: the proof bytes are decoded
field by field in the serde declaration order of ProofPlain / ProofMF / RoundMF / MerkleProofBytes / FBytes (crates/channel/src/lib.rs:893-979,
bincode 1.x: fixint little-endian, u64 length prefixes, 1-byte Option tag, serde_bytes = u64 length + raw bytes), with the structural facts the
reference's prover guarantees (k rounds, 2 openings per query, arity 16, 32-byte canonical field elements), and every transcript label the
reference's sum-check code absorbs (string constants copied from crates/channel/src/lib.rs:22-56, 175, 442-482, 593-597, 613, 627-668, 735, 1064)
must occur verbatim in both the product source and the oracle.  CPU only."""
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001

# crates/channel/src/lib.rs — labels on the sum-check paths (prove_plain, prove_mf and the channel under them)
REFERENCE_LABELS = [
    "CHAN/SEND/DIGEST", "CHAN/SEND/OPEN", "PROOF/ARITY", "PROOF/GROUP_SIZES", "PROOF/SIBLINGS",      # :22-56
    "commit/root",                                                                                 # :175
    "SUMCHECK/CLAIM", "SUMCHECK/ROUND", "COEFF/c0", "COEFF/c1", "SUMCHECK/FINAL/EVAL",                # :442-482
    "SUMCHECK-MF/ROUND-CHAL", "SUMCHECK/MF/R", "r_i",                                               # :593-597
    "sumcheck-mf/root/0", "SUMCHECK/MF/CLAIM", "SUMCHECK/MF/ROUND", "sumcheck-mf/root/next", "sumcheck-mf/q", "SUMCHECK/MF/FINAL/EVAL",   # :613-735
    "sumcheck/r",                                                                                  # :1064
]


class Dec:
    def __init__(self, b): self.b, self.o = b, 0
    def u64(self):
        v = struct.unpack_from("<Q", self.b, self.o)[0]; self.o += 8; return v
    def u8(self):
        v = self.b[self.o]; self.o += 1; return v
    def fbytes(self):                                  # struct FBytes(#[serde(with = "serde_bytes")] Vec<u8>), :895; serialize_compressed of Fr = 32 B LE canonical
        n = self.u64(); assert n == 32, n
        v = int.from_bytes(self.b[self.o:self.o + 32], "little"); self.o += 32
        assert v < R; return v
    def vec(self, item):
        return [item() for _ in range(self.u64())]
    def merkle_proof(self):                            # MerkleProofBytes { arity, group_sizes: Vec<Vec<u8>>, indices, siblings: Vec<Vec<FBytes>> }, :972-977
        arity = self.u64()
        group_sizes = self.vec(lambda: self.vec(self.u8))
        indices = self.vec(self.u64)
        siblings = self.vec(lambda: self.vec(self.fbytes))
        return arity, group_sizes, indices, siblings


def decode_plain(b):                                   # ProofPlain { root, rounds: Vec<(FBytes, FBytes)>, extra_openings: Option<..>, final_eval }, :942-947
    d = Dec(b)
    root = d.fbytes(); rounds = d.vec(lambda: (d.fbytes(), d.fbytes())); tag = d.u8(); assert tag == 0    # prove_plain never fills extra_openings (:1077)
    fin = d.fbytes(); assert d.o == len(b), "trailing bytes"
    return root, rounds, fin


def decode_mf(b):                                      # ProofMF { initial_root, rounds: Vec<RoundMF>, final_eval }, :949-966
    d = Dec(b)
    root0 = d.fbytes()
    def round_():
        c0, c1, nxt = d.fbytes(), d.fbytes(), d.fbytes()
        ci = d.vec(d.u64); cv = d.vec(d.fbytes); cp = d.merkle_proof()
        ni = d.vec(d.u64); nv = d.vec(d.fbytes); np_ = d.merkle_proof()
        return dict(c0=c0, c1=c1, next_root=nxt, cur_indices=ci, cur_values=cv, cur_proof=cp, next_indices=ni, next_values=nv, next_proof=np_)
    rounds = d.vec(round_); fin = d.fbytes(); assert d.o == len(b), "trailing bytes"
    return root0, rounds, fin


@pytest.mark.parametrize("k", [3, 6])
def test_proof_plain_bytes_follow_the_reference_struct_order(oracle, k):
    w = oracle.synth_column(900 + k, 0, 0, 1 << k)
    b = oracle.sumcheck_prove(0, k, 2025, w)
    root, rounds, fin = decode_plain(b)
    assert len(rounds) == k                                                                 # one (c0, c1) per variable (:1063-1066)
    assert len(b) == 40 + 8 + 80 * k + 1 + 40
    can = lambda x: oracle.to_int(x)
    assert root == can(oracle.commitment_root(2025, w))                                     # MerkleCommitment::commit of the witness (:1052-1057)
    # the first round's coefficients satisfy s = 2 c0 + c1 with s = sum of the table (sumcheck_round_coeffs, :400-418)
    s = sum(can(x) for x in w) % R
    assert (2 * rounds[0][0] + rounds[0][1]) % R == s


@pytest.mark.parametrize("k,q", [(4, 2), (6, 3)])
def test_proof_mf_bytes_follow_the_reference_struct_order(oracle, k, q):
    w = oracle.synth_column(950 + k, 0, 0, 1 << k)
    b = oracle.sumcheck_prove(1, k, 77, w, q)
    root0, rounds, fin = decode_mf(b)
    assert len(rounds) == k
    assert root0 == oracle.to_int(oracle.commitment_root(77, w))
    n = 1 << k
    for i, r in enumerate(rounds):
        half = n >> (i + 1)
        nq = min(max(q, 1), half)                                                           # q_target (:662)
        assert len(r["next_indices"]) <= nq and len(r["next_indices"]) >= 1                  # unique sorted queries, at most q_target
        assert r["next_indices"] == sorted(set(r["next_indices"]))
        assert r["cur_indices"] == [x for j in r["next_indices"] for x in (2 * j, 2 * j + 1)]   # both children of every queried parent (:686-690)
        assert len(r["cur_values"]) == len(r["cur_indices"]) and len(r["next_values"]) == len(r["next_indices"])
        assert r["cur_proof"][0] == 16 and r["next_proof"][0] == 16                           # MerkleChannelCfg arity 16 (commitment/src/lib.rs:85-90)
        assert r["cur_proof"][2] == r["cur_indices"] and r["next_proof"][2] == r["next_indices"]
        assert all(x < 2 * half for x in r["cur_indices"]) and all(x < half for x in r["next_indices"])
        # the fold relation the verifier checks (:838-850): next = (1 - r_i) a + r_i b — here only its shape; r_i needs the transcript
    assert (2 * rounds[0]["c0"] + rounds[0]["c1"]) % R == sum(oracle.to_int(x) for x in w) % R


def test_transcript_labels_are_the_reference_strings():
    prod = open(os.path.join(ROOT, "stark_mlwe_amd", "csrc", "sumcheck_impl.hpp")).read()
    orc = open(os.path.join(ROOT, "oracle", "channel.hpp")).read()
    for lab in REFERENCE_LABELS:
        assert f'"{lab}"' in prod, f"product: label {lab!r} of crates/channel/src/lib.rs not found verbatim"
        assert f'"{lab}"' in orc, f"oracle: label {lab!r} of crates/channel/src/lib.rs not found verbatim"

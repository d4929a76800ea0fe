"""The oracle against public known-answer tests, the SURVEY Appendix-A constants, and an independent
pure-Python restatement (tests/pyref.py).  CPU only."""
import random

import numpy as np
import pytest

import pyref

BLAKE3_EMPTY = "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"   # official test vector, input_len 0
BLAKE3_ONE_ZERO = "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213"  # official test vector, input_len 1 (byte 0x00)
# ChaCha12, 256-bit zero key, zero IV, first 32 keystream bytes (eSTREAM / reference implementation vector)
CHACHA12_ZERO = "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f"


def test_blake3_kats(oracle, hostcheck):
    for impl in (oracle.blake3, pyref.blake3, hostcheck.blake3):
        assert impl(b"").hex() == BLAKE3_EMPTY
        assert impl(b"\x00").hex() == BLAKE3_ONE_ZERO


@pytest.mark.parametrize("n", [1, 2, 31, 58, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 2048, 2049, 3072, 4097, 7000])
def test_blake3_three_implementations_agree(oracle, hostcheck, n):
    data = bytes((i % 251) for i in range(n))          # the official test-vector input pattern
    a, b, c = oracle.blake3(data), pyref.blake3(data), hostcheck.blake3(data)
    assert a == b == c


def test_chacha12_kat_and_agreement(oracle, hostcheck):
    ks = oracle.stdrng_from_seed_u64s(bytes(32), 4)
    assert b"".join(int(x).to_bytes(8, "little") for x in ks).hex() == CHACHA12_ZERO
    seed = bytes(range(32))
    a = [int(x) for x in oracle.stdrng_from_seed_u64s(seed, 40)]   # crosses a 64-byte block boundary
    assert a == pyref.chacha12_u64s(seed, 40) == [int(x) for x in hostcheck.chacha12_u64s(seed, 40)]


def test_field_constants_appendix_a(oracle):
    # SURVEY.md Appendix A (byte patterns found in the reference's own build artefacts)
    one = oracle.from_u64(1)
    assert sum(int(one[i]) << (64 * i) for i in range(4)) == 0x3fffffffffffffffffffffffffffffff992c350be34205675b2b3e9cfffffffd
    one_b = oracle.from_u64(1, field=1)
    assert sum(int(one_b[i]) << (64 * i) for i in range(4)) == 0x1824b159acc5056f998c4fefecbc4ff55884b7fa0003480200000001fffffffe
    assert oracle.to_int(oracle.root_of_unity(32)) == 0x2de6a9b8746d3f589e5c4dfd492ae26e9bb97ea3c106f049a70e2c1102b6d05f
    w32 = oracle.root_of_unity(32)
    assert sum(int(w32[i]) << (64 * i) for i in range(4)) == 0x0b79fa897f2db056ac2e5d27b2efbee2cc49578921b60494218077428c9942de
    assert oracle.to_int(oracle.root_of_unity(12)) == 0x0d22c941a0b04b71d4ce80179437306907072be91feb7c9acf2eecbb96072570
    assert oracle.to_int(oracle.root_of_unity(20)) == 0x19e9df871f6c4b14a3615751d1f43923b5f9ce8d08f92d2629c2afb67fb34869
    assert oracle.to_int(oracle.root_of_unity(24)) == 0x144e6eba6d684acf84e02261743f788b2ae72b37bffdb012102c8c2f7fa53afe
    assert oracle.to_int(oracle.root_of_unity(32, field=1), field=1) == 0x16a2a19edfe81f20d09b681922c813b4b63683508c2280b93829971f439f0d2b
    # field/src/lib.rs:262-279 roots_have_correct_order (n = 2048): w^n == 1, w^(n/2) != 1
    w = oracle.root_of_unity(11)
    assert oracle.to_int(oracle.pow(w, 2048)) == 1 and oracle.to_int(oracle.pow(w, 1024)) != 1


@pytest.mark.parametrize("field,p", [(0, pyref.P_PALLAS), (1, pyref.P_BLS)])
def test_field_arithmetic_vs_bigint(oracle, field, p):
    rng = random.Random(1234 + field)
    for _ in range(200):
        x, y = rng.randrange(p), rng.randrange(p)
        a, b = oracle.from_int(x, field), oracle.from_int(y, field)
        assert oracle.to_int(oracle.add(a, b, field), field) == (x + y) % p
        assert oracle.to_int(oracle.sub(a, b, field), field) == (x - y) % p
        assert oracle.to_int(oracle.mul(a, b, field), field) == (x * y) % p
    for x in (1, 2, p - 1, rng.randrange(p)):
        assert oracle.to_int(oracle.inv(oracle.from_int(x, field), field), field) == pow(x, -1, p)
    # edge: values at the top of the range
    a = oracle.from_int(p - 1, field)
    assert oracle.to_int(oracle.mul(a, a, field), field) == 1
    assert oracle.to_int(oracle.add(a, oracle.from_int(1, field), field), field) == 0


def test_from_le_bytes_mod_order_and_serialization(oracle):
    rng = random.Random(7)
    for n in (0, 1, 20, 31, 32, 33, 64):
        b = bytes(rng.randrange(256) for _ in range(n))
        assert oracle.to_int(oracle.from_le_bytes_mod_order(b)) == int.from_bytes(b, "little") % pyref.P_PALLAS
    x = rng.randrange(pyref.P_PALLAS)
    assert oracle.to_bytes_le(oracle.from_int(x)) == x.to_bytes(32, "little")      # fr_to_bytes_compressed, field/src/lib.rs:206-215


@pytest.mark.parametrize("t", [9, 17])
def test_poseidon_params_and_permute_vs_pyref(oracle, t):
    P = pyref.params_for_width(t)
    rf, rp, mds, rcf, rcp = oracle.poseidon_params(0, t)
    assert (rf, rp) == (8, pyref.RP_FOR_T[t])
    assert [oracle.to_int(mds[i]) for i in range(t * t)] == [P["mds"][i][j] for i in range(t) for j in range(t)]
    assert [oracle.to_int(rcf[i]) for i in range(8 * t)] == [P["rc_full"][r][i] for r in range(8) for i in range(t)]
    assert [oracle.to_int(x) for x in rcp] == P["rc_partial"]
    st = list(range(1, t + 1))
    out = oracle.permute(0, t, np.array([pyref.to_limbs(x) for x in st], np.uint64))
    assert [pyref.from_limbs(out[i]) for i in range(t)] == pyref.permute(st, P)
    zero = oracle.permute(0, t, np.zeros((t, 4), np.uint64))
    assert [pyref.from_limbs(zero[i]) for i in range(t)] == pyref.permute([0] * t, P)


def test_params_exist_for_supported_widths(oracle):
    # poseidon/src/lib.rs:457-470 params_exist_for_supported_widths
    for t in (9, 17, 33, 65, 129):
        rf, rp, mds, rcf, rcp = oracle.poseidon_params(0, t)
        assert rf == 8 and rp > 0 and mds.shape[0] == t * t and rcf.shape[0] == 8 * t and rcp.shape[0] == rp
    assert [oracle.pick_arity_for_layer(1 << 12, m) for m in (16, 8, 1, 128)] == [16, 8, 2, 128]


def test_sponges_vs_pyref(oracle):
    L = lambda xs: np.array([pyref.to_limbs(x) for x in xs], np.uint64)
    P17, P9 = pyref.params_for_width(17), pyref.params_for_width(9)
    # merkle/src/lib.rs:966-1010 shapes: 16 children (2 blocks), 5 children, 5 children + explicit zero, 11 children (exact block)
    ds = [16, 0, 3, 42]
    for children in (list(range(1, 17)), list(range(1, 6)), list(range(1, 6)) + [0], list(range(1, 12))):
        got = oracle.hash_with_ds_dynamic(0, 17, L(ds), L(children), len(children))
        assert pyref.from_limbs(got) == pyref.hash_with_ds_dynamic(ds, children, P17)
    ds9 = [8, 2, 5, 7]
    for children in (list(range(11, 19)), [21, 22, 23]):
        got = oracle.hash_with_ds_dynamic(0, 9, L(ds9), L(children), len(children))
        assert pyref.from_limbs(got) == pyref.hash_with_ds_dynamic(ds9, children, P9)
    # hash_leaf_pair and tr_hash_fields_tagged (fri.rs:28-44)
    assert pyref.from_limbs(oracle.leaf_pair_hash(L([1]), L([2]), 1)[0]) == pyref.hash_leaf_pair(1, 2)
    for n in (0, 1, 3, 12, 13, 29, 40):
        xs = [(7 * i + 3) % pyref.P_PALLAS for i in range(n)]
        assert pyref.from_limbs(oracle.tr_hash_fields_tagged(b"FRI/index", L(xs) if n else np.zeros((0, 4), np.uint64))) == pyref.tr_hash_fields_tagged(b"FRI/index", xs)


def test_transcript_reference_tests(oracle):
    # transcript/src/lib.rs:124-151 deterministic / sensitive_to_input
    a = oracle.transcript_vec(b"ctx-A", b"hello", b"alpha")
    assert (a == oracle.transcript_vec(b"ctx-A", b"hello", b"alpha")).all()
    assert (a != oracle.transcript_vec(b"ctx-A", b"hellp", b"alpha")).any()
    tr = pyref.Transcript(b"ctx-A"); tr.absorb_bytes(b"hello")
    assert pyref.from_limbs(a) == tr.challenge(b"alpha")


@pytest.mark.parametrize("field,p", [(0, pyref.P_PALLAS), (1, pyref.P_BLS)])
def test_ntt_oracle_is_the_dft(oracle, field, p):
    rng = random.Random(5 + field)
    for lg in (1, 3, 6):
        n = 1 << lg
        xs = [rng.randrange(p) for _ in range(n)]
        a = np.array([pyref.to_limbs(x, p) for x in xs], np.uint64)
        w = oracle.to_int(oracle.root_of_unity(lg, field), field)
        want = pyref.dft(xs, w, p)
        assert [pyref.from_limbs(v, p) for v in oracle.dft_naive(field, a)] == want
        assert [pyref.from_limbs(v, p) for v in oracle.ntt(field, a)] == want
        assert (oracle.ntt(field, oracle.ntt(field, a), inverse=True) == a).all()
    # fft/src/lib.rs:39-54 roundtrip_fft_ifft: all-ones vector, n = 8
    ones = np.tile(oracle.from_u64(1, field), (8, 1))
    assert (oracle.ntt(field, oracle.ntt(field, ones), inverse=True) == ones).all()
    # larger: radix-2 against the O(n^2) definition
    a = oracle.synth_column(99, 7, 0, 512)
    assert (oracle.ntt(field, a) == oracle.dft_naive(field, a)).all()
    assert (oracle.ntt(field, a, inverse=True) == oracle.dft_naive(field, a, inverse=True)).all()
    # LDE definition: restricting the blown-up evaluations (shift 1) to every 2^b-th point gives back the input
    ev = oracle.synth_column(5, 1, 0, 64)
    assert (oracle.lde(field, ev, 3)[::8] == ev).all()

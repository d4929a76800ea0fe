"""Product host logic and kernel BODIES (host build of the same inline code, libstark_mlwe_hostcheck.so)
against the oracle.  CPU only; this is a diagnostic library — no product path calls it."""
import random

import numpy as np
import pytest

import pyref


@pytest.mark.parametrize("field,p", [(0, pyref.P_PALLAS), (1, pyref.P_BLS)])
def test_fr_hpp_vs_oracle(oracle, hostcheck, field, p):
    rng = random.Random(99 + field)
    vals = [0, 1, 2, p - 1, p - 2, (1 << 254) % p, (1 << 255) % p] + [rng.randrange(p) for _ in range(60)]
    for x in vals:
        for y in (vals[rng.randrange(len(vals))], vals[rng.randrange(len(vals))]):
            a, b = oracle.from_int(x, field), oracle.from_int(y, field)
            for op in (0, 1, 2):
                assert (hostcheck.fr_op(field, op, a, b) == oracle.fr_op(field, op, a, b)).all(), (op, x, y)
    for x in (1, 5, p - 1, rng.randrange(p)):
        a = oracle.from_int(x, field)
        assert (hostcheck.fr_op(field, 3, a) == oracle.inv(a, field)).all()
        assert (hostcheck.fr_op(field, 5, a) == oracle.to_canonical(a, field)).all()
        assert (hostcheck.fr_op(field, 7, a, np.array([12345, 0, 0, 0], np.uint64)) == oracle.pow(a, 12345, field)).all()
    for u in (0, 1, 2**32, 2**64 - 1):
        assert (hostcheck.fr_op(field, 4, np.array([u, 0, 0, 0], np.uint64)) == oracle.from_u64(u, field)).all()
    for lg in (1, 11, 20, 32):
        assert (hostcheck.fr_op(field, 6, np.array([lg, 0, 0, 0], np.uint64)) == oracle.root_of_unity(lg, field)).all()


def test_host_bytes_helpers(oracle, hostcheck):
    rng = random.Random(3)
    for n in (1, 20, 31, 32, 33, 64):
        b = bytes(rng.randrange(256) for _ in range(n))
        assert (hostcheck.from_le_bytes_mod_order(b) == oracle.from_le_bytes_mod_order(b)).all()
    a = oracle.from_int(rng.randrange(pyref.P_PALLAS))
    assert hostcheck.to_bytes_le(a) == oracle.to_bytes_le(a)


@pytest.mark.parametrize("kind,t,seed,okind", [(0, 9, b"", 0), (0, 17, b"", 0), (1, 17, b"", 1), (2, 17, b"POSEIDON-T17-X5-SEED", 2), (2, 17, b"POSEIDON-T17-X5", 3), (0, 33, b"", 0)])
def test_constants_and_kernel_form_permutation(oracle, hostcheck, kind, t, seed, okind):
    h = hostcheck.params(kind, t, seed)
    assert hostcheck.params_ok(h) == 1
    rf, rp, mds, rcf, rcp = oracle.poseidon_params(okind, t)
    m2, f2, p2 = hostcheck.params_export(h, t, rf, rp)
    assert (m2 == mds).all() and (f2 == rcf).all() and (p2 == rcp).all()
    n = 3 if t > 17 else 6
    states = np.concatenate([oracle.synth_column(11, 2, 0, (n - 1) * t), np.zeros((t, 4), np.uint64)]).reshape(n, t, 4)
    want = oracle.permute(okind, t, states)
    assert (hostcheck.permute_dense(h, states, t) == want).all()
    assert (hostcheck.permute_kernel_form(h, states, t) == want).all()      # LU + sparse partial rounds == dense rounds, bit for bit
    if t == 17:
        # the partial rounds unrolled over all 64 rounds from the chain tables (E_q, Gamma_{q,p}, w_{p,j} with their 2^25 scaling): the algebra
        # and the tables of the five-wave latency kernel (poseidon_chain.hpp) give the reference's permutation, bit for bit
        assert (hostcheck.permute_chain_model(h, states, t) == want).all()
    hostcheck.params_free(h)


def test_leaf_pair_body(oracle, hostcheck):
    h = hostcheck.params(1)
    f = oracle.synth_column(1, 0, 0, 40); fn = oracle.synth_column(1, 1, 0, 5)
    assert (hostcheck.leaf_pair(h, f, fn, 8) == oracle.leaf_pair_hash(f, fn, 8)).all()
    assert (hostcheck.leaf_pair(h, f[:7], None, 1) == oracle.leaf_pair_hash(f[:7], None, 1)).all()
    hostcheck.params_free(h)


@pytest.mark.parametrize("arity,n", [(16, 64), (16, 55), (8, 24), (8, 19), (2, 6), (4, 16), (16, 16), (16, 11)])
def test_hash_ds_body_builds_oracle_tree(oracle, hostcheck, arity, n):
    t = 9 if arity <= 8 else 17
    h = hostcheck.params(0, t)
    leaves = oracle.synth_column(4, 3, 0, n)
    tree = oracle.merkle_build(arity, 77, leaves)
    cur, level = leaves, 0
    while cur.shape[0] > 1:
        cur = hostcheck.hash_ds_level(h, 0, arity, level, 0, 77, cur)
        level += 1
        assert (cur == tree.level(level)).all()
    tree.free()
    # pair-leaf level (merkle/src/lib.rs:380-388)
    cp = oracle.synth_column(4, 4, 0, n)
    ptree = oracle.merkle_build(arity, 5, leaves, cp)
    assert (hostcheck.hash_ds_level(h, 1, arity, 0xFFFFFFFF, 0, 5, leaves, cp) == ptree.level(0)).all()
    ptree.free(); hostcheck.params_free(h)


def test_sharded_level_positions(oracle, hostcheck):
    # a shard hashing parents [pos0, pos0+k) of a larger level gets the same digests (global DS positions)
    h = hostcheck.params(0, 17)
    leaves = oracle.synth_column(8, 0, 0, 256)
    whole = hostcheck.hash_ds_level(h, 0, 16, 0, 0, 3, leaves)
    part = hostcheck.hash_ds_level(h, 0, 16, 0, 8, 3, leaves[128:])
    assert (part == whole[8:]).all()
    hostcheck.params_free(h)


def test_tr_hash_body(oracle, hostcheck):
    h = hostcheck.params(1)
    for n in (0, 1, 3, 12, 13, 28, 29, 45):
        xs = oracle.synth_column(6, 0, 0, n) if n else np.zeros((0, 4), np.uint64)
        assert (hostcheck.tr_hash(h, b"ALI/A", xs) == oracle.tr_hash_fields_tagged(b"ALI/A", xs)).all(), n
    batch = oracle.synth_column(6, 1, 0, 15)
    got = hostcheck.tr_hash(h, b"FRI/index", batch, n=5)
    for i in range(5):
        assert (got[i] == oracle.tr_hash_fields_tagged(b"FRI/index", batch[3 * i:3 * i + 3])).all()
    hostcheck.params_free(h)


def test_hash_stream_body(oracle, hostcheck):
    h17, hseed = hostcheck.params(0, 17), hostcheck.params(2, 17, b"POSEIDON-T17-X5-SEED")
    ds = np.array([oracle.from_u64(x) for x in (16, 0, 3, 42)])
    for cnt in (0, 5, 11, 12, 16, 27, 28):
        ch = oracle.synth_column(2, 0, 0, cnt) if cnt else np.zeros((0, 4), np.uint64)
        assert (hostcheck.hash_stream(h17, 0, ds, 4, ch, cnt) == oracle.hash_with_ds_dynamic(0, 17, ds, ch, cnt)).all(), cnt
    tag = oracle.from_u64(77)
    for cnt in (0, 1, 2, 16, 17, 37):
        ch = oracle.synth_column(2, 1, 0, cnt) if cnt else np.zeros((0, 4), np.uint64)
        assert (hostcheck.hash_stream(hseed, 1, None, 0, ch, cnt, tag) == oracle.hash_with_ds(2, ch, tag)).all(), cnt
    hostcheck.params_free(h17); hostcheck.params_free(hseed)


@pytest.mark.parametrize("n", [1, 2, 6, 7, 12, 13, 17, 24, 25, 27, 48, 59, 60, 61, 120, 129])
def test_wide_dot_worst_case_and_random(oracle, hostcheck, n):
    """Sums of products with ONE Montgomery reduction per chunk (radix-2^29 columns, constants in the 2^261 domain:
    carry pass every 6 terms, chunks of 60 — csrc/fr29.hpp): worst case (all operands r-1) and random, across every
    chunk boundary; the radix-2^32 accumulator of the cooperative kernels must agree where it applies (<= 24 terms)."""
    p = pyref.P_PALLAS
    top = oracle.from_int(p - 1)
    a = np.tile(top, (n, 1)); b = np.tile(top, (n, 1))
    assert oracle.to_int(hostcheck.wide_dot(a, b)) == (n * (p - 1) * (p - 1)) % p
    a = oracle.synth_column(17, 0, 0, n); b = oracle.synth_column(17, 1, 0, n)
    want = sum(oracle.to_int(a[i]) * oracle.to_int(b[i]) for i in range(n)) % p
    assert oracle.to_int(hostcheck.wide_dot(a, b)) == want
    if n <= 24:
        assert oracle.to_int(hostcheck.wide_dot32(a, b)) == want
        assert oracle.to_int(hostcheck.wide_dot32(np.tile(top, (n, 1)), np.tile(top, (n, 1)))) == (n * (p - 1) * (p - 1)) % p


@pytest.mark.parametrize("field,p", [(0, pyref.P_PALLAS), (1, pyref.P_BLS)])
def test_lazy_nine_limb_ntt_tile_vs_oracle(oracle, hostcheck, field, p):
    """The NTT kernels' arithmetic (csrc/ntt_dev.hpp: nine 29-bit limbs, lazily reduced decimation-in-time butterflies, tables carrying the
    factor 32, carry pass before stages 4/7/10) instantiated on the host: equal to the oracle's radix-2 NTT for every sub-NTT size a pass can
    take, forward and inverse, on random values and on the worst case (every input r - 1), with the operand bounds the column sums rely on."""
    rng = random.Random(4242 + field)
    for log_b in list(range(1, 11)) + [12]:
        n = 1 << log_b
        for kind in ("random", "max", "alternating"):
            if kind == "random": vals = [rng.randrange(p) for _ in range(n)]
            elif kind == "max": vals = [p - 1] * n
            else: vals = [(p - 1) if i & 1 else 0 for i in range(n)]
            x = np.stack([oracle.from_int(v, field) for v in vals])
            for inverse in (False, True):
                got, max_limb, max_top = hostcheck.ntt29(field, x, inverse)
                assert (got == oracle.ntt(field, x, inverse)).all(), (log_b, kind, inverse)
                assert max_limb <= 6 << 29 and max_top < 1 << 29, (log_b, kind, max_limb, max_top)


@pytest.mark.parametrize("field,p", [(0, pyref.P_PALLAS), (1, pyref.P_BLS)])
def test_partial_reduce_without_product(hostcheck, field, p):
    """fr29_partial_reduce (the NTT's product-free reduction of a lazy nine-limb value): same residue, limbs below 2^29, result below
    1.00002 r — for random lazy values up to the 2^261 capacity, for limbs at the 7 * 2^29 bound the butterflies can reach, for exact
    multiples of r (the quotient estimate must never overshoot), for values just below them and for canonical inputs."""
    rng = random.Random(77 + field)
    M = (1 << 29) - 1

    def limbs_of(v, lazy):
        """nine limbs of v; with `lazy` some weight is moved between neighbours so that limbs exceed 29 bits (value unchanged)"""
        l = [(v >> (29 * i)) & M for i in range(8)] + [v >> 232]
        if lazy:
            for i in range(8):
                if l[i + 1] > 0 and l[i] + (1 << 29) <= 7 << 29:
                    k = rng.randrange(0, min(l[i + 1], 6 - (l[i] >> 29)) + 1); l[i + 1] -= k; l[i] += k << 29
        return l

    cases = []
    for _ in range(300):
        cases.append(limbs_of(rng.randrange(1 << 261), rng.random() < 0.7))
    for k in list(range(0, 70)) + [100, 127]:
        if k * p < 1 << 261:
            cases.append(limbs_of(k * p, False))
            if k: cases.append(limbs_of(k * p - 1, True)); cases.append(limbs_of(k * p + 1, True))
    cases.append([7 << 29] * 8 + [(1 << 28)])
    cases.append(limbs_of(p - 1, False)); cases.append([0] * 9)
    arr = np.array(cases, dtype=np.uint64).astype(np.uint32)
    assert (arr.astype(np.uint64) == np.array(cases, dtype=np.uint64)).all()
    out = hostcheck.partial_reduce(field, arr)
    for lin, lout in zip(cases, out.tolist()):
        vin = sum(x << (29 * i) for i, x in enumerate(lin)); vout = sum(x << (29 * i) for i, x in enumerate(lout))
        assert vin < 1 << 261
        assert vout % p == vin % p
        assert all(x <= M for x in lout[:8]) and vout < p + (p >> 15), (lin, lout)


def test_matrix_core_full_round_tables_and_fold(oracle, hostcheck):
    """The t = 17 wave-pair kernels multiply the full rounds' dense matrices on the int8 matrix cores (poseidon_pair.hpp).  Everything of that path
    that is not the MFMA instruction itself is host-checkable: the signed radix-256 recoding, the Toeplitz FRAGMENT TABLES (host_util.hpp mfma_frags,
    emulated as D[row][col] = sum_k A[row][k] B[k][col] over the tables' lane layout), the accumulator-row order the lanes see, the fold into 29-bit
    columns, the signed carry pass and the Montgomery step.  Equal to the L*U rows of the VALU path on random S-box outputs and on the corners of the
    recoding (0, r - 1, 0x7f / 0x80 / 0xff byte patterns), for M and for B_1 * M, with the Merkle and the transcript parameter sets."""
    p = pyref.P_PALLAS
    rng = random.Random(1717)
    corner = [0, 1, p - 1, p - 2, (p - 1) // 2] + [int.from_bytes(bytes([b]) * 32, "little") % p for b in (0x7f, 0x80, 0x81, 0xff)]
    for kind in (0, 1):
        h = hostcheck.params(kind, 17)
        vals = [[rng.randrange(p) for _ in range(17)] for _ in range(6)] + [[corner[(i + j) % len(corner)] for j in range(17)] for i in range(4)]
        st = np.stack([np.stack([oracle.from_int(v) for v in row]) for row in vals])
        for pre in (False, True):
            a = hostcheck.full_round_linear(h, 0, pre, st); b = hostcheck.full_round_linear(h, 1, pre, st)
            assert (a == b).all(), (kind, pre)
            assert not (a == st).all()
        hostcheck.params_free(h)


def test_row_form_montgomery_constants_and_chain_table_shapes(hostcheck):
    """The constants the row-form product (poseidon_chain.hpp row::mul) is built on, against big-integer arithmetic: ni = -r^-1 mod 2^261 and t = r - 2^254
    in radix 2^29 (r = Pallas Fr); and the chain tables of the transcript parameters have the documented shapes, every limb below
    2^29, wave B's table zero exactly where q < p + 2."""
    r = pyref.P_PALLAS; X = 1 << 29
    ni = (-pow(r, -1, X ** 9)) % (X ** 9); t = r - (1 << 254)
    k = hostcheck.row_consts()
    assert [int(x) for x in k[:9]] == [(ni >> (29 * i)) & (X - 1) for i in range(9)]
    assert [int(x) for x in k[9:]] == [(t >> (29 * i)) & (X - 1) for i in range(5)] and t < (1 << (29 * 5))
    h = hostcheck.params(1)
    a, g, w = (hostcheck.chain_table(h, i) for i in range(3))
    assert a.shape == (64 * 64,) and g.shape == (64 * 9 * 64,) and w.shape == (64 * 16 * 9,)
    assert int(max(a.max(), g.max(), w.max())) < X
    g = g.reshape(64, 9, 64)
    for p in range(64):
        assert not g[p, :, :min(p + 2, 64)].any()                       # lane q < p + 2 takes nothing from y_p on wave B
        if p + 2 < 64: assert g[p, :, p + 2:].any()
    a = a.reshape(64, 4, 16)
    assert not a[:, 0].any() and not a[:, 3].any() and not a[:, :, 9:].any() and not a[63, 2].any()   # rows 0 and 3 unused, limbs 9..15 zero, no Gamma after the last round
    hostcheck.params_free(h)

// oracle/poseidon.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// Line-by-line restatement of crates/poseidon/src/lib.rs, crates/utils/src/lib.rs:7-22 and
// crates/transcript/src/lib.rs:13-117 of the reference (dense MDS in EVERY round, eager dynamic
// sponge, lazy transcript duplex).  No algebraic shortcuts here: this is the checker.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "fr.hpp"
#include "blake3.hpp"

namespace oracle {

// utils/src/lib.rs:7-13  fr_from_le_bytes_mod_p: zero-extend to 64 bytes, reduce mod r.
static inline Fr fr_from_le_bytes_mod_p(const uint8_t* b, size_t n) {
    uint8_t wide[64]; memset(wide, 0, 64);
    size_t len = n < 64 ? n : 64;
    memcpy(wide, b, len);
    return Fr::from_le_bytes_mod_order(wide, 64);
}
// utils/src/lib.rs:16-22  fr_from_hash(tag, data) = reduce(BLAKE3(tag || data)).
static inline Fr fr_from_hash(const std::string& tag, const std::vector<uint8_t>& data) {
    std::vector<uint8_t> buf(tag.begin(), tag.end());
    buf.insert(buf.end(), data.begin(), data.end());
    uint8_t out[32]; blake3::hash(buf.data(), buf.size(), out);
    return fr_from_le_bytes_mod_p(out, 32);
}
static inline void push_le64(std::vector<uint8_t>& v, uint64_t x) { for (int j = 0; j < 8; ++j) v.push_back((uint8_t)(x >> (8 * j))); }

// poseidon/src/lib.rs:104-114  PoseidonParamsDynamic (also carries the static t=17 PoseidonParams, :16-21).
struct PoseidonParams {
    size_t t = 0, rate = 0, rounds_full = 0, rounds_partial = 0;
    std::vector<std::vector<Fr>> mds;      // t x t, mds[i][j]
    std::vector<std::vector<Fr>> rc_full;  // RF x t
    std::vector<Fr> rc_partial;            // RP
};

// poseidon/src/lib.rs:176-216 (derive_mds / derive_rc_full / derive_rc_partial) and :318-356
// (params::generate_params_t17_x5) — identical derivations, parameterised by (seed, t, rf, rp).
static inline PoseidonParams derive_params(const std::vector<uint8_t>& seed, size_t t, size_t rf, size_t rp) {
    PoseidonParams p; p.t = t; p.rate = t - 1; p.rounds_full = rf; p.rounds_partial = rp;
    p.mds.assign(t, std::vector<Fr>(t));
    for (size_t i = 0; i < t; ++i)
        for (size_t j = 0; j < t; ++j) {
            std::vector<uint8_t> d; push_le64(d, i); push_le64(d, j); d.insert(d.end(), seed.begin(), seed.end());
            p.mds[i][j] = fr_from_hash("POSEIDON-MDS", d);
        }
    p.rc_full.assign(rf, std::vector<Fr>(t));
    for (size_t r = 0; r < rf; ++r)
        for (size_t i = 0; i < t; ++i) {
            std::vector<uint8_t> d; push_le64(d, r); push_le64(d, i); d.insert(d.end(), seed.begin(), seed.end());
            p.rc_full[r][i] = fr_from_hash("POSEIDON-RC-FULL", d);
        }
    p.rc_partial.assign(rp, Fr::zero());
    for (size_t r = 0; r < rp; ++r) {
        std::vector<uint8_t> d; push_le64(d, r); d.insert(d.end(), seed.begin(), seed.end());
        p.rc_partial[r] = fr_from_hash("POSEIDON-RC-PART", d);
    }
    return p;
}
static inline std::vector<uint8_t> bytes_of(const char* s) { return std::vector<uint8_t>(s, s + strlen(s)); }

// poseidon/src/lib.rs:318  generate_params_t17_x5(seed): T=17, RF=8, RP=64 (:7-11).
static inline PoseidonParams generate_params_t17_x5(const std::vector<uint8_t>& seed) { return derive_params(seed, 17, 8, 64); }

// poseidon/src/lib.rs:120-146  poseidon_params_for_width; seed_for_t :168-174.
static inline PoseidonParams poseidon_params_for_width(size_t t) {
    size_t rp;
    switch (t) { case 9: rp = 60; break; case 17: rp = 64; break; case 33: rp = 68; break; case 65: rp = 76; break; case 129: rp = 84; break;
                 default: throw std::string("unsupported Poseidon width"); }
    std::vector<uint8_t> seed = bytes_of("POSEIDON-PALLAS-T"); push_le64(seed, t);
    return derive_params(seed, t, 8, rp);
}
// poseidon/src/lib.rs:155-166  poseidon_params_for_arity.
static inline size_t width_for_arity(size_t arity) {
    if (arity <= 8) return 9; if (arity <= 16) return 17; if (arity <= 32) return 33; if (arity <= 64) return 65;
    if (arity <= 128) return 129; throw std::string("unsupported Merkle arity");
}
static inline const PoseidonParams& poseidon_params_for_arity(size_t arity) {
    // cached: the reference re-derives on every MerkleChannelCfg::new (merkle/src/lib.rs:99-106); values identical.
    static PoseidonParams cache[5]; static bool have[5] = {false, false, false, false, false};
    size_t t = width_for_arity(arity); int k = t == 9 ? 0 : t == 17 ? 1 : t == 33 ? 2 : t == 65 ? 3 : 4;
    #pragma omp critical(oracle_params_cache)
    { if (!have[k]) { cache[k] = poseidon_params_for_width(t); have[k] = true; } }
    return cache[k];
}

// poseidon/src/lib.rs:24-29  sbox5.
static inline Fr sbox5(const Fr& x) { Fr x2 = x.square(); Fr x4 = x2.square(); return x * x4; }

// poseidon/src/lib.rs:71-81 / :260-272  dense MDS: out[i] = sum_j mds[i][j]*state[j].
static inline void mds_mul(const PoseidonParams& p, Fr* state) {
    size_t t = p.t; Fr out[129];
    for (size_t i = 0; i < t; ++i) { Fr acc = Fr::zero(); for (size_t j = 0; j < t; ++j) acc += p.mds[i][j] * state[j]; out[i] = acc; }
    for (size_t i = 0; i < t; ++i) state[i] = out[i];
}
// poseidon/src/lib.rs:31-68 (permute, t=17) and :219-258 (permute_dynamic): same round structure.
static inline void permute(Fr* state, const PoseidonParams& p) {
    size_t t = p.t, rf = p.rounds_full, rp = p.rounds_partial, half = rf / 2;
    for (size_t r = 0; r < half; ++r) {
        for (size_t i = 0; i < t; ++i) state[i] += p.rc_full[r][i];
        for (size_t i = 0; i < t; ++i) state[i] = sbox5(state[i]);
        mds_mul(p, state);
    }
    for (size_t r = 0; r < rp; ++r) {
        state[0] += p.rc_partial[r];
        state[0] = sbox5(state[0]);
        mds_mul(p, state);
    }
    for (size_t r = half; r < rf; ++r) {
        for (size_t i = 0; i < t; ++i) state[i] += p.rc_full[r][i];
        for (size_t i = 0; i < t; ++i) state[i] = sbox5(state[i]);
        mds_mul(p, state);
    }
}
// poseidon/src/lib.rs:85-100  hash_with_ds (legacy, t=17, DS tag in the capacity lane, no padding).
static inline Fr hash_with_ds(const Fr* inputs, size_t n, const Fr& ds_tag, const PoseidonParams& p) {
    std::vector<Fr> st(p.t, Fr::zero()); st[p.t - 1] = ds_tag;
    for (size_t off = 0; off < n; off += p.rate) {
        size_t c = (n - off) < p.rate ? (n - off) : p.rate;
        for (size_t i = 0; i < c; ++i) st[i] += inputs[off + i];
        permute(st.data(), p);
    }
    return st[0];
}
// poseidon/src/lib.rs:276-312  absorb_one + hash_with_ds_dynamic (eager permute, pad 1 || 0*).
static inline Fr hash_with_ds_dynamic(const Fr* ds, size_t nds, const Fr* in, size_t n, const PoseidonParams& p) {
    std::vector<Fr> st(p.t, Fr::zero()); size_t cur = 0;
    auto absorb_one = [&](const Fr& x) { st[cur] += x; if (++cur == p.rate) { cur = 0; permute(st.data(), p); } };
    for (size_t i = 0; i < nds; ++i) absorb_one(ds[i]);
    for (size_t i = 0; i < n; ++i) absorb_one(in[i]);
    absorb_one(Fr::from_u64(1));
    while (cur != 0) absorb_one(Fr::zero());
    return st[0];
}

// ---------------------------------------------------------------------------------------------
// transcript/src/lib.rs
// ---------------------------------------------------------------------------------------------
// :13-29 domain_tag_to_field.
static inline Fr domain_tag_to_field(const uint8_t* tag, size_t n) {
    if (n <= 32) { uint8_t le[32]; memset(le, 0, 32); memcpy(le, tag, n); return Fr::from_le_bytes_mod_order(le, 32); }
    Fr acc = Fr::zero();
    for (size_t off = 0; off < n; off += 32) {
        size_t c = (n - off) < 32 ? (n - off) : 32; uint8_t le[32]; memset(le, 0, 32); memcpy(le, tag + off, c);
        acc += Fr::from_le_bytes_mod_order(le, 32);
    }
    return acc;
}
// :32-41 bytes_to_field_words (31-byte LE words).
static inline std::vector<Fr> bytes_to_field_words(const uint8_t* b, size_t n) {
    std::vector<Fr> out;
    for (size_t off = 0; off < n; off += 31) {
        size_t c = (n - off) < 31 ? (n - off) : 31; uint8_t le[32]; memset(le, 0, 32); memcpy(le, b + off, c);
        out.push_back(Fr::from_le_bytes_mod_order(le, 32));
    }
    return out;
}
// :44-46 default_params() — cached here (the reference regenerates it on every call; same values).
static inline const PoseidonParams& transcript_params() {
    static PoseidonParams P; static bool have = false;
    #pragma omp critical(oracle_tparams_cache)
    { if (!have) { P = generate_params_t17_x5(bytes_of("POSEIDON-T17-X5-TRANSCRIPT")); have = true; } }
    return P;
}
// :48-117 Transcript.
struct Transcript {
    Fr state[17]; size_t pos; const PoseidonParams* params;
    Transcript(const char* label, const PoseidonParams& p) : pos(0), params(&p) {
        for (int i = 0; i < 17; ++i) state[i] = Fr::zero();
        state[16] = domain_tag_to_field((const uint8_t*)"FSv1-TRANSCRIPT-INIT", 20);      // :62
        absorb_bytes((const uint8_t*)label, strlen(label));                                // :63
    }
    Transcript(const uint8_t* label, size_t n, const PoseidonParams& p) : pos(0), params(&p) {   // labels with embedded zero bytes
        for (int i = 0; i < 17; ++i) state[i] = Fr::zero();
        state[16] = domain_tag_to_field((const uint8_t*)"FSv1-TRANSCRIPT-INIT", 20);
        absorb_bytes(label, n);
    }
    Fr challenge(const uint8_t* label, size_t n) {                                         // :92-101
        absorb_field(domain_tag_to_field((const uint8_t*)"FSv1-CHALLENGE", 14));
        absorb_bytes(label, n);
        permute(state, *params); pos = 0;
        return state[0];
    }
    void absorb_bytes(const uint8_t* b, size_t n) {                                        // :67-73
        absorb_field(domain_tag_to_field((const uint8_t*)"FSv1-ABSORB-BYTES", 17));
        std::vector<Fr> w = bytes_to_field_words(b, n);
        for (auto& x : w) absorb_field(x);
    }
    void absorb_field(const Fr& x) {                                                       // :79-88 (lazy permute)
        if (pos == 16) { permute(state, *params); pos = 0; }
        state[pos] += x; pos += 1;
    }
    Fr challenge(const char* label) {                                                      // :92-101
        absorb_field(domain_tag_to_field((const uint8_t*)"FSv1-CHALLENGE", 14));
        absorb_bytes((const uint8_t*)label, strlen(label));
        permute(state, *params); pos = 0;
        return state[0];
    }
};

}  // namespace oracle

// oracle/oracle_capi.cpp — TEST INFRASTRUCTURE ONLY.
//
// extern "C" surface of the CPU oracle for ctypes (tests/, __graft_entry__.smoke(), bench.py's
// cpu_baseline leg).  The product library (stark_mlwe_amd/csrc) never links or loads this file.
// Field elements cross this boundary as 4 little-endian u64 limbs in Montgomery form (ark-ff layout).
#include <chrono>
#include <omp.h>
#include <cstdio>
#include <stdexcept>
#include "fri.hpp"
#include "ntt.hpp"
#include "channel.hpp"

using namespace oracle;

static inline Fr ld(const uint64_t* p) { return Fr::from_raw(p); }
static inline void st(uint64_t* p, const Fr& x) { memcpy(p, x.l, 32); }
static inline std::vector<Fr> ldv(const uint64_t* p, size_t n) { std::vector<Fr> v(n); if (n) memcpy((void*)v.data(), p, n * 32); return v; }

// ---- proof decoder (inverse of encode_proof) so the oracle verifier can consume GPU proof bytes ----
struct Dec {
    const uint8_t* p; size_t n, o = 0; bool bad = false;
    uint64_t u64() { if (o + 8 > n) { bad = true; return 0; } uint64_t x = 0; for (int j = 0; j < 8; ++j) x |= (uint64_t)p[o + j] << (8 * j); o += 8; return x; }
    uint8_t u8() { if (o + 1 > n) { bad = true; return 0; } return p[o++]; }
    Fr fr() { if (o + 32 > n) { bad = true; return Fr::zero(); } uint64_t c[4]; for (int i = 0; i < 4; ++i) { c[i] = 0; for (int j = 0; j < 8; ++j) c[i] |= (uint64_t)p[o + 8 * i + j] << (8 * j); }
              o += 32; if (Fr::geq_mod(c)) { bad = true; return Fr::zero(); } return Fr::from_canonical(c); }
    size_t len() { uint64_t x = u64(); if (x > n) { bad = true; return 0; } return (size_t)x; }
    std::vector<size_t> idxs() { size_t k = len(); std::vector<size_t> v; for (size_t i = 0; i < k && !bad; ++i) v.push_back((size_t)u64()); return v; }
    MerkleProof mproof() {
        MerkleProof m; m.indices = idxs();
        size_t a = len(); for (size_t i = 0; i < a && !bad; ++i) { size_t k = len(); std::vector<Fr> l; for (size_t j = 0; j < k && !bad; ++j) l.push_back(fr()); m.siblings.push_back(l); }
        size_t b = len(); for (size_t i = 0; i < b && !bad; ++i) { size_t k = len(); std::vector<uint8_t> l; for (size_t j = 0; j < k && !bad; ++j) l.push_back(u8()); m.group_sizes.push_back(l); }
        m.arity = (size_t)u64(); return m;
    }
};
static bool decode_proof(const uint8_t* b, size_t n, DeepFriProof& p) {
    Dec d{b, n};
    size_t nr = d.len(); for (size_t i = 0; i < nr && !d.bad; ++i) p.roots.push_back(d.fr());
    size_t nl = d.len();
    for (size_t i = 0; i < nl && !d.bad; ++i) { LayerBatchProof lb; { uint8_t h = d.u8(); if (h > 1) d.bad = true; lb.hashed_leaves = h == 1; }   /* bool = one byte, 0 or 1 */ lb.child_indices = d.idxs(); lb.child_proof = d.mproof(); lb.parent_indices = d.idxs(); lb.parent_proof = d.mproof(); p.layer_batches.layers.push_back(lb); }
    p.layer_batches.final_proof = d.mproof();
    size_t nq = d.len();
    for (size_t i = 0; i < nq && !d.bad; ++i) {
        FriQueryPayload q; size_t a = d.len();
        for (size_t j = 0; j < a && !d.bad; ++j) { LayerQueryRef r; r.i = d.u64(); r.child_pos = d.u64(); r.parent_index = d.u64(); r.parent_pos = d.u64(); q.per_layer_refs.push_back(r); }
        size_t c = d.len();
        for (size_t j = 0; j < c && !d.bad; ++j) { LayerOpenPayload pl; pl.f_i = d.fr(); pl.s_i = d.fr(); pl.f_parent_b = d.fr(); pl.s_parent_b = d.fr(); q.per_layer_payloads.push_back(pl); }
        q.final_index = d.u64(); q.final_f = d.fr(); q.final_s = d.fr(); p.queries.push_back(q);
    }
    p.n0 = d.u64(); p.omega0 = d.fr();
    return !d.bad && d.o == n;
}

static const PoseidonParams& params_by_kind(int kind, int t) {
    // kind 0: poseidon_params_for_width(t) (Merkle);  1: transcript (t=17);  2: "POSEIDON-T17-X5-SEED";  3: "POSEIDON-T17-X5" (benches)
    static PoseidonParams k2, k3; static bool h2 = false, h3 = false;
    if (kind == 0) return poseidon_params_for_arity(t - 1);
    if (kind == 1) return transcript_params();
    if (kind == 2) { if (!h2) { k2 = generate_params_t17_x5(bytes_of("POSEIDON-T17-X5-SEED")); h2 = true; } return k2; }
    if (!h3) { k3 = generate_params_t17_x5(bytes_of("POSEIDON-T17-X5")); h3 = true; } return k3;
}

#define TRY try {
#define CATCH } catch (const std::string& e) { fprintf(stderr, "oracle: %s\n", e.c_str()); return -1; } catch (...) { return -2; } return 0;

extern "C" {

int oracle_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); return 0; }

// ---- field ----------------------------------------------------------------------------------------
int oracle_fr_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    // op: 0 add, 1 sub, 2 mul, 3 inverse(a), 4 from_u64(a[0]), 5 to_canonical(a), 6 from_canonical(a), 7 root_of_unity_log(a[0]), 8 pow(a, b as 4 limbs)
    if (field == 0) {
        Fr x = Fr::from_raw(a), y = b ? Fr::from_raw(b) : Fr::zero(), z;
        switch (op) { case 0: z = x + y; break; case 1: z = x - y; break; case 2: z = x * y; break; case 3: z = x.inverse(); break; case 4: z = Fr::from_u64(a[0]); break;
            case 5: x.to_canonical(out); return 0; case 6: z = Fr::from_canonical(a); break; case 7: z = Fr::root_of_unity_log((unsigned)a[0]); break; case 8: z = x.pow(b); break; default: return -1; }
        memcpy(out, z.l, 32); return 0;
    } else {
        FrBls x = FrBls::from_raw(a), y = b ? FrBls::from_raw(b) : FrBls::zero(), z;
        switch (op) { case 0: z = x + y; break; case 1: z = x - y; break; case 2: z = x * y; break; case 3: z = x.inverse(); break; case 4: z = FrBls::from_u64(a[0]); break;
            case 5: x.to_canonical(out); return 0; case 6: z = FrBls::from_canonical(a); break; case 7: z = FrBls::root_of_unity_log((unsigned)a[0]); break; case 8: z = x.pow(b); break; default: return -1; }
        memcpy(out, z.l, 32); return 0;
    }
}
int oracle_fr_from_le_bytes_mod_order(const uint8_t* b, size_t n, uint64_t* out) { st(out, Fr::from_le_bytes_mod_order(b, n)); return 0; }
int oracle_fr_to_bytes_le(const uint64_t* a, uint8_t* out32) { ld(a).to_bytes_le(out32); return 0; }

// ---- BLAKE3 / RNG ------------------------------------------------------------------------------
int oracle_blake3(const uint8_t* data, size_t n, uint8_t* out32) { blake3::hash(data, n, out32); return 0; }
int oracle_stdrng_from_seed_u64s(const uint8_t* seed32, size_t n, uint64_t* out) { StdRng r = StdRng::from_seed(seed32); for (size_t i = 0; i < n; ++i) out[i] = r.next_u64(); return 0; }
int oracle_stdrng_seed_from_u64_u64s(uint64_t seed, size_t n, uint64_t* out) { StdRng r = StdRng::seed_from_u64(seed); for (size_t i = 0; i < n; ++i) out[i] = r.next_u64(); return 0; }
// `(0..n).map(|_| F::rand(&mut rng))` for `ncols` consecutive vectors from one StdRng::seed_from_u64(seed)
// (end_to_end.rs:249-253: a, s, e, t drawn in that order from one rng).
int oracle_rand_fr_columns(uint64_t seed, size_t n, size_t ncols, uint64_t* out) {
    StdRng r = StdRng::seed_from_u64(seed);
    for (size_t i = 0; i < n * ncols; ++i) { Fr x = fr_rand<Fr>(r); st(out + 4 * i, x); }
    return 0;
}
int oracle_synth_column(uint64_t seed, uint64_t col, size_t i0, size_t n, uint64_t* out) { for (size_t i = 0; i < n; ++i) synth_element(seed, col, i0 + i, out + 4 * i); return 0; }

// ---- Poseidon ----------------------------------------------------------------------------------
int oracle_poseidon_params(int kind, int t, uint64_t* mds, uint64_t* rc_full, uint64_t* rc_partial, int* rf, int* rp) {
    TRY const PoseidonParams& p = params_by_kind(kind, t);
    if (rf) *rf = (int)p.rounds_full; if (rp) *rp = (int)p.rounds_partial;
    if (mds) for (size_t i = 0; i < p.t; ++i) for (size_t j = 0; j < p.t; ++j) st(mds + 4 * (i * p.t + j), p.mds[i][j]);
    if (rc_full) for (size_t r = 0; r < p.rounds_full; ++r) for (size_t i = 0; i < p.t; ++i) st(rc_full + 4 * (r * p.t + i), p.rc_full[r][i]);
    if (rc_partial) for (size_t r = 0; r < p.rounds_partial; ++r) st(rc_partial + 4 * r, p.rc_partial[r]);
    CATCH }
int oracle_permute(int kind, int t, uint64_t* states, size_t nstates) {
    TRY const PoseidonParams& p = params_by_kind(kind, t);
    #pragma omp parallel for schedule(static)
    for (long s = 0; s < (long)nstates; ++s) { std::vector<Fr> v = ldv(states + 4 * t * s, t); permute(v.data(), p); memcpy(states + 4 * t * s, v.data(), 32 * t); }
    CATCH }
// hash_with_ds_dynamic, batched: ds4 given as 4 Fr per hash; `cnt` inputs per hash.
int oracle_hash_with_ds_dynamic(int kind, int t, const uint64_t* ds4, const uint64_t* inputs, size_t cnt, size_t n, uint64_t* out) {
    TRY const PoseidonParams& p = params_by_kind(kind, t);
    #pragma omp parallel for schedule(static)
    for (long k = 0; k < (long)n; ++k) { std::vector<Fr> d = ldv(ds4 + 16 * k, 4), in = ldv(inputs + 4 * cnt * k, cnt); st(out + 4 * k, hash_with_ds_dynamic(d.data(), 4, in.data(), cnt, p)); }
    CATCH }
int oracle_hash_with_ds(int kind, const uint64_t* inputs, size_t cnt, const uint64_t* ds_tag, uint64_t* out) {
    TRY const PoseidonParams& p = params_by_kind(kind, 17); std::vector<Fr> in = ldv(inputs, cnt); st(out, hash_with_ds(in.data(), cnt, ld(ds_tag), p)); CATCH }
int oracle_tr_hash_fields_tagged(const char* tag, const uint64_t* fields, size_t n, uint64_t* out) {
    TRY std::vector<Fr> f = ldv(fields, n); st(out, tr_hash_fields_tagged(tag, f.data(), n)); CATCH }
// hash_leaf_pair over a layer: s_i = f_next[i/m] (zero when f_next == NULL, fri.rs:266).
int oracle_leaf_pair_hash(const uint64_t* f, const uint64_t* f_next, size_t n, size_t m, uint64_t* h) {
    TRY
    #pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) { Fr s = f_next ? ld(f_next + 4 * (i / m)) : Fr::zero(); st(h + 4 * i, hash_leaf_pair(ld(f + 4 * i), s)); }
    CATCH }
// Transcript test vector (transcript/src/lib.rs:124-136): new(label) ; absorb_bytes(msg) ; challenge(ch)
int oracle_transcript_vec(const char* label, const char* msg, const char* ch, uint64_t* out) {
    TRY Transcript tr(label, transcript_params()); tr.absorb_bytes((const uint8_t*)msg, strlen(msg)); st(out, tr.challenge(ch)); CATCH }

// ---- Merkle ------------------------------------------------------------------------------------
struct OTree { MerkleTree t; };
// params_kind: 0 => MerkleChannelCfg::new(arity); 2/3 => t=17 params by seed kind (commitment / bench trees).
int oracle_merkle_build(int params_kind, size_t arity, uint64_t tree_label, const uint64_t* leaves, size_t n, int pairs, const uint64_t* cp, void** out) {
    TRY MerkleChannelCfg cfg = (params_kind == 0 ? MerkleChannelCfg::make(arity) : MerkleChannelCfg::with_params(arity, params_by_kind(params_kind, 17))).with_tree_label(tree_label);
    OTree* o = new OTree();
    try { o->t = pairs ? MerkleTree::make_pairs(ldv(leaves, n), ldv(cp, n), cfg) : MerkleTree::make(ldv(leaves, n), cfg); } catch (...) { delete o; throw; }
    *out = o; CATCH }
int oracle_merkle_num_levels(void* h) { return (int)((OTree*)h)->t.levels.size(); }
size_t oracle_merkle_level_len(void* h, int lvl) { return ((OTree*)h)->t.levels[lvl].size(); }
int oracle_merkle_level(void* h, int lvl, uint64_t* out) { auto& l = ((OTree*)h)->t.levels[lvl]; memcpy(out, l.data(), l.size() * 32); return 0; }
int oracle_merkle_root(void* h, uint64_t* out) { st(out, ((OTree*)h)->t.root); return 0; }
// open + verify round trip (verify_many_ds on raw leaves, or verify_pairs_ds when pairs given); returns 1 accept, 0 reject.
int oracle_merkle_open_verify(void* h, const size_t* idx, size_t k, const uint64_t* values, const uint64_t* cp_values, size_t* n_siblings) {
    OTree* o = (OTree*)h; std::vector<size_t> ix(idx, idx + k);
    MerkleProof pr = o->t.open(ix);
    if (n_siblings) { size_t c = 0; for (auto& l : pr.siblings) c += l.size(); *n_siblings = c; }
    if (cp_values) { std::vector<std::pair<Fr, Fr>> ps; for (size_t i = 0; i < k; ++i) ps.push_back({ld(values + 4 * i), ld(cp_values + 4 * i)});
                     return verify_pairs_ds(o->t.root, ix, ps, pr, o->t.cfg.tree_label, o->t.cfg.params) ? 1 : 0; }
    return verify_many_ds(o->t.root, ix, ldv(values, k), pr, o->t.cfg.tree_label, o->t.cfg.params) ? 1 : 0;
}
// open_union_of_paths -> canonical MerkleProof encoding; returns the length (buf may be NULL)
size_t oracle_merkle_open_bytes(void* h, const size_t* idx, size_t k, uint8_t* buf, size_t cap) {
    OTree* o = (OTree*)h; std::vector<size_t> ix(idx, idx + k);
    MerkleProof pr = o->t.open(ix); Enc e; e.mproof(pr); std::vector<uint8_t>& b = e.b;
    if (buf && cap >= b.size()) memcpy(buf, b.data(), b.size());
    return b.size();
}
void oracle_merkle_free(void* h) { delete (OTree*)h; }

// ---- FRI / ALI ---------------------------------------------------------------------------------
int oracle_fri_sample_z_ell(uint64_t seed_z, size_t level, size_t domain_size, uint64_t* out) { TRY st(out, fri_sample_z_ell(seed_z, level, domain_size)); CATCH }
int oracle_fri_fold_layer(const uint64_t* f, size_t n, const uint64_t* z, size_t m, uint64_t* out) {
    TRY std::vector<Fr> o = fri_fold_layer(ldv(f, n), ld(z), m); memcpy(out, o.data(), o.size() * 32); CATCH }
int oracle_compute_s_layer(const uint64_t* f, size_t n, const uint64_t* z, size_t m, uint64_t* out) {
    TRY std::vector<Fr> o = compute_s_layer(ldv(f, n), ld(z), m); memcpy(out, o.data(), o.size() * 32); CATCH }
size_t oracle_pick_arity_for_layer(size_t n, size_t m) { return pick_arity_for_layer(n, m); }
int oracle_domain_omega(size_t n, uint64_t* out) { st(out, domain_omega(n)); return 0; }
int oracle_ali_merge(const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* r_opt, const uint64_t* beta,
                     const uint64_t* omega, const uint64_t* z, size_t n, uint64_t* f0, uint64_t* c_star) {
    TRY std::vector<Fr> rv; if (r_opt) rv = ldv(r_opt, n); Fr cs;
    std::vector<Fr> o = deep_ali_merge_evals_blinded(ldv(a, n), ldv(s, n), ldv(e, n), ldv(t, n), r_opt ? &rv : nullptr, beta ? ld(beta) : Fr::zero(), ld(omega), ld(z), c_star ? &cs : nullptr);
    memcpy(f0, o.data(), n * 32); if (c_star) st(c_star, cs); CATCH }
// DeepAliRealBuilder::build_f0; aux16 (optional) receives col_digest[4], seed_f, z, beta (7 Fr).
int oracle_build_f0(const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, size_t n0, int mock, uint64_t* f0, uint64_t* aux7) {
    TRY BuildF0Aux aux; std::vector<Fr> o = mock ? build_f0_mock(ldv(a, n0), ldv(s, n0), ldv(e, n0), ldv(t, n0), n0) : build_f0_real(ldv(a, n0), ldv(s, n0), ldv(e, n0), ldv(t, n0), n0, &aux);
    memcpy(f0, o.data(), n0 * 32);
    if (aux7 && !mock) { for (int c = 0; c < 4; ++c) st(aux7 + 4 * c, aux.col_digest[c]); st(aux7 + 16, aux.seed_f); st(aux7 + 20, aux.z); st(aux7 + 24, aux.beta); }
    CATCH }

struct OProof { DeepFriProof p; std::vector<uint8_t> bytes; FriProverState st; double secs[4]; };
// deep_fri_prove.  If f0_in != NULL the builder step is skipped (prove "given f0").  keep_state: keep layers for inspection.
int oracle_deep_fri_prove(const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* f0_in, size_t n0,
                          const size_t* schedule, size_t L, size_t r, uint64_t seed_z, void** out) {
    TRY DeepFriParams prm; prm.schedule.assign(schedule, schedule + L); prm.r = r; prm.seed_z = seed_z;
    OProof* o = new OProof();
    try {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<Fr> f0 = f0_in ? ldv(f0_in, n0) : build_f0_real(ldv(a, n0), ldv(s, n0), ldv(e, n0), ldv(t, n0), n0);
        auto t1 = std::chrono::steady_clock::now();
        o->p = deep_fri_prove_from_f0(f0, n0, prm, &o->st);
        auto t2 = std::chrono::steady_clock::now();
        o->bytes = encode_proof(o->p);
        o->secs[0] = std::chrono::duration<double>(t1 - t0).count(); o->secs[1] = std::chrono::duration<double>(t2 - t1).count();
    } catch (...) { delete o; throw; }
    *out = o; CATCH }
size_t oracle_proof_len(void* h) { return ((OProof*)h)->bytes.size(); }
int oracle_proof_bytes(void* h, uint8_t* out) { auto& b = ((OProof*)h)->bytes; memcpy(out, b.data(), b.size()); return 0; }
size_t oracle_proof_size_estimate(void* h) { return deep_fri_proof_size_bytes(((OProof*)h)->p); }
int oracle_proof_num_layers(void* h) { return (int)((OProof*)h)->st.layers.size(); }
int oracle_proof_root(void* h, int layer, uint64_t* out) { st(out, ((OProof*)h)->p.roots[layer]); return 0; }
size_t oracle_proof_layer_len(void* h, int layer) { return ((OProof*)h)->st.f_layers[layer].size(); }
int oracle_proof_layer_f(void* h, int layer, uint64_t* out) { auto& l = ((OProof*)h)->st.f_layers[layer]; memcpy(out, l.data(), l.size() * 32); return 0; }
int oracle_proof_z(void* h, int layer, uint64_t* out) { st(out, ((OProof*)h)->st.z_layers[layer]); return 0; }
double oracle_proof_secs(void* h, int which) { return ((OProof*)h)->secs[which]; }
void oracle_proof_free(void* h) { delete (OProof*)h; }
// deep_fri_verify on encoded bytes: 1 accept, 0 reject, -1 undecodable.
int oracle_deep_fri_verify(const uint8_t* bytes, size_t n, const size_t* schedule, size_t L, size_t r, uint64_t seed_z) {
    DeepFriParams prm; prm.schedule.assign(schedule, schedule + L); prm.r = r; prm.seed_z = seed_z;
    DeepFriProof p; if (!decode_proof(bytes, n, p)) return -1;
    try { return deep_fri_verify(prm, p) ? 1 : 0; } catch (...) { return 0; }
}
// Horner evaluation of sum_j c[j] x^j at k points (the O(n)-per-point DEFINITION of polynomial evaluation: pins sampled outputs
// of the large NTT / LDE / six-step transforms, where the O(n^2) DFT is out of reach).
int oracle_poly_eval_many(int field, const uint64_t* coeffs, size_t n, const uint64_t* points, size_t k, uint64_t* out) {
    TRY
    if (field == 0) {
        #pragma omp parallel for schedule(dynamic)
        for (long p = 0; p < (long)k; ++p) { const Fr x = Fr::from_raw(points + 4 * p); Fr acc = Fr::zero(); for (size_t j = n; j-- > 0;) acc = acc * x + Fr::from_raw(coeffs + 4 * j); memcpy(out + 4 * p, acc.l, 32); }
    } else {
        #pragma omp parallel for schedule(dynamic)
        for (long p = 0; p < (long)k; ++p) { const FrBls x = FrBls::from_raw(points + 4 * p); FrBls acc = FrBls::zero(); for (size_t j = n; j-- > 0;) acc = acc * x + FrBls::from_raw(coeffs + 4 * j); memcpy(out + 4 * p, acc.l, 32); }
    }
    CATCH }

// deep_fri_proof_size_bytes recomputed from encoded bytes.
long oracle_proof_size_estimate_from_bytes(const uint8_t* bytes, size_t n) { DeepFriProof p; if (!decode_proof(bytes, n, p)) return -1; return (long)deep_fri_proof_size_bytes(p); }

// ---- sum-check (N4): prove / verify over the bincode layout of ProofPlain / ProofMF ---------------------------------------
struct BinDec {
    const uint8_t* p; size_t n, o = 0; bool bad = false;
    uint64_t u64() { if (o + 8 > n) { bad = true; return 0; } uint64_t x = 0; for (int j = 0; j < 8; ++j) x |= (uint64_t)p[o + j] << (8 * j); o += 8; return x; }
    size_t len(size_t item) { uint64_t x = u64(); if (bad || (item && x > (n - o) / item)) { bad = true; return 0; } return (size_t)x; }
    Fr fbytes() { if (u64() != 32 || o + 32 > n) { bad = true; return Fr::zero(); } uint64_t c[4]; for (int i = 0; i < 4; ++i) { c[i] = 0; for (int j = 0; j < 8; ++j) c[i] |= (uint64_t)p[o + 8 * i + j] << (8 * j); }
                  o += 32; if (Fr::geq_mod(c)) { bad = true; return Fr::zero(); } return Fr::from_canonical(c); }
    std::vector<size_t> idxs() { size_t k = len(8); std::vector<size_t> v; for (size_t i = 0; i < k && !bad; ++i) v.push_back((size_t)u64()); return v; }
    std::vector<Fr> fvec() { size_t k = len(40); std::vector<Fr> v; for (size_t i = 0; i < k && !bad; ++i) v.push_back(fbytes()); return v; }
    MerkleProof mproof() {
        MerkleProof m; m.arity = (size_t)u64();
        size_t g = len(8); for (size_t i = 0; i < g && !bad; ++i) { size_t k = len(1); std::vector<uint8_t> l; for (size_t j = 0; j < k && !bad; ++j) { if (o >= n) { bad = true; break; } l.push_back(p[o++]); } m.group_sizes.push_back(l); }
        m.indices = idxs();
        size_t a = len(8); for (size_t i = 0; i < a && !bad; ++i) m.siblings.push_back(fvec());
        return m;
    }
};
static bool decode_plain(const uint8_t* b, size_t n, ProofPlain& P) {
    BinDec d{b, n}; P.root = d.fbytes(); size_t k = d.len(80); for (size_t i = 0; i < k && !d.bad; ++i) { Fr c0 = d.fbytes(), c1 = d.fbytes(); P.rounds.push_back({c0, c1}); }
    if (d.o >= n || b[d.o] != 0) return false; d.o += 1;      // extra_openings: None
    P.final_eval = d.fbytes(); return !d.bad && d.o == n;
}
static bool decode_mf(const uint8_t* b, size_t n, ProofMF& P) {
    BinDec d{b, n}; P.initial_root = d.fbytes(); size_t k = d.len(120);
    for (size_t i = 0; i < k && !d.bad; ++i) { RoundMF R; R.c0 = d.fbytes(); R.c1 = d.fbytes(); R.next_root = d.fbytes(); R.cur_indices = d.idxs(); R.cur_values = d.fvec(); R.cur_proof = d.mproof();
        R.next_indices = d.idxs(); R.next_values = d.fvec(); R.next_proof = d.mproof(); P.rounds.push_back(R); }
    P.final_eval = d.fbytes(); return !d.bad && d.o == n;
}
struct OBytes { std::vector<uint8_t> b; };
// variant 0: prove_plain, 1: prove_mf(queries_per_round = q)
int oracle_sumcheck_prove(int variant, size_t k, uint64_t tree_label, size_t q, const uint64_t* witness, void** out) {
    TRY std::vector<Fr> w = ldv(witness, (size_t)1 << k); OBytes* o = new OBytes();
    try { o->b = variant == 0 ? encode_proof_plain(prove_plain(k, tree_label, w)) : encode_proof_mf(prove_mf(k, tree_label, q, w)); } catch (...) { delete o; throw; }
    *out = o; CATCH }
size_t oracle_bytes_len(void* h) { return ((OBytes*)h)->b.size(); }
int oracle_bytes_copy(void* h, uint8_t* out) { auto& b = ((OBytes*)h)->b; memcpy(out, b.data(), b.size()); return 0; }
void oracle_bytes_free(void* h) { delete (OBytes*)h; }
// 1 accept, 0 reject, -1 undecodable
int oracle_sumcheck_verify(int variant, size_t k, uint64_t tree_label, size_t q, const uint8_t* bytes, size_t n) {
    try {
        if (variant == 0) { ProofPlain P; if (!decode_plain(bytes, n, P)) return -1; return verify_plain(k, tree_label, P) ? 1 : 0; }
        ProofMF P; if (!decode_mf(bytes, n, P)) return -1; return verify_mf(k, tree_label, q, P) ? 1 : 0;
    } catch (...) { return 0; }
}
// MerkleCommitment::commit root (commitment/src/lib.rs:85-90)
int oracle_commitment_root(uint64_t tree_label, const uint64_t* leaves, size_t n, uint64_t* root) {
    TRY MerkleTree t = MerkleTree::make(ldv(leaves, n), commitment_tree_cfg(tree_label)); st(root, t.root); CATCH }
// Mle::evaluate (channel/src/lib.rs:279-295)
int oracle_mle_evaluate(const uint64_t* table, size_t k, const uint64_t* r, uint64_t* out) {
    TRY std::vector<Fr> layer = ldv(table, (size_t)1 << k); for (size_t j = 0; j < k; ++j) layer = fold_layer(layer, ld(r + 4 * j)); st(out, layer[0]); CATCH }

// ---- NTT ---------------------------------------------------------------------------------------
int oracle_ntt(int field, uint64_t* data, unsigned log_n, int inverse, const uint64_t* coset) {
    if (field == 0) { if (coset) ntt_coset((Fr*)data, log_n, inverse != 0, Fr::from_raw(coset)); else ntt_radix2((Fr*)data, log_n, inverse != 0); }
    else { if (coset) ntt_coset((FrBls*)data, log_n, inverse != 0, FrBls::from_raw(coset)); else ntt_radix2((FrBls*)data, log_n, inverse != 0); }
    return 0;
}
int oracle_dft_naive(int field, const uint64_t* in, unsigned log_n, int inverse, uint64_t* out) {
    size_t n = (size_t)1 << log_n;
    if (field == 0) { std::vector<Fr> a(n); memcpy((void*)a.data(), in, n * 32); auto o = dft_naive(a, inverse != 0); memcpy(out, o.data(), n * 32); }
    else { std::vector<FrBls> a(n); memcpy((void*)a.data(), in, n * 32); auto o = dft_naive(a, inverse != 0); memcpy(out, o.data(), n * 32); }
    return 0;
}
int oracle_lde(int field, const uint64_t* evals, unsigned log_n, unsigned log_blowup, const uint64_t* coset, uint64_t* out) {
    size_t n = (size_t)1 << log_n, N = n << log_blowup;
    if (field == 0) { std::vector<Fr> a(n); memcpy((void*)a.data(), evals, n * 32); auto o = lde(a, log_n, log_blowup, coset ? Fr::from_raw(coset) : Fr::one()); memcpy(out, o.data(), N * 32); }
    else { std::vector<FrBls> a(n); memcpy((void*)a.data(), evals, n * 32); auto o = lde(a, log_n, log_blowup, coset ? FrBls::from_raw(coset) : FrBls::one()); memcpy(out, o.data(), N * 32); }
    return 0;
}

}  // extern "C"

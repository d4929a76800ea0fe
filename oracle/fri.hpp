// oracle/fri.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// Restatement of crates/deep_ali/src/fri.rs (DEEP-FRI prover/verifier, query phase, size estimator)
// and crates/deep_ali/src/lib.rs:5-105 (DEEP-ALI merge).  Every function cites the lines it follows.
#pragma once
#include <algorithm>
#include <map>
#include <vector>
#include "merkle.hpp"
#include "rng.hpp"

namespace oracle {

// fri.rs:28-35  tr_hash_fields_tagged.
static inline Fr tr_hash_fields_tagged(const char* tag, const Fr* fields, size_t n) {
    Transcript tr("FRI/FS", transcript_params());
    tr.absorb_bytes((const uint8_t*)tag, strlen(tag));
    for (size_t i = 0; i < n; ++i) tr.absorb_field(fields[i]);
    return tr.challenge("out");
}
// fri.rs:38-44  hash_leaf_pair.
static inline Fr hash_leaf_pair(const Fr& f, const Fr& s) {
    Transcript tr("FRI/leaf/poseidon", transcript_params());
    tr.absorb_bytes((const uint8_t*)"FRI/leaf", 8);
    tr.absorb_field(f); tr.absorb_field(s);
    return tr.challenge("leaf");
}
static inline unsigned log2_exact(size_t n) { unsigned k = 0; while (((size_t)1 << k) < n) ++k; return k; }
// fri.rs:53-56  FriDomain::new_radix2: group_gen of Radix2EvaluationDomain::new(size) (size -> next pow2).
static inline Fr domain_omega(size_t size) { return Fr::root_of_unity_log(log2_exact(size)); }

// fri.rs:59-82  fri_sample_z_ell.
static inline Fr fri_sample_z_ell(uint64_t seed_z, size_t level, size_t domain_size) {
    Fr in[3] = {Fr::from_u64(seed_z), Fr::from_u64((uint64_t)level), Fr::from_u64((uint64_t)domain_size)};
    Fr fused = tr_hash_fields_tagged("FRI/z/l", in, 3);
    uint8_t seed[32]; fused.to_bytes_le(seed);
    StdRng rng = StdRng::from_seed(seed);
    size_t tries = 0;
    for (;;) {
        Fr cand = Fr::from_u64(rng.next_u64());
        if (!cand.is_zero() && cand.pow_u64((uint64_t)domain_size) != Fr::one()) return cand;
        if (++tries >= 1000) {
            Fr fb = Fr::from_u64(seed_z + (uint64_t)level + 7);
            if (fb.pow_u64((uint64_t)domain_size) != Fr::one()) return fb;
            return Fr::from_u64(11);
        }
    }
}
// fri.rs:85-102  fri_fold_layer.
static inline std::vector<Fr> fri_fold_layer(const std::vector<Fr>& f, const Fr& z, size_t m) {
    if (m < 2) throw std::string("m >= 2");
    if (f.size() % m) throw std::string("layer size must be divisible by m");
    size_t nn = f.size() / m; std::vector<Fr> out(nn), zp(m);
    Fr acc = Fr::one(); for (size_t t = 0; t < m; ++t) { zp[t] = acc; acc *= z; }
    #pragma omp parallel for schedule(static)
    for (long b = 0; b < (long)nn; ++b) { Fr s = Fr::zero(); for (size_t t = 0; t < m; ++t) s += f[b * m + t] * zp[t]; out[b] = s; }
    return out;
}
// fri.rs:123-143  compute_s_layer.
static inline std::vector<Fr> compute_s_layer(const std::vector<Fr>& f, const Fr& z, size_t m) {
    std::vector<Fr> bucket = fri_fold_layer(f, z, m), s(f.size());
    for (size_t i = 0; i < f.size(); ++i) s[i] = bucket[i / m];
    return s;
}
// fri.rs:220-229  pick_arity_for_layer.
static inline size_t pick_arity_for_layer(size_t n, size_t m) {
    if (m >= 128 && n % 128 == 0) return 128; if (m >= 64 && n % 64 == 0) return 64; if (m >= 32 && n % 32 == 0) return 32;
    if (m >= 16 && n % 16 == 0) return 16; if (m >= 8 && n % 8 == 0) return 8; if (m >= 4 && n % 4 == 0) return 4;
    if (n % 2 == 0) return 2; return 1;
}
static inline bool hashed_arity(size_t a) { return a == 128 || a == 64 || a == 32 || a == 16 || a == 8; }  // fri.rs:275

// fri.rs:193-216
struct FriLayerCommitment { size_t n, m; Fr root; std::vector<Fr> f, s; bool hashed_leaves; MerkleTree tree; };
struct FriProverState { std::vector<std::vector<Fr>> f_layers, s_layers; std::vector<size_t> schedule; std::vector<FriLayerCommitment> layers;
                        std::vector<Fr> omega_layers, z_layers; };

// fri.rs:231-312  fri_build_transcript.
static inline FriProverState fri_build_transcript(const std::vector<Fr>& f0, size_t n0, const std::vector<size_t>& schedule, uint64_t seed_z) {
    FriProverState st; st.schedule = schedule; size_t L = schedule.size();
    std::vector<Fr> cur = f0; size_t cur_size = n0; st.f_layers.push_back(cur);
    for (size_t ell = 0; ell < L; ++ell) {
        size_t m = schedule[ell];
        if (cur_size % m) throw std::string("schedule not dividing domain size");
        Fr z = fri_sample_z_ell(seed_z, ell, cur_size);
        st.z_layers.push_back(z); st.omega_layers.push_back(domain_omega(cur_size));
        cur = fri_fold_layer(cur, z, m); cur_size /= m; st.f_layers.push_back(cur);
    }
    for (size_t ell = 0; ell < L; ++ell) st.s_layers.push_back(compute_s_layer(st.f_layers[ell], st.z_layers[ell], schedule[ell]));
    st.s_layers.push_back(std::vector<Fr>(st.f_layers[L].size(), Fr::zero()));                       // :266
    for (size_t ell = 0; ell <= L; ++ell) {
        size_t n = st.f_layers[ell].size(), m_ell = ell < L ? schedule[ell] : 1;
        size_t arity = pick_arity_for_layer(n, m_ell); bool hashed = hashed_arity(arity);
        MerkleChannelCfg cfg = MerkleChannelCfg::make(arity).with_tree_label((uint64_t)ell);         // :277
        FriLayerCommitment lc; lc.n = n; lc.m = m_ell; lc.f = st.f_layers[ell]; lc.s = st.s_layers[ell]; lc.hashed_leaves = hashed;
        if (hashed) {
            std::vector<Fr> h(n);
            #pragma omp parallel for schedule(static)
            for (long i = 0; i < (long)n; ++i) h[i] = hash_leaf_pair(st.f_layers[ell][i], st.s_layers[ell][i]);   // :283
            lc.tree = MerkleTree::make(h, cfg);
        } else {
            lc.tree = MerkleTree::make_pairs(st.f_layers[ell], st.s_layers[ell], cfg);                            // :289
        }
        lc.root = lc.tree.root;
        st.layers.push_back(std::move(lc));
    }
    return st;
}

// fri.rs:178-191  fs_seed_from_roots, index_from_seed, index_seed.
static inline Fr fs_seed_from_roots(const std::vector<Fr>& roots) { return tr_hash_fields_tagged("FRI/seed", roots.data(), roots.size()); }
static inline size_t index_from_seed(const Fr& seed_f, size_t n_pow2) {
    uint8_t seed[32]; seed_f.to_bytes_le(seed); StdRng rng = StdRng::from_seed(seed);
    return (size_t)rng.next_u64() & (n_pow2 - 1);
}
static inline Fr index_seed(const Fr& roots_seed, size_t ell, size_t q) {
    Fr in[3] = {roots_seed, Fr::from_u64((uint64_t)ell), Fr::from_u64((uint64_t)q)};
    return tr_hash_fields_tagged("FRI/index", in, 3);
}
static inline size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }

// fri.rs:314-346
struct LayerBatchProof { bool hashed_leaves; std::vector<size_t> child_indices; MerkleProof child_proof; std::vector<size_t> parent_indices; MerkleProof parent_proof; };
struct LayerQueryRef { size_t i, child_pos, parent_index, parent_pos; };
struct LayerOpenPayload { Fr f_i, s_i, f_parent_b, s_parent_b; };
struct FriQueryPayload { std::vector<LayerQueryRef> per_layer_refs; std::vector<LayerOpenPayload> per_layer_payloads; size_t final_index; Fr final_f, final_s; };
struct FriLayerBatches { std::vector<LayerBatchProof> layers; MerkleProof final_proof; };
// fri.rs:589-599
struct DeepFriParams { std::vector<size_t> schedule; size_t r; uint64_t seed_z; };
struct DeepFriProof { std::vector<Fr> roots; FriLayerBatches layer_batches; std::vector<FriQueryPayload> queries; size_t n0; Fr omega0; };

// fri.rs:355-466  fri_prove_queries  +  fri.rs:620-638 payload assembly (deep_fri_prove tail).
static inline void fri_prove_queries(const FriProverState& st, size_t r, const Fr& roots_seed, DeepFriProof& proof) {
    size_t L = st.schedule.size();
    std::vector<std::vector<LayerQueryRef>> refs(r);
    std::vector<std::vector<size_t>> child_b(L), parent_b(L);
    for (size_t q = 0; q < r; ++q)
        for (size_t ell = 0; ell < L; ++ell) {
            const FriLayerCommitment& layer = st.layers[ell];
            size_t n = layer.n, n_pow2 = next_pow2(n), m = layer.m;
            Fr seed = index_seed(roots_seed, ell, q);
            size_t i0 = index_from_seed(seed, n_pow2), i;
            if (i0 < n) i = i0;
            else { Fr in[2] = {seed, Fr::from_u64(1)}; Fr reseed = tr_hash_fields_tagged("FRI/index", in, 2);
                   size_t i2 = index_from_seed(reseed, n_pow2); i = i2 < n ? i2 : (i2 & (n - 1)); }
            size_t b = i / m;
            child_b[ell].push_back(i); parent_b[ell].push_back(b);
            refs[q].push_back(LayerQueryRef{i, 0, b, 0});
        }
    for (size_t ell = 0; ell < L; ++ell) {
        const FriLayerCommitment& lay = st.layers[ell];
        std::vector<size_t> ci = child_b[ell]; std::sort(ci.begin(), ci.end()); ci.erase(std::unique(ci.begin(), ci.end()), ci.end());
        std::vector<size_t> pi = parent_b[ell]; std::sort(pi.begin(), pi.end()); pi.erase(std::unique(pi.begin(), pi.end()), pi.end());
        LayerBatchProof lb; lb.hashed_leaves = lay.hashed_leaves; lb.child_indices = ci; lb.child_proof = lay.tree.open(ci);
        lb.parent_indices = pi; lb.parent_proof = st.layers[ell + 1].tree.open(pi);
        for (size_t q = 0; q < r; ++q) {
            LayerQueryRef& rr = refs[q][ell];
            rr.child_pos = (size_t)(std::lower_bound(ci.begin(), ci.end(), rr.i) - ci.begin());
            rr.parent_pos = (size_t)(std::lower_bound(pi.begin(), pi.end(), rr.parent_index) - pi.begin());
        }
        proof.layer_batches.layers.push_back(std::move(lb));
    }
    const FriLayerCommitment& last = st.layers[L];
    proof.layer_batches.final_proof = last.tree.open(std::vector<size_t>{0});
    for (size_t q = 0; q < r; ++q) {
        FriQueryPayload qp; qp.per_layer_refs = refs[q]; qp.final_index = 0; qp.final_f = last.f[0]; qp.final_s = last.s[0];
        for (size_t ell = 0; ell < L; ++ell) {
            const LayerQueryRef& rr = refs[q][ell];
            qp.per_layer_payloads.push_back(LayerOpenPayload{st.layers[ell].f[rr.i], st.layers[ell].s[rr.i],
                                                             st.layers[ell + 1].f[rr.parent_index], st.layers[ell + 1].s[rr.parent_index]});
        }
        proof.queries.push_back(std::move(qp));
    }
}

// ---------------------------------------------------------------------------------------------
// deep_ali/src/lib.rs
// ---------------------------------------------------------------------------------------------
// lib.rs:17-45  lagrange_eval_on_h (barycentric; n inversions as in the reference).
static inline Fr lagrange_eval_on_h(const std::vector<Fr>& values, const Fr& z, const Fr& omega) {
    size_t n = values.size();
    if (z.pow_u64((uint64_t)n) == Fr::one()) {
        Fr w = Fr::one();
        for (size_t j = 0; j < n; ++j) { if (z == w) return values[j]; w *= omega; }
        throw std::string("z in domain but not matching a power of omega");
    }
    Fr zh = z.pow_u64((uint64_t)n) - Fr::one();
    Fr n_inv = Fr::from_u64((uint64_t)n).inverse();
    // Same sum as the reference loop (:39-43); evaluated with a batch inversion — identical field values.
    std::vector<Fr> wj(n), den(n), pre(n);
    Fr w = Fr::one(); for (size_t j = 0; j < n; ++j) { wj[j] = w; den[j] = z - w; w *= omega; }
    Fr acc = Fr::one(); for (size_t j = 0; j < n; ++j) { pre[j] = acc; acc *= den[j]; }
    Fr inv = acc.inverse(); Fr sum = Fr::zero();
    for (size_t j = n; j-- > 0;) { Fr dinv = inv * pre[j]; inv *= den[j]; sum += values[j] * wj[j] * dinv; }
    return zh * n_inv * sum;
}
// lib.rs:60-105  deep_ali_merge_evals_blinded (r_eval may be empty => None).  Returns f0; c_star out.
static inline std::vector<Fr> deep_ali_merge_evals_blinded(const std::vector<Fr>& a, const std::vector<Fr>& s, const std::vector<Fr>& e,
                                                           const std::vector<Fr>& t, const std::vector<Fr>* r_eval, const Fr& beta,
                                                           const Fr& omega, const Fr& z, Fr* c_star_out) {
    size_t n = a.size();
    if (n <= 1) throw std::string("n > 1");
    if (z.pow_u64((uint64_t)n) == Fr::one()) throw std::string("z must be outside H");
    std::vector<Fr> phi(n);
    for (size_t i = 0; i < n; ++i) { Fr base = a[i] * s[i] + e[i] - t[i]; phi[i] = r_eval ? base + beta * (*r_eval)[i] : base; }
    if (c_star_out) {
        Fr phi_z = lagrange_eval_on_h(phi, z, omega);
        Fr zh = z.pow_u64((uint64_t)n) - Fr::one();
        *c_star_out = phi_z * zh.inverse();
    }
    // f0[j] = phi[j] / (omega^j - z)  (:97-102), batch-inverted (same field values).
    std::vector<Fr> den(n), pre(n), f0(n);
    Fr w = Fr::one(); for (size_t j = 0; j < n; ++j) { den[j] = w - z; w *= omega; }
    Fr acc = Fr::one(); for (size_t j = 0; j < n; ++j) { pre[j] = acc; acc *= den[j]; }
    Fr inv = acc.inverse();
    for (size_t j = n; j-- > 0;) { Fr dinv = inv * pre[j]; inv *= den[j]; f0[j] = phi[j] * dinv; }
    return f0;
}
// fri.rs:511-533  ali_sample_z_beta_fs.
static inline void ali_sample_z_beta_fs(const char* tag, size_t n0, const Fr& roots_seed, Fr& z, Fr& beta) {
    Fr in[2] = {roots_seed, Fr::from_u64((uint64_t)n0)};
    Fr fused = tr_hash_fields_tagged(tag, in, 2);
    uint8_t seed[32]; fused.to_bytes_le(seed); StdRng rng = StdRng::from_seed(seed);
    beta = Fr::from_u64(rng.next_u64());
    size_t tries = 0;
    for (;;) {
        Fr cand = Fr::from_u64(rng.next_u64());
        if (!cand.is_zero() && cand.pow_u64((uint64_t)n0) != Fr::one()) { z = cand; return; }
        if (++tries >= 1000) {
            Fr fb = roots_seed + Fr::from_u64(17);
            if (fb.pow_u64((uint64_t)n0) != Fr::one()) { z = fb; return; }
            z = Fr::from_u64(19); return;
        }
    }
}
// fri.rs:535-569  DeepAliRealBuilder::build_f0 (default: no blinding, ds_tag "ALI/DEEP").  Also
// returns the FS point z and the four column digests for stage-level parity checks.
struct BuildF0Aux { Fr col_digest[4]; Fr seed_f, z, beta; };
static inline std::vector<Fr> build_f0_real(const std::vector<Fr>& a, const std::vector<Fr>& s, const std::vector<Fr>& e, const std::vector<Fr>& t,
                                            size_t n0, BuildF0Aux* aux = nullptr) {
    Fr h[5];
    const std::vector<Fr>* cols[4] = {&a, &s, &e, &t}; const char* tags[4] = {"ALI/A", "ALI/S", "ALI/E", "ALI/T"};
    #pragma omp parallel for num_threads(4)
    for (int c = 0; c < 4; ++c) h[c] = tr_hash_fields_tagged(tags[c], cols[c]->data(), cols[c]->size());   // :551-554 (serial sponges)
    h[4] = Fr::from_u64((uint64_t)n0);
    Fr seed_f = tr_hash_fields_tagged("ALI/seed", h, 5);
    Fr z, beta; ali_sample_z_beta_fs("ALI/DEEP", n0, seed_f, z, beta);
    if (aux) { for (int c = 0; c < 4; ++c) aux->col_digest[c] = h[c]; aux->seed_f = seed_f; aux->z = z; aux->beta = beta; }
    return deep_ali_merge_evals_blinded(a, s, e, t, nullptr, Fr::zero(), domain_omega(n0), z, nullptr);
}
// fri.rs:484-495  DeepAliMock::build_f0.
static inline std::vector<Fr> build_f0_mock(const std::vector<Fr>& a, const std::vector<Fr>& s, const std::vector<Fr>& e, const std::vector<Fr>& t, size_t n0) {
    Fr h[5] = {tr_hash_fields_tagged("ALI/a", a.data(), a.size()), tr_hash_fields_tagged("ALI/s", s.data(), s.size()),
               tr_hash_fields_tagged("ALI/e", e.data(), e.size()), tr_hash_fields_tagged("ALI/t", t.data(), t.size()), Fr::from_u64((uint64_t)n0)};
    Fr seed_f = tr_hash_fields_tagged("ALI/mock/seed", h, 5);
    uint8_t seed[32]; seed_f.to_bytes_le(seed); StdRng rng = StdRng::from_seed(seed);
    std::vector<Fr> out(n0); for (size_t i = 0; i < n0; ++i) out[i] = Fr::from_u64(rng.next_u64());
    return out;
}

// fri.rs:601-641  deep_fri_prove (with f0 supplied; callers compose with build_f0_*).
static inline DeepFriProof deep_fri_prove_from_f0(const std::vector<Fr>& f0, size_t n0, const DeepFriParams& params, FriProverState* st_out = nullptr) {
    FriProverState st = fri_build_transcript(f0, n0, params.schedule, params.seed_z);
    DeepFriProof proof; proof.n0 = n0; proof.omega0 = domain_omega(n0);
    for (auto& l : st.layers) proof.roots.push_back(l.root);
    Fr roots_seed = fs_seed_from_roots(proof.roots);
    fri_prove_queries(st, params.r, roots_seed, proof);
    if (st_out) *st_out = std::move(st);
    return proof;
}
static inline DeepFriProof deep_fri_prove(const std::vector<Fr>& a, const std::vector<Fr>& s, const std::vector<Fr>& e, const std::vector<Fr>& t,
                                          size_t n0, const DeepFriParams& params) {
    return deep_fri_prove_from_f0(build_f0_real(a, s, e, t, n0), n0, params);
}

// fri.rs:643-762  deep_fri_verify.
static inline bool deep_fri_verify(const DeepFriParams& params, const DeepFriProof& proof) {
    size_t L = params.schedule.size();
    if (proof.roots.size() != L + 1 || proof.layer_batches.layers.size() != L || proof.queries.size() != params.r) return false;
    std::vector<size_t> sizes; { size_t n = proof.n0; sizes.push_back(n); for (size_t m : params.schedule) { if (n % m) return false; n /= m; sizes.push_back(n); } }
    std::vector<std::map<size_t, std::pair<Fr, Fr>>> cm(L), pm(L);
    for (size_t q = 0; q < params.r; ++q) {
        const FriQueryPayload& qp = proof.queries[q];
        if (qp.per_layer_refs.size() != L || qp.per_layer_payloads.size() != L) return false;
        for (size_t ell = 0; ell < L; ++ell) {
            const LayerQueryRef& rr = qp.per_layer_refs[ell]; const LayerOpenPayload& pay = qp.per_layer_payloads[ell];
            cm[ell].insert({rr.i, {pay.f_i, pay.s_i}}); pm[ell].insert({rr.parent_index, {pay.f_parent_b, pay.s_parent_b}});
        }
    }
    auto check = [&](size_t layer, size_t n, size_t m_req, const Fr& root, const std::vector<size_t>& idx, const MerkleProof& mp,
                     const std::map<size_t, std::pair<Fr, Fr>>& vals) -> bool {
        size_t ar = pick_arity_for_layer(n, m_req); bool hashed = hashed_arity(ar);
        MerkleChannelCfg cfg = MerkleChannelCfg::make(ar).with_tree_label((uint64_t)layer);
        if (hashed) {
            std::vector<Fr> lv; for (size_t i : idx) { auto it = vals.find(i); if (it == vals.end()) return false; lv.push_back(hash_leaf_pair(it->second.first, it->second.second)); }
            return verify_many_ds(root, idx, lv, mp, cfg.tree_label, cfg.params);
        }
        std::vector<std::pair<Fr, Fr>> pr; for (size_t i : idx) { auto it = vals.find(i); if (it == vals.end()) return false; pr.push_back(it->second); }
        return verify_pairs_ds(root, idx, pr, mp, cfg.tree_label, cfg.params);
    };
    for (size_t ell = 0; ell < L; ++ell) {
        const LayerBatchProof& lb = proof.layer_batches.layers[ell];
        if (!check(ell, sizes[ell], params.schedule[ell], proof.roots[ell], lb.child_indices, lb.child_proof, cm[ell])) return false;
        if (!check(ell + 1, sizes[ell + 1], ell + 1 < L ? params.schedule[ell + 1] : 1, proof.roots[ell + 1], lb.parent_indices, lb.parent_proof, pm[ell])) return false;
    }
    for (size_t q = 0; q < params.r; ++q)                                       // :724-738 local fold checks
        for (size_t ell = 0; ell < L; ++ell) {
            const LayerQueryRef& rr = proof.queries[q].per_layer_refs[ell]; const LayerOpenPayload& pay = proof.queries[q].per_layer_payloads[ell];
            size_t b = rr.i / params.schedule[ell];
            if (b >= sizes[ell] / params.schedule[ell]) return false;
            if (pay.s_i != pay.f_parent_b) return false;
        }
    {   // :740-759 final layer opening at index 0
        if (proof.queries[0].final_index != 0) return false;
        std::map<size_t, std::pair<Fr, Fr>> v; v[0] = {proof.queries[0].final_f, proof.queries[0].final_s};
        if (!check(L, sizes[L], 1, proof.roots[L], std::vector<size_t>{0}, proof.layer_batches.final_proof, v)) return false;
    }
    return true;
}

// fri.rs:764-805  deep_fri_proof_size_bytes (FR_BYTES=32, INDEX_BYTES=8).
static inline size_t merkle_proof_size_bytes(const MerkleProof& mp) { size_t t = 0; for (auto& g : mp.siblings) t += g.size() * 32; return t; }
static inline size_t deep_fri_proof_size_bytes(const DeepFriProof& p) {
    size_t total = p.roots.size() * 32 + 32 + 8;
    for (auto& lb : p.layer_batches.layers) {
        total += merkle_proof_size_bytes(lb.child_proof) + merkle_proof_size_bytes(lb.parent_proof);
        total += lb.child_indices.size() * 8 + lb.parent_indices.size() * 8;
    }
    total += merkle_proof_size_bytes(p.layer_batches.final_proof);
    for (auto& q : p.queries) total += 8 + 2 * 32 + q.per_layer_refs.size() * 16 + q.per_layer_payloads.size() * 128;
    return total;
}

// ---------------------------------------------------------------------------------------------
// Canonical proof encoding: the reference has NO serializer, so the build
// defines one and applies it to both the oracle and the GPU path.  Fields in declaration order of
// DeepFriProof (fri.rs:591-599); F = 32-byte canonical LE; usize/len = u64 LE; bool/u8 = 1 byte;
// Vec = u64 length prefix.
// ---------------------------------------------------------------------------------------------
struct Enc {
    std::vector<uint8_t> b;
    void u64(uint64_t x) { for (int j = 0; j < 8; ++j) b.push_back((uint8_t)(x >> (8 * j))); }
    void u8(uint8_t x) { b.push_back(x); }
    void fr(const Fr& x) { uint8_t t[32]; x.to_bytes_le(t); b.insert(b.end(), t, t + 32); }
    void idxs(const std::vector<size_t>& v) { u64(v.size()); for (size_t x : v) u64(x); }
    void mproof(const MerkleProof& p) {
        idxs(p.indices);
        u64(p.siblings.size()); for (auto& l : p.siblings) { u64(l.size()); for (auto& x : l) fr(x); }
        u64(p.group_sizes.size()); for (auto& l : p.group_sizes) { u64(l.size()); for (uint8_t x : l) u8(x); }
        u64(p.arity);
    }
};
static inline std::vector<uint8_t> encode_proof(const DeepFriProof& p) {
    Enc e;
    e.u64(p.roots.size()); for (auto& r : p.roots) e.fr(r);
    e.u64(p.layer_batches.layers.size());
    for (auto& lb : p.layer_batches.layers) { e.u8(lb.hashed_leaves ? 1 : 0); e.idxs(lb.child_indices); e.mproof(lb.child_proof); e.idxs(lb.parent_indices); e.mproof(lb.parent_proof); }
    e.mproof(p.layer_batches.final_proof);
    e.u64(p.queries.size());
    for (auto& q : p.queries) {
        e.u64(q.per_layer_refs.size()); for (auto& r : q.per_layer_refs) { e.u64(r.i); e.u64(r.child_pos); e.u64(r.parent_index); e.u64(r.parent_pos); }
        e.u64(q.per_layer_payloads.size()); for (auto& pl : q.per_layer_payloads) { e.fr(pl.f_i); e.fr(pl.s_i); e.fr(pl.f_parent_b); e.fr(pl.s_parent_b); }
        e.u64(q.final_index); e.fr(q.final_f); e.fr(q.final_s);
    }
    e.u64(p.n0); e.fr(p.omega0);
    return e.b;
}

}  // namespace oracle

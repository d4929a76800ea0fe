// oracle/channel.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle).  "Next" row N4 of SURVEY.md §8(f): the sum-check consumer of the
// Merkle / Poseidon / transcript path, restated from crates/channel/src/lib.rs:
//   ProverChannel / VerifierChannel          :7-117      MerkleCommitment (arity 16, "POSEIDON-T17-X5-SEED")  commitment/src/lib.rs:60-114
//   Mle, sumcheck_round_coeffs               :252-296, 406-416
//   SumCheckProver / Verifier (plain)        :418-541    SumCheckMFProver / Verifier (Merkle-folded)          :547-879
//   prove_plain / verify_plain               :1045-1128  prove_mf / verify_mf                                 :1130-1240
// The reference holds no golden values for this path (its tests are round trips, :1246-1452, restated in tests/), so proof BYTES are
// pinned only through the primitives underneath (the CSV fingerprint of oracle/fingerprint.cpp covers Poseidon, both sponges, the
// transcript and the Merkle tree) — "parity unpinned" for the sum-check messages themselves.
// Proof encoding = bincode 1.x (fixint, little endian) of the reference's serde structs ProofPlain / ProofMF (:925-979), which is
// what its bench measures (benches/end_to_end.rs:145-146): FBytes = u64 length (32) + 32-byte compressed Fr; Vec = u64 count;
// usize = u64; Option = one tag byte.
#pragma once
#include <set>
#include <string>
#include <vector>
#include "fr.hpp"
#include "poseidon.hpp"
#include "merkle.hpp"

namespace oracle {

static inline const PoseidonParams& commitment_default_params() {                       // commitment/src/lib.rs:48-51
    static PoseidonParams P; static bool have = false;
    #pragma omp critical(oracle_commit_params_cache)
    { if (!have) { P = generate_params_t17_x5(bytes_of("POSEIDON-T17-X5-SEED")); have = true; } }
    return P;
}
static inline MerkleChannelCfg commitment_tree_cfg(uint64_t ds_tag) {                    // commitment/src/lib.rs:65-73
    return MerkleChannelCfg::with_params(16, commitment_default_params()).with_tree_label(ds_tag);
}
static inline void le64(std::vector<uint8_t>& v, uint64_t x) { for (int j = 0; j < 8; ++j) v.push_back((uint8_t)(x >> (8 * j))); }

// ProverChannel / VerifierChannel share one transcript flow (:7-117)
struct Channel {
    Transcript tr;
    Channel(const char* label) : tr(label, transcript_params()) {}
    void bytes(const char* s) { tr.absorb_bytes((const uint8_t*)s, strlen(s)); }
    void send_digest(const char* label, const Fr& d) { bytes("CHAN/SEND/DIGEST"); bytes(label); tr.absorb_field(d); }            // :22-26
    Fr challenge_scalar(const std::vector<uint8_t>& label) { return tr.challenge(label.data(), label.size()); }                   // :28-30
    void send_opening(const std::vector<size_t>& indices, const std::vector<Fr>& values, const MerkleProof& proof) {               // :32-62
        bytes("CHAN/SEND/OPEN");
        for (size_t i : indices) { std::vector<uint8_t> b; le64(b, (uint64_t)i); tr.absorb_bytes(b.data(), 8); }
        for (auto& v : values) tr.absorb_field(v);
        bytes("PROOF/ARITY"); { std::vector<uint8_t> b; le64(b, (uint64_t)proof.arity); tr.absorb_bytes(b.data(), 8); }
        bytes("PROOF/GROUP_SIZES");
        for (auto& lvl : proof.group_sizes) { std::vector<uint8_t> b; le64(b, (uint64_t)lvl.size()); tr.absorb_bytes(b.data(), 8); for (uint8_t sz : lvl) tr.absorb_bytes(&sz, 1); }
        bytes("PROOF/SIBLINGS");
        for (auto& lvl : proof.siblings) { std::vector<uint8_t> b; le64(b, (uint64_t)lvl.size()); tr.absorb_bytes(b.data(), 8); for (auto& s : lvl) tr.absorb_field(s); }
    }
};

static inline void sumcheck_round_coeffs(const std::vector<Fr>& layer, Fr& c0, Fr& c1) {   // :406-416
    c0 = Fr::zero(); c1 = Fr::zero();
    for (size_t i = 0; i + 1 < layer.size(); i += 2) { c0 += layer[i]; c1 += layer[i + 1] - layer[i]; }
}
static inline std::vector<Fr> fold_layer(const std::vector<Fr>& layer, const Fr& r) {       // :456-462 / :640-647
    const Fr one_minus = Fr::one() - r; std::vector<Fr> next(layer.size() / 2);
    #pragma omp parallel for schedule(static)
    for (long j = 0; j < (long)next.size(); ++j) next[j] = one_minus * layer[2 * j] + r * layer[2 * j + 1];
    return next;
}
static inline std::vector<uint8_t> label_idx(const char* base, uint64_t i) { std::vector<uint8_t> l((const uint8_t*)base, (const uint8_t*)base + strlen(base)); le64(l, i); return l; }

// ---- proofs (the reference's serde structs, :925-979) ---------------------------------------------------------------
struct ProofPlain { Fr root; std::vector<std::pair<Fr, Fr>> rounds; Fr final_eval; };
struct RoundMF { Fr c0, c1, next_root; std::vector<size_t> cur_indices; std::vector<Fr> cur_values; MerkleProof cur_proof; std::vector<size_t> next_indices; std::vector<Fr> next_values; MerkleProof next_proof; };
struct ProofMF { Fr initial_root; std::vector<RoundMF> rounds; Fr final_eval; };

// prove_plain (:1045-1076).  vk = (k, tree_label).
static inline ProofPlain prove_plain(size_t k, uint64_t tree_label, const std::vector<Fr>& witness) {
    if (witness.size() != ((size_t)1 << k)) throw std::string("MLE length must be 2^k");
    Channel ch("E2E/PLAIN");
    MerkleTree tree = MerkleTree::make(witness, commitment_tree_cfg(tree_label));         // MerkleProver::commit_vector (:172-179)
    ch.send_digest("commit/root", tree.root);
    std::vector<Fr> layer = witness;                                                      // SumCheckProver::new (:429-432)
    Fr s = Fr::zero(); for (auto& v : layer) s += v;                                      // send_claim (:434-446)
    ch.bytes("SUMCHECK/CLAIM"); ch.tr.absorb_field(s);
    ProofPlain P; P.root = tree.root;
    for (size_t i = 0; i < k; ++i) {                                                      // round (:448-472)
        Fr c0, c1; sumcheck_round_coeffs(layer, c0, c1);
        ch.bytes("SUMCHECK/ROUND"); { std::vector<uint8_t> b; le64(b, (uint64_t)i); ch.tr.absorb_bytes(b.data(), 8); }
        ch.bytes("COEFF/c0"); ch.tr.absorb_field(c0); ch.bytes("COEFF/c1"); ch.tr.absorb_field(c1);
        const Fr r = ch.challenge_scalar(label_idx("sumcheck/r", i));
        layer = fold_layer(layer, r);
        P.rounds.push_back({c0, c1});
    }
    ch.bytes("SUMCHECK/FINAL/EVAL"); ch.tr.absorb_field(layer[0]);                         // :474-484
    P.final_eval = layer[0];
    return P;
}
// verify_plain (:1080-1128): false on a failed check (the reference asserts inside SumCheckVerifier::round / finalize_and_check: a
// failed assert_eq! is a panic there; callers see "not accepted" either way).
static inline bool verify_plain(size_t k, uint64_t tree_label, const ProofPlain& P) {
    (void)tree_label;
    Channel ch("E2E/PLAIN");
    ch.send_digest("commit/root", P.root);                                                 // receive_root (:213-216)
    if (P.rounds.empty()) return false;                                                    // :1100-1102
    const Fr two = Fr::from_u64(2);
    Fr s0 = two * P.rounds[0].first + P.rounds[0].second;
    ch.bytes("SUMCHECK/CLAIM"); ch.tr.absorb_field(s0);
    Fr running = s0;
    for (size_t i = 0; i < P.rounds.size(); ++i) {                                         // the loop runs over proof.rounds, not vk.k (:1111)
        const Fr& c0 = P.rounds[i].first; const Fr& c1 = P.rounds[i].second;
        ch.bytes("SUMCHECK/ROUND"); { std::vector<uint8_t> b; le64(b, (uint64_t)i); ch.tr.absorb_bytes(b.data(), 8); }
        ch.bytes("COEFF/c0"); ch.tr.absorb_field(c0); ch.bytes("COEFF/c1"); ch.tr.absorb_field(c1);
        if (!(two * c0 + c1 == running)) return false;                                     // :511-512
        const Fr r = ch.challenge_scalar(label_idx("sumcheck/r", i));
        running = c0 + c1 * r;
    }
    ch.bytes("SUMCHECK/FINAL/EVAL"); ch.tr.absorb_field(P.final_eval);
    (void)k;
    return P.final_eval == running;                                                        // :528
}

// mf_round_challenge_from_root (:592-598)
static inline Fr mf_round_challenge_from_root(size_t round_idx, const Fr& prev_root) {
    Transcript tmp("SUMCHECK-MF/ROUND-CHAL", transcript_params());
    tmp.absorb_bytes((const uint8_t*)"SUMCHECK/MF/R", 13);
    { std::vector<uint8_t> b; le64(b, (uint64_t)round_idx); tmp.absorb_bytes(b.data(), 8); }
    tmp.absorb_field(prev_root);
    return tmp.challenge("r_i");
}
// query index from a challenge (:667-676): XOR of the four 64-bit limbs of the canonical value, mod half
static inline size_t mf_query_index(const Fr& r, size_t half) { uint64_t c[4]; r.to_canonical(c); return (size_t)((c[0] ^ c[1] ^ c[2] ^ c[3]) % (uint64_t)half); }

// prove_mf (:1130-1172)
static inline ProofMF prove_mf(size_t k, uint64_t tree_label, size_t queries_per_round, const std::vector<Fr>& witness) {
    if (witness.size() != ((size_t)1 << k)) throw std::string("MLE length must be 2^k");
    Channel ch("E2E/MF");
    const MerkleChannelCfg cfg = commitment_tree_cfg(tree_label);
    MerkleTree cur_tree = MerkleTree::make(witness, cfg);                                  // SumCheckMFProver::new (:601-622)
    ch.send_digest("sumcheck-mf/root/0", cur_tree.root);
    std::vector<Fr> cur = witness;
    ProofMF P; P.initial_root = cur_tree.root;
    { Fr s = Fr::zero(); for (auto& v : cur) s += v; ch.bytes("SUMCHECK/MF/CLAIM"); ch.tr.absorb_field(s); }    // send_claim (:624-629)
    for (size_t i = 0; i < k; ++i) {                                                       // round (:631-737)
        RoundMF R; sumcheck_round_coeffs(cur, R.c0, R.c1);
        ch.bytes("SUMCHECK/MF/ROUND"); { std::vector<uint8_t> b; le64(b, (uint64_t)i); ch.tr.absorb_bytes(b.data(), 8); }
        ch.bytes("COEFF/c0"); ch.tr.absorb_field(R.c0); ch.bytes("COEFF/c1"); ch.tr.absorb_field(R.c1);
        const Fr r = mf_round_challenge_from_root(i, cur_tree.root);
        const size_t half = cur.size() / 2;
        std::vector<Fr> next = fold_layer(cur, r);
        MerkleTree next_tree = MerkleTree::make(next, cfg);
        ch.send_digest("sumcheck-mf/root/next", next_tree.root);
        const size_t q_target = std::min(std::max(queries_per_round, (size_t)1), half);    // :656
        std::set<size_t> set; size_t attempt = 0, j = 0; const size_t max_attempts = std::max(q_target * 16, (size_t)16);
        while (set.size() < q_target && attempt < max_attempts) {
            std::vector<uint8_t> ql((const uint8_t*)"sumcheck-mf/q", (const uint8_t*)"sumcheck-mf/q" + 13); le64(ql, (uint64_t)i); le64(ql, (uint64_t)j);
            const Fr rr = ch.challenge_scalar(ql);
            if (half > 0) set.insert(mf_query_index(rr, half));
            ++j; ++attempt;
        }
        if (set.size() < q_target) for (size_t idx = 0; idx < half && set.size() < q_target; ++idx) set.insert(idx);     // :683-690
        std::vector<size_t> queries(set.begin(), set.end());
        for (size_t jj : queries) { R.cur_indices.push_back(2 * jj); R.cur_indices.push_back(2 * jj + 1); }
        for (size_t ix : R.cur_indices) R.cur_values.push_back(cur[ix]);
        R.cur_proof = cur_tree.open(R.cur_indices);
        R.next_indices = queries; for (size_t ix : queries) R.next_values.push_back(next[ix]);
        R.next_proof = next_tree.open(R.next_indices);
        ch.send_opening(R.cur_indices, R.cur_values, R.cur_proof);
        ch.send_opening(R.next_indices, R.next_values, R.next_proof);
        R.next_root = next_tree.root;
        cur.swap(next); cur_tree = next_tree;
        P.rounds.push_back(R);
    }
    ch.bytes("SUMCHECK/MF/FINAL/EVAL"); ch.tr.absorb_field(cur[0]);                         // finalize_eval (:739-745)
    P.final_eval = cur[0];
    return P;
}
// verify_mf (:1176-1240)
static inline bool verify_mf(size_t k, uint64_t tree_label, size_t queries_per_round, const ProofMF& P) {
    (void)k; (void)queries_per_round;
    Channel ch("E2E/MF");
    const MerkleChannelCfg cfg = commitment_tree_cfg(tree_label);
    ch.send_digest("sumcheck-mf/root/0", P.initial_root);                                   // receive_initial_root (:778-781)
    const Fr two = Fr::from_u64(2);
    bool have_running = false; Fr running = Fr::zero(); Fr prev_root = P.initial_root;
    for (size_t i = 0; i < P.rounds.size(); ++i) {
        const RoundMF& R = P.rounds[i];
        const Fr s_prev = have_running ? running : two * R.c0 + R.c1;                        // :1206
        ch.bytes("SUMCHECK/MF/ROUND"); { std::vector<uint8_t> b; le64(b, (uint64_t)i); ch.tr.absorb_bytes(b.data(), 8); }
        ch.bytes("COEFF/c0"); ch.tr.absorb_field(R.c0); ch.bytes("COEFF/c1"); ch.tr.absorb_field(R.c1);
        if (!(two * R.c0 + R.c1 == s_prev)) return false;                                    // start_round (:803-804)
        const Fr r = mf_round_challenge_from_root(i, prev_root);                             // derive_round_challenge (:807-810) — cur_root == prev_root here
        ch.send_digest("sumcheck-mf/root/next", R.next_root);                                // recv_next_root (:812-815)
        // verify_fold_openings (:821-869): note that the verifier does NOT absorb the openings into its channel
        if (!verify_many_ds(prev_root, R.cur_indices, R.cur_values, R.cur_proof, cfg.tree_label, cfg.params)) return false;
        if (!verify_many_ds(R.next_root, R.next_indices, R.next_values, R.next_proof, cfg.tree_label, cfg.params)) return false;
        if (R.cur_indices.size() != R.cur_values.size() || R.next_indices.size() != R.next_values.size()) return false;
        std::map<size_t, std::pair<std::pair<bool, Fr>, std::pair<bool, Fr>>> pairs;
        for (size_t t = 0; t < R.cur_indices.size(); ++t) { size_t ix = R.cur_indices[t], jx = ix / 2; if (ix % 2 == 0) pairs[jx].first = {true, R.cur_values[t]}; else pairs[jx].second = {true, R.cur_values[t]}; }
        const Fr one_minus = Fr::one() - r;
        for (size_t t = 0; t < R.next_indices.size(); ++t) {
            auto it = pairs.find(R.next_indices[t]);
            if (it == pairs.end() || !it->second.first.first || !it->second.second.first) return false;
            if (!(one_minus * it->second.first.second + r * it->second.second.second == R.next_values[t])) return false;
        }
        running = R.c0 + R.c1 * r; have_running = true; prev_root = R.next_root;
    }
    ch.bytes("SUMCHECK/MF/FINAL/EVAL"); ch.tr.absorb_field(P.final_eval);
    return P.final_eval == (have_running ? running : P.final_eval);                          // :1237-1238
}

// ---- bincode layout of the serde structs ----------------------------------------------------------------------------
struct BinEnc {
    std::vector<uint8_t> b;
    void u64(uint64_t x) { for (int j = 0; j < 8; ++j) b.push_back((uint8_t)(x >> (8 * j))); }
    void fbytes(const Fr& x) { u64(32); uint8_t t[32]; x.to_bytes_le(t); b.insert(b.end(), t, t + 32); }
    void idxs(const std::vector<size_t>& v) { u64(v.size()); for (size_t x : v) u64(x); }
    void mproof(const MerkleProof& p) {                                                      // MerkleProofBytes { arity, group_sizes, indices, siblings } (:974-979)
        u64(p.arity);
        u64(p.group_sizes.size()); for (auto& l : p.group_sizes) { u64(l.size()); for (uint8_t x : l) b.push_back(x); }
        idxs(p.indices);
        u64(p.siblings.size()); for (auto& l : p.siblings) { u64(l.size()); for (auto& x : l) fbytes(x); }
    }
};
static inline std::vector<uint8_t> encode_proof_plain(const ProofPlain& P) {
    BinEnc e; e.fbytes(P.root); e.u64(P.rounds.size()); for (auto& r : P.rounds) { e.fbytes(r.first); e.fbytes(r.second); }
    e.b.push_back(0);                                                                        // extra_openings: None
    e.fbytes(P.final_eval); return e.b;
}
static inline std::vector<uint8_t> encode_proof_mf(const ProofMF& P) {
    BinEnc e; e.fbytes(P.initial_root); e.u64(P.rounds.size());
    for (auto& R : P.rounds) {
        e.fbytes(R.c0); e.fbytes(R.c1); e.fbytes(R.next_root);
        e.idxs(R.cur_indices); e.u64(R.cur_values.size()); for (auto& v : R.cur_values) e.fbytes(v); e.mproof(R.cur_proof);
        e.idxs(R.next_indices); e.u64(R.next_values.size()); for (auto& v : R.next_values) e.fbytes(v); e.mproof(R.next_proof);
    }
    e.fbytes(P.final_eval); return e.b;
}

}  // namespace oracle

// stark_mlwe_amd/csrc/fri_verify.hpp — the verifier side of the path as host logic over abstract batch hashers (product code):
//   deep_fri_verify                 crates/deep_ali/src/fri.rs:643-762
//   verify_many_ds / verify_pairs_ds crates/merkle/src/lib.rs:587-722, 723-773  (MerkleProver::verify_single / verify_pairs, :800-855)
// over the canonical proof encoding of DESIGN.md §7 (the reference's DeepFriProof has no serialisation of its own).
//
// Written against the Rust text, independently of oracle/ (the oracle's verifier is the CHECKER of this one in tests/).
// The decisions follow the reference statement by statement — including what it does NOT check (query indices are taken
// from the proof, not re-derived; child_pos / parent_pos are unused; the fold relation itself is not recomputed: only
// s_i == f_parent[b], fri.rs:168-176) — so accept/reject agrees with `deep_fri_verify` on every input the reference answers.
// Where the reference would PANIC on malformed input (index out of bounds on an empty query list, division by a zero arity,
// a schedule that does not divide n0) this code answers `false`.
// All hashing goes through `VerifyHasher`, which the library implements with its batched GPU kernels (capi_verify.hip) and the
// CPU diagnostic build with the host instantiation of the same kernel bodies (hostcheck.cpp).  Host-only C++.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>
#include "fr.hpp"
#include "host_util.hpp"
#include "fri_plan.hpp"

namespace stark {

struct VerifyHasher {
    virtual ~VerifyHasher() {}
    // out[i] = hash_leaf_pair(f[i], s[i])                                                        fri.rs:38-44
    virtual int32_t leaf_pairs(const fr_t* f, const fr_t* s, size_t n, fr_t* out) = 0;
    // out[k] = hash_with_ds_dynamic([arity, level, positions[k], label], children[k*chunk .. (k+1)*chunk), params(arity))   merkle/src/lib.rs:683-689
    virtual int32_t ds_nodes(size_t arity, size_t chunk, uint32_t level, uint64_t label, const uint64_t* positions, const fr_t* children, size_t n, fr_t* out) = 0;
    // out[k] = hash_with_ds_dynamic([arity, 2^32-1, positions[k], label], [f[k], cp[k]], params(arity))                      merkle/src/lib.rs:757-766
    virtual int32_t ds_pair_leaves(size_t arity, uint64_t label, const uint64_t* positions, const fr_t* f, const fr_t* cp, size_t n, fr_t* out) = 0;
};

// ---- decoding (bounds-checked: the bytes are untrusted) ---------------------------------------------------------------
struct ByteReader {
    const uint8_t* p; size_t n, pos = 0; bool ok = true;
    ByteReader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
    size_t left() const { return n - pos; }
    uint64_t u64() { if (!ok || left() < 8) { ok = false; return 0; } uint64_t x = 0; for (int j = 0; j < 8; ++j) x |= (uint64_t)p[pos + j] << (8 * j); pos += 8; return x; }
    uint8_t u8() { if (!ok || left() < 1) { ok = false; return 0; } return p[pos++]; }
    // a length prefix of items of `item_bytes` each: must fit in what is left (so a forged length cannot drive an allocation)
    size_t len(size_t item_bytes) { uint64_t k = u64(); if (!ok || (item_bytes && k > left() / item_bytes)) { ok = false; return 0; } return (size_t)k; }
    fr_t fr() {    // 32-byte canonical little-endian; a value >= r is not a field element
        fr_t z = fr_zero<PallasFr>(); if (!ok || left() < 32) { ok = false; return z; }
        uint32_t t[9]; for (int i = 0; i < 8; ++i) t[i] = (uint32_t)p[pos + 4 * i] | ((uint32_t)p[pos + 4 * i + 1] << 8) | ((uint32_t)p[pos + 4 * i + 2] << 16) | ((uint32_t)p[pos + 4 * i + 3] << 24);
        t[8] = 0;
        if (fr_geq_p<PallasFr>(t)) { ok = false; return z; }
        z = host::h_from_le_bytes_mod_order(p + pos, 32); pos += 32; return z;
    }
};
inline bool dec_idxs(ByteReader& R, std::vector<size_t>& v) { size_t k = R.len(8); v.resize(k); for (size_t i = 0; i < k; ++i) v[i] = (size_t)R.u64(); return R.ok; }
inline bool dec_mproof(ByteReader& R, MerkleProofHost& p) {
    if (!dec_idxs(R, p.indices)) return false;
    size_t nl = R.len(8); p.siblings.assign(nl, {});
    for (size_t l = 0; l < nl && R.ok; ++l) { size_t k = R.len(32); p.siblings[l].resize(k); for (size_t i = 0; i < k; ++i) p.siblings[l][i] = R.fr(); }
    size_t ng = R.len(8); p.group_sizes.assign(ng, {});
    for (size_t l = 0; l < ng && R.ok; ++l) { size_t k = R.len(1); p.group_sizes[l].resize(k); for (size_t i = 0; i < k; ++i) p.group_sizes[l][i] = R.u8(); }
    p.arity = (size_t)R.u64();
    return R.ok;
}
struct LayerBatchHost { bool hashed = false; std::vector<size_t> child_indices, parent_indices; MerkleProofHost child_proof, parent_proof; };      // fri.rs:316-325
struct QueryRefHost { size_t i, child_pos, parent_index, parent_pos; };                                                                          // fri.rs:328-334
struct QueryPayHost { fr_t f_i, s_i, f_parent_b, s_parent_b; };                                                                                  // fri.rs:573-578
struct QueryHost { std::vector<QueryRefHost> refs; std::vector<QueryPayHost> pays; size_t final_index = 0; fr_t final_f, final_s; };              // fri.rs:581-586
struct DeepFriProofHost { std::vector<fr_t> roots; std::vector<LayerBatchHost> layers; MerkleProofHost final_proof; std::vector<QueryHost> queries; size_t n0 = 0; fr_t omega0; };   // fri.rs:591-599
inline bool decode_proof(const uint8_t* bytes, size_t len, DeepFriProofHost& P) {
    ByteReader R(bytes, len);
    size_t nr = R.len(32); P.roots.resize(nr); for (size_t i = 0; i < nr; ++i) P.roots[i] = R.fr();
    size_t L = R.len(1); P.layers.assign(L, LayerBatchHost());
    for (size_t l = 0; l < L && R.ok; ++l) {
        LayerBatchHost& b = P.layers[l]; uint8_t h = R.u8(); if (h > 1) R.ok = false; b.hashed = h == 1;
        if (!dec_idxs(R, b.child_indices) || !dec_mproof(R, b.child_proof) || !dec_idxs(R, b.parent_indices) || !dec_mproof(R, b.parent_proof)) return false;
    }
    if (!R.ok || !dec_mproof(R, P.final_proof)) return false;
    size_t r = R.len(8); P.queries.assign(r, QueryHost());
    for (size_t q = 0; q < r && R.ok; ++q) {
        QueryHost& Q = P.queries[q];
        size_t k = R.len(32); Q.refs.resize(k); for (size_t l = 0; l < k; ++l) { Q.refs[l].i = (size_t)R.u64(); Q.refs[l].child_pos = (size_t)R.u64(); Q.refs[l].parent_index = (size_t)R.u64(); Q.refs[l].parent_pos = (size_t)R.u64(); }
        size_t k2 = R.len(128); Q.pays.resize(k2); for (size_t l = 0; l < k2; ++l) { Q.pays[l].f_i = R.fr(); Q.pays[l].s_i = R.fr(); Q.pays[l].f_parent_b = R.fr(); Q.pays[l].s_parent_b = R.fr(); }
        Q.final_index = (size_t)R.u64(); Q.final_f = R.fr(); Q.final_s = R.fr();
    }
    P.n0 = (size_t)R.u64(); P.omega0 = R.fr();
    return R.ok && R.left() == 0;
}

inline bool ok_width_for(size_t arity) { return arity >= 1 && arity <= 128; }      // the dynamic params ARE poseidon_params_for_arity(arity) (MerkleChannelCfg::new), so the t / arity guard (:605-614) reduces to the arity range

// verify_many_ds (merkle/src/lib.rs:587-722) with cfg = MerkleChannelCfg::new(cfg_arity).with_tree_label(label):
// the Poseidon parameters belong to the VERIFIER's arity (fri.rs:676-678), the hashing arity comes from the proof (:602).
inline int32_t verify_many_ds_host(VerifyHasher& H, size_t cfg_arity, const fr_t& root, const std::vector<size_t>& indices, const std::vector<fr_t>& values, const MerkleProofHost& proof, uint64_t label, bool& ok) {
    ok = false;
    if (indices.empty() || indices.size() != values.size()) return 0;                                        // :595-597
    std::vector<size_t> req = indices; std::sort(req.begin(), req.end()); req.erase(std::unique(req.begin(), req.end()), req.end());
    if (proof.indices != req) return 0;                                                                      // :601-603
    if (proof.siblings.size() != proof.group_sizes.size()) return 0;                                         // :604-606
    const size_t arity = proof.arity;
    if (arity == 0) return 0;                                                                                // (the reference divides by it)
    if (host::width_for_arity(arity) != host::width_for_arity(cfg_arity) || host::width_for_arity(cfg_arity) < 0) return 0;   // ok_width, :610-618
    std::map<size_t, fr_t> mp; for (size_t k = 0; k < indices.size(); ++k) mp[indices[k]] = values[k];      // later entries win (:622-625)
    std::vector<size_t> cur_idx = req; std::vector<fr_t> cur_val; for (size_t i : cur_idx) cur_val.push_back(mp[i]);
    for (size_t level = 0; level < proof.siblings.size(); ++level) {
        const std::vector<fr_t>& sib = proof.siblings[level]; const std::vector<uint8_t>& gs = proof.group_sizes[level];
        // groups by parent, in parent order (cur_idx is sorted, so consecutive runs)
        std::vector<size_t> parents; std::vector<std::vector<std::pair<size_t, fr_t>>> opened;
        for (size_t k = 0; k < cur_idx.size(); ++k) {
            const size_t par = cur_idx[k] / arity, cpos = cur_idx[k] % arity;
            if (parents.empty() || parents.back() != par) { parents.push_back(par); opened.emplace_back(); }
            opened.back().push_back({cpos, cur_val[k]});
        }
        if (parents.size() != gs.size()) return 0;                                                           // :641-643
        // children of every group: opened where the position matches, siblings otherwise (:656-676)
        size_t off = 0; std::vector<std::vector<fr_t>> kids(parents.size());
        for (size_t g = 0; g < parents.size(); ++g) {
            const size_t cc = gs[g]; if (cc == 0 || cc > arity) return 0;                                     // :653-655
            size_t o = 0;                                                                                     // opened[g] is sorted by cpos already
            for (size_t c = 0; c < cc; ++c) {
                if (o < opened[g].size() && opened[g][o].first == c) { kids[g].push_back(opened[g][o].second); ++o; continue; }
                if (off >= sib.size()) return 0;
                kids[g].push_back(sib[off++]);
            }
        }
        if (off != sib.size()) return 0;                                                                      // :696-698
        // hash the groups in batches of equal child count (at most two distinct counts in an honest proof)
        std::vector<fr_t> nv(parents.size());
        std::vector<char> done(parents.size(), 0);
        for (size_t g0 = 0; g0 < parents.size(); ++g0) {
            if (done[g0]) continue;
            const size_t cc = kids[g0].size(); std::vector<size_t> members; std::vector<uint64_t> pos; std::vector<fr_t> ch;
            for (size_t g = g0; g < parents.size(); ++g) if (!done[g] && kids[g].size() == cc) { done[g] = 1; members.push_back(g); pos.push_back((uint64_t)parents[g]); ch.insert(ch.end(), kids[g].begin(), kids[g].end()); }
            std::vector<fr_t> outv(members.size());
            int32_t rc = H.ds_nodes(arity, cc, (uint32_t)level, label, pos.data(), ch.data(), members.size(), outv.data()); if (rc) return rc;
            for (size_t k = 0; k < members.size(); ++k) nv[members[k]] = outv[k];
        }
        cur_idx = parents; cur_val = nv;
    }
    if (cur_val.size() != 1) return 0;                                                                        // :704-706
    ok = fr_eq(cur_val[0], root); return 0;
}
// verify_pairs_ds (merkle/src/lib.rs:723-773)
inline int32_t verify_pairs_ds_host(VerifyHasher& H, size_t cfg_arity, const fr_t& root, const std::vector<size_t>& indices, const std::vector<fr_t>& f, const std::vector<fr_t>& cp, const MerkleProofHost& proof, uint64_t label, bool& ok) {
    ok = false;
    if (indices.size() != f.size() || indices.size() != cp.size() || indices.empty()) return 0;              // :731-733
    const size_t arity = proof.arity;
    if (arity == 0 || host::width_for_arity(cfg_arity) < 0 || host::width_for_arity(arity) != host::width_for_arity(cfg_arity)) return 0;   // :737-745
    std::vector<size_t> req = indices; std::sort(req.begin(), req.end()); req.erase(std::unique(req.begin(), req.end()), req.end());
    std::map<size_t, std::pair<fr_t, fr_t>> mp; for (size_t k = 0; k < indices.size(); ++k) mp[indices[k]] = {f[k], cp[k]};
    std::vector<fr_t> ff, cc; std::vector<uint64_t> pos; for (size_t i : req) { ff.push_back(mp[i].first); cc.push_back(mp[i].second); pos.push_back((uint64_t)i); }
    std::vector<fr_t> leaves(req.size());
    { int32_t rc = H.ds_pair_leaves(arity, label, pos.data(), ff.data(), cc.data(), req.size(), leaves.data()); if (rc) return rc; }
    return verify_many_ds_host(H, cfg_arity, root, req, leaves, proof, label, ok);
}

// deep_fri_verify (fri.rs:643-762)
inline int32_t deep_fri_verify_host(VerifyHasher& H, const DeepFriProofHost& P, const size_t* schedule, size_t L, size_t r, bool& ok) {
    ok = false;
    if (P.roots.size() != L + 1 || P.layers.size() != L || P.queries.size() != r) return 0;                  // :645-647
    std::vector<size_t> sizes(1, P.n0);
    for (size_t l = 0; l < L; ++l) { if (schedule[l] == 0 || sizes.back() % schedule[l]) return 0; sizes.push_back(sizes.back() / schedule[l]); }   // layer_sizes_from_schedule (:144-154) asserts
    std::vector<std::map<size_t, std::pair<fr_t, fr_t>>> cmap(L), pmap(L);
    for (size_t q = 0; q < r; ++q) {
        const QueryHost& Q = P.queries[q];
        if (Q.refs.size() != L || Q.pays.size() != L) return 0;                                               // :657-659
        for (size_t l = 0; l < L; ++l) {                                                                      // entry().or_insert: the FIRST payload for an index wins (:663-664)
            cmap[l].insert({Q.refs[l].i, {Q.pays[l].f_i, Q.pays[l].s_i}});
            pmap[l].insert({Q.refs[l].parent_index, {Q.pays[l].f_parent_b, Q.pays[l].s_parent_b}});
        }
    }
    auto check_opening = [&](size_t layer, size_t m_req, const std::vector<size_t>& idx, const std::map<size_t, std::pair<fr_t, fr_t>>& mp, const MerkleProofHost& pr, bool& good) -> int32_t {
        good = false;
        const size_t ar = pick_arity_for_layer(sizes[layer], m_req); const bool hashed = hashed_arity(ar);     // :671-673
        std::vector<fr_t> ff, ss;
        for (size_t i : idx) { auto it = mp.find(i); if (it == mp.end()) return 0; ff.push_back(it->second.first); ss.push_back(it->second.second); }   // :677-680
        if (hashed) {
            std::vector<fr_t> lh(idx.size()); if (!idx.empty()) { int32_t rc = H.leaf_pairs(ff.data(), ss.data(), idx.size(), lh.data()); if (rc) return rc; }
            return verify_many_ds_host(H, ar, P.roots[layer], idx, lh, pr, (uint64_t)layer, good);           // verify_single, :682
        }
        return verify_pairs_ds_host(H, ar, P.roots[layer], idx, ff, ss, pr, (uint64_t)layer, good);           // verify_pairs, :691
    };
    for (size_t l = 0; l < L; ++l) {
        const LayerBatchHost& lb = P.layers[l]; bool good = false;
        { int32_t rc = check_opening(l, schedule[l], lb.child_indices, cmap[l], lb.child_proof, good); if (rc) return rc; if (!good) return 0; }
        { int32_t rc = check_opening(l + 1, l + 1 < L ? schedule[l + 1] : 1, lb.parent_indices, pmap[l], lb.parent_proof, good); if (rc) return rc; if (!good) return 0; }
    }
    for (size_t q = 0; q < r; ++q) for (size_t l = 0; l < L; ++l) {                                            // verify_local_check_fold (:168-176): s_i == f_parent[b]
        const QueryRefHost& rf = P.queries[q].refs[l]; const QueryPayHost& py = P.queries[q].pays[l];
        const size_t b = rf.i / schedule[l];
        if (b >= sizes[l] / schedule[l]) return 0;
        if (!fr_eq(py.s_i, py.f_parent_b)) return 0;
    }
    if (r == 0) return 0;                                                                                     // proof.queries[0] (:742) would panic
    if (P.queries[0].final_index != 0) return 0;                                                              // :743
    {
        std::map<size_t, std::pair<fr_t, fr_t>> one; one[0] = {P.queries[0].final_f, P.queries[0].final_s};
        bool good = false; int32_t rc = check_opening(L, 1, std::vector<size_t>{0}, one, P.final_proof, good); if (rc) return rc; if (!good) return 0;
    }
    ok = true; return 0;
}

}  // namespace stark

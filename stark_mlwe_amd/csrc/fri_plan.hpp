// stark_mlwe_amd/csrc/fri_plan.hpp — the query phase of deep_fri_prove as host logic over an abstract
// value source: fri_prove_queries + payload assembly + canonical encoding
// (crates/deep_ali/src/fri.rs:355-466, 613-640; open_union_of_paths, crates/merkle/src/lib.rs:246-315).
//
// Everything the query phase reads from the commit phase is (i) the L+1 roots and the layer shapes,
// which fix every index, and (ii) a few thousand field elements / digests AT those indices.  The code
// below therefore asks a `FriSource` for "layer l, elements idx[]" and "tree l, level v, nodes idx[]":
//   * single GPU     : the source gathers from the device-resident layers and tree levels;
//   * several GPUs   : the layers and the lower tree levels are block-sharded over the ranks.  The same
//                      code runs once against a RECORDING source (returns zeros, logs the requests — the
//                      requests do not depend on the values), the orchestrator (stark_mlwe_amd/dist.py)
//                      lets every rank fill in what it owns and combines them with one all-reduce, and
//                      the code runs again against a REPLAY source.  No callbacks cross the C-ABI.
// Host-only C++ (no HIP), so the CPU diagnostic build (hostcheck.cpp) compiles the identical logic.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>
#include "fr.hpp"
#include "host_util.hpp"

namespace stark {

// MerkleProof (merkle/src/lib.rs:131-143) on the host side of the product + canonical encoders.
struct MerkleProofHost { std::vector<size_t> indices; std::vector<std::vector<fr_t>> siblings; std::vector<std::vector<uint8_t>> group_sizes; size_t arity = 0; };
inline void enc_u64(std::vector<uint8_t>& b, uint64_t x) { for (int j = 0; j < 8; ++j) b.push_back((uint8_t)(x >> (8 * j))); }
inline void enc_fr(std::vector<uint8_t>& b, const fr_t& x) { uint8_t t[32]; host::h_to_bytes_le(x, t); b.insert(b.end(), t, t + 32); }
inline void enc_idxs(std::vector<uint8_t>& b, const std::vector<size_t>& v) { enc_u64(b, v.size()); for (size_t x : v) enc_u64(b, x); }
inline void enc_mproof(std::vector<uint8_t>& b, const MerkleProofHost& p) {
    enc_idxs(b, p.indices);
    enc_u64(b, p.siblings.size()); for (auto& l : p.siblings) { enc_u64(b, l.size()); for (auto& x : l) enc_fr(b, x); }
    enc_u64(b, p.group_sizes.size()); for (auto& l : p.group_sizes) { enc_u64(b, l.size()); for (uint8_t x : l) b.push_back(x); }
    enc_u64(b, p.arity);
}

inline size_t pick_arity_for_layer(size_t n, size_t m) {    // fri.rs:220-229
    if (m >= 128 && n % 128 == 0) return 128; if (m >= 64 && n % 64 == 0) return 64; if (m >= 32 && n % 32 == 0) return 32;
    if (m >= 16 && n % 16 == 0) return 16; if (m >= 8 && n % 8 == 0) return 8; if (m >= 4 && n % 4 == 0) return 4;
    if (n % 2 == 0) return 2; return 1;
}
inline bool hashed_arity(size_t a) { return a == 128 || a == 64 || a == 32 || a == 16 || a == 8; }   // fri.rs:275
inline int ilog2_ceil(size_t x) { int k = 0; while (((size_t)1 << k) < x) ++k; return k; }

// Shapes of the L+1 committed layers: sizes, Merkle arities, leaf kinds, level lengths; plus the roots.
struct FriShape {
    size_t n0 = 0; std::vector<size_t> schedule, n, arity; std::vector<char> hashed;
    std::vector<std::vector<size_t>> lens;           // lens[l][v] = number of nodes of tree l at level v (level 0 = leaf digests)
    std::vector<fr_t> roots;
    bool make(size_t n0_, const size_t* sched, size_t L, const fr_t* roots_, std::string& err) {
        if (!n0_) { err = "empty layer"; return false; }
        n0 = n0_; schedule.assign(sched, sched + L); n.assign(1, n0);
        for (size_t l = 0; l < L; ++l) { if (sched[l] < 2 || n.back() % sched[l]) { err = "schedule not dividing domain size"; return false; } n.push_back(n.back() / sched[l]); }   // fri.rs:150
        arity.clear(); hashed.clear(); lens.clear();
        for (size_t l = 0; l <= L; ++l) {
            size_t a = pick_arity_for_layer(n[l], l < L ? sched[l] : 1); arity.push_back(a); hashed.push_back(hashed_arity(a) ? 1 : 0);
            std::vector<size_t> lv(1, n[l]); while (lv.back() > 1) lv.push_back((lv.back() + a - 1) / a);                       // merkle/src/lib.rs:166-190
            if (a < 2 && n[l] > 1) { err = "layer with arity 1"; return false; }
            lens.push_back(lv);
        }
        roots.assign(roots_, roots_ + L + 1);
        return true;
    }
};

struct FriSource {
    virtual ~FriSource() {}
    virtual int32_t layer(size_t l, const std::vector<size_t>& idx, std::vector<fr_t>& out) = 0;                     // out[i] = f_l[idx[i]]
    virtual int32_t digests(size_t tree, size_t level, const std::vector<size_t>& idx, std::vector<fr_t>& out) = 0;  // out[i] = levels[level][idx[i]] of tree l
};
// tr_hash_fields_tagged (fri.rs:28-35) over n independent inputs of k fields each.
struct TrHasher {
    virtual ~TrHasher() {}
    virtual int32_t hash(const char* tag, const fr_t* fields, size_t k, size_t n, fr_t* out) = 0;
};

// open_union_of_paths (merkle/src/lib.rs:246-315) over a tree known by its level lengths.
inline int32_t merkle_open_from(FriSource& src, size_t tree, const std::vector<size_t>& lens, size_t arity, const std::vector<size_t>& indices, MerkleProofHost& pr) {
    if (indices.empty()) return -1;                                                                          // :247
    std::vector<size_t> cur = indices; std::sort(cur.begin(), cur.end()); cur.erase(std::unique(cur.begin(), cur.end()), cur.end());
    if (cur.back() >= lens[0]) return -1;
    pr.arity = arity; pr.indices = cur; pr.siblings.clear(); pr.group_sizes.clear();
    const size_t height = lens.size() - 1;
    for (size_t level = 0; level < height; ++level) {
        const size_t len = lens[level];
        std::vector<size_t> want; std::vector<uint8_t> gs;
        size_t i = 0;
        while (i < cur.size()) {                       // cur is sorted: one group per distinct parent
            size_t parent = cur[i] / arity, base = parent * arity, end = std::min(base + arity, len);
            gs.push_back((uint8_t)(end - base));
            for (size_t c = base; c < end; ++c) { if (i < cur.size() && cur[i] == c) ++i; else want.push_back(c); }
        }
        std::vector<fr_t> sib;
        if (!want.empty()) { int32_t rc = src.digests(tree, level, want, sib); if (rc) return rc; }
        pr.siblings.push_back(std::move(sib)); pr.group_sizes.push_back(std::move(gs));
        std::vector<size_t> nx; for (size_t x : cur) { size_t q = x / arity; if (nx.empty() || nx.back() != q) nx.push_back(q); }
        cur.swap(nx);
    }
    return 0;
}

// fri_prove_queries + FriQueryPayload assembly + canonical encoding.  `est` = deep_fri_proof_size_bytes (fri.rs:775-805).
inline int32_t assemble_proof(const FriShape& S, size_t r, TrHasher& H, FriSource& src, std::vector<uint8_t>& b, size_t& est) {
    const size_t L = S.schedule.size();
#define FP_TRY(e) do { int32_t rc__ = (e); if (rc__) return rc__; } while (0)
    fr_t roots_seed; FP_TRY(H.hash("FRI/seed", S.roots.data(), S.roots.size(), 1, &roots_seed));                        // fs_seed_from_roots, fri.rs:178
    // index seeds for all (q, l) in one batch of transcript hashes (fri.rs:374, :189-191)
    std::vector<fr_t> in(3 * r * L); for (size_t q = 0; q < r; ++q) for (size_t l = 0; l < L; ++l) { fr_t* p = &in[3 * (q * L + l)]; p[0] = roots_seed; p[1] = host::h_u64(l); p[2] = host::h_u64(q); }
    std::vector<fr_t> seeds(r * L);
    if (r * L) FP_TRY(H.hash("FRI/index", in.data(), 3, r * L, seeds.data()));
    auto index_from_seed = [](const fr_t& sd, size_t n_pow2) { uint8_t bb[32]; host::h_to_bytes_le(sd, bb); host::ChaCha12Rng rng(bb); return (size_t)rng.next_u64() & (n_pow2 - 1); };   // fri.rs:180-187
    struct Ref { size_t i, child_pos, parent_index, parent_pos; };
    std::vector<std::vector<Ref>> refs(r, std::vector<Ref>(L));
    std::vector<std::vector<size_t>> child_b(L), parent_b(L);
    for (size_t q = 0; q < r; ++q) for (size_t l = 0; l < L; ++l) {
        size_t n = S.n[l], n_pow2 = 1; while (n_pow2 < n) n_pow2 <<= 1; size_t m = S.schedule[l];
        const fr_t& seed = seeds[q * L + l];
        size_t i0 = index_from_seed(seed, n_pow2), i;
        if (i0 < n) i = i0;
        else { fr_t two[2] = {seed, host::h_u64(1)}, reseed; FP_TRY(H.hash("FRI/index", two, 2, 1, &reseed)); size_t i2 = index_from_seed(reseed, n_pow2); i = i2 < n ? i2 : (i2 & (n - 1)); }   // fri.rs:379-381
        refs[q][l] = Ref{i, 0, i / m, 0}; child_b[l].push_back(i); parent_b[l].push_back(i / m);
    }
    b.clear();
    est = S.roots.size() * 32 + 32 + 8;                                                                       // fri.rs:779-783
    enc_u64(b, S.roots.size()); for (auto& x : S.roots) enc_fr(b, x);
    enc_u64(b, L);
    std::vector<std::vector<size_t>> ci(L), pi(L);
    for (size_t l = 0; l < L; ++l) {
        ci[l] = child_b[l]; std::sort(ci[l].begin(), ci[l].end()); ci[l].erase(std::unique(ci[l].begin(), ci[l].end()), ci[l].end());
        pi[l] = parent_b[l]; std::sort(pi[l].begin(), pi[l].end()); pi[l].erase(std::unique(pi[l].begin(), pi[l].end()), pi[l].end());
        MerkleProofHost cp, pp;
        FP_TRY(merkle_open_from(src, l, S.lens[l], S.arity[l], ci[l], cp)); FP_TRY(merkle_open_from(src, l + 1, S.lens[l + 1], S.arity[l + 1], pi[l], pp));
        b.push_back(S.hashed[l] ? 1 : 0); enc_idxs(b, ci[l]); enc_mproof(b, cp); enc_idxs(b, pi[l]); enc_mproof(b, pp);
        for (auto& g : cp.siblings) est += g.size() * 32; for (auto& g : pp.siblings) est += g.size() * 32;
        est += ci[l].size() * 8 + pi[l].size() * 8;
        for (size_t q = 0; q < r; ++q) {
            refs[q][l].child_pos = (size_t)(std::lower_bound(ci[l].begin(), ci[l].end(), refs[q][l].i) - ci[l].begin());
            refs[q][l].parent_pos = (size_t)(std::lower_bound(pi[l].begin(), pi[l].end(), refs[q][l].parent_index) - pi[l].begin());
        }
    }
    { MerkleProofHost fp; FP_TRY(merkle_open_from(src, L, S.lens[L], S.arity[L], std::vector<size_t>{0}, fp)); enc_mproof(b, fp); for (auto& g : fp.siblings) est += g.size() * 32; }
    // opened field elements: f_i, s_i = f_{l+1}[i/m], f_parent_b = f_{l+1}[b], s_parent_b = f_{l+2}[b/m_{l+1}] (0 on the last layer)
    std::vector<std::vector<fr_t>> fi(L), fpar(L), spar(L);
    for (size_t l = 0; l < L; ++l) {
        std::vector<size_t> a(r), bb(r), cc(r);
        for (size_t q = 0; q < r; ++q) { a[q] = refs[q][l].i; bb[q] = refs[q][l].parent_index; cc[q] = l + 1 < L ? refs[q][l].parent_index / S.schedule[l + 1] : 0; }
        if (r) { FP_TRY(src.layer(l, a, fi[l])); FP_TRY(src.layer(l + 1, bb, fpar[l])); }
        if (l + 1 < L) { if (r) FP_TRY(src.layer(l + 2, cc, spar[l])); } else spar[l].assign(r, host::h_zero());
    }
    std::vector<fr_t> last_f; FP_TRY(src.layer(L, std::vector<size_t>{0}, last_f));
    enc_u64(b, r);
    for (size_t q = 0; q < r; ++q) {
        enc_u64(b, L); for (size_t l = 0; l < L; ++l) { enc_u64(b, refs[q][l].i); enc_u64(b, refs[q][l].child_pos); enc_u64(b, refs[q][l].parent_index); enc_u64(b, refs[q][l].parent_pos); }
        enc_u64(b, L); for (size_t l = 0; l < L; ++l) { enc_fr(b, fi[l][q]); enc_fr(b, fpar[l][q]) /* s_i == f_parent_b */; enc_fr(b, fpar[l][q]); enc_fr(b, spar[l][q]); }
        enc_u64(b, 0); enc_fr(b, last_f[0]); enc_fr(b, host::h_zero());                                        // final_index, final_pair (fri.rs:398-399; last s is zero, :266)
        est += 8 + 2 * 32 + L * 16 + L * 128;                                                                 // fri.rs:796-801
    }
    enc_u64(b, S.n0); enc_fr(b, fr_root_of_unity<PallasFr>((unsigned)ilog2_ceil(S.n0)));
#undef FP_TRY
    return 0;
}

// ---- plan / replay ------------------------------------------------------------------------------------------
struct FriRequest { uint32_t kind; uint32_t which; uint32_t level; uint64_t index; };   // kind 0: f_which[index]; kind 1: tree `which`, level, node index
struct RecordingSource : FriSource {
    std::vector<FriRequest> req;
    int32_t layer(size_t l, const std::vector<size_t>& idx, std::vector<fr_t>& out) override {
        for (size_t i : idx) req.push_back(FriRequest{0u, (uint32_t)l, 0u, (uint64_t)i});
        out.assign(idx.size(), host::h_zero()); return 0;
    }
    int32_t digests(size_t tree, size_t level, const std::vector<size_t>& idx, std::vector<fr_t>& out) override {
        for (size_t i : idx) req.push_back(FriRequest{1u, (uint32_t)tree, (uint32_t)level, (uint64_t)i});
        out.assign(idx.size(), host::h_zero()); return 0;
    }
};
struct ReplaySource : FriSource {
    const fr_t* vals; size_t n, pos = 0;
    ReplaySource(const fr_t* v, size_t n_) : vals(v), n(n_) {}
    int32_t take(size_t k, std::vector<fr_t>& out) { if (pos + k > n) return -1; out.assign(vals + pos, vals + pos + k); pos += k; return 0; }
    int32_t layer(size_t, const std::vector<size_t>& idx, std::vector<fr_t>& out) override { return take(idx.size(), out); }
    int32_t digests(size_t, size_t, const std::vector<size_t>& idx, std::vector<fr_t>& out) override { return take(idx.size(), out); }
};
struct FriPlan { FriShape shape; size_t r = 0; std::vector<FriRequest> req; };
inline int32_t fri_plan_make(FriPlan& P, TrHasher& H) {
    RecordingSource rec; std::vector<uint8_t> scratch; size_t est = 0;
    int32_t rc = assemble_proof(P.shape, P.r, H, rec, scratch, est); if (rc) return rc;
    P.req.swap(rec.req); return 0;
}

}  // namespace stark

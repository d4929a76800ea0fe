// oracle/merkle.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// Restatement of crates/merkle/src/lib.rs (DS-aware m-ary Poseidon Merkle tree, union-of-paths
// multiproofs and their verifiers) plus the arity-16 adapter of crates/commitment/src/lib.rs:60-114.
#pragma once
#include <algorithm>
#include <map>
#include <vector>
#include "poseidon.hpp"

namespace oracle {

// merkle/src/lib.rs:58-74  DsLabel::to_fields.
static inline void ds_fields(size_t arity, uint32_t level, uint64_t position, uint64_t tree_label, Fr out[4]) {
    out[0] = Fr::from_u64((uint64_t)arity); out[1] = Fr::from_u64((uint64_t)level);
    out[2] = Fr::from_u64(position);        out[3] = Fr::from_u64(tree_label);
}
static const uint32_t LEAF_LEVEL_DS = 0xFFFFFFFFu;  // merkle/src/lib.rs:378

// merkle/src/lib.rs:84-112  MerkleChannelCfg.
struct MerkleChannelCfg {
    size_t arity; PoseidonParams params; uint64_t tree_label;
    static MerkleChannelCfg make(size_t arity) { MerkleChannelCfg c; c.arity = arity; c.params = poseidon_params_for_arity(arity); c.tree_label = 0; return c; }
    static MerkleChannelCfg with_params(size_t arity, const PoseidonParams& p) { MerkleChannelCfg c; c.arity = arity; c.params = p; c.tree_label = 0; return c; }
    MerkleChannelCfg with_tree_label(uint64_t l) const { MerkleChannelCfg c = *this; c.tree_label = l; return c; }
};
static inline bool ok_width(size_t arity, size_t t) {  // merkle/src/lib.rs:155-161
    return (arity <= 8 && t == 9) || (arity >= 9 && arity <= 16 && t == 17) || (arity >= 17 && arity <= 32 && t == 33) ||
           (arity >= 33 && arity <= 64 && t == 65) || (arity >= 65 && arity <= 128 && t == 129);
}

// merkle/src/lib.rs:131-143  MerkleProof.
struct MerkleProof {
    std::vector<size_t> indices;
    std::vector<std::vector<Fr>> siblings;
    std::vector<std::vector<uint8_t>> group_sizes;
    size_t arity = 0;
};

// merkle/src/lib.rs:114-128  MerkleTree (levels[0] = leaves as digests).
struct MerkleTree {
    std::vector<std::vector<Fr>> levels; Fr root; MerkleChannelCfg cfg;
    size_t height() const { return levels.empty() ? 0 : levels.size() - 1; }

    // :163-179 / :415-431  level-by-level build shared by `new` and `new_pairs`.
    void build_up() {
        if (!ok_width(cfg.arity, cfg.params.t)) throw std::string("arity incompatible with Poseidon width");
        size_t arity = cfg.arity; uint32_t cur_level = 0;
        while (levels.back().size() > 1) {
            const std::vector<Fr>& cur = levels.back();
            size_t np = (cur.size() + arity - 1) / arity;
            std::vector<Fr> next(np);
            #pragma omp parallel for schedule(static)
            for (long pi = 0; pi < (long)np; ++pi) {
                size_t base = (size_t)pi * arity, cnt = std::min(arity, cur.size() - base);
                Fr ds[4]; ds_fields(arity, cur_level, (uint64_t)pi, cfg.tree_label, ds);
                next[pi] = hash_with_ds_dynamic(ds, 4, &cur[base], cnt, cfg.params);
            }
            levels.push_back(std::move(next));
            cur_level += 1;
        }
        root = levels.back()[0];
    }
    // :147-193  MerkleTree::new.
    static MerkleTree make(const std::vector<Fr>& leaves, const MerkleChannelCfg& cfg) {
        if (leaves.empty()) throw std::string("no leaves");
        MerkleTree t; t.cfg = cfg; t.levels.push_back(leaves); t.build_up(); return t;
    }
    // :380-388 encode_leaf_digest_ds + :392-445 MerkleTree::new_pairs.
    static MerkleTree make_pairs(const std::vector<Fr>& f, const std::vector<Fr>& cp, const MerkleChannelCfg& cfg) {
        if (f.size() != cp.size()) throw std::string("f and cp length mismatch");
        if (f.empty()) throw std::string("no leaves");
        MerkleTree t; t.cfg = cfg; std::vector<Fr> l0(f.size());
        #pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)f.size(); ++i) {
            Fr ds[4]; ds_fields(cfg.arity, LEAF_LEVEL_DS, (uint64_t)i, cfg.tree_label, ds);
            Fr in[2] = {f[i], cp[i]};
            l0[i] = hash_with_ds_dynamic(ds, 4, in, 2, cfg.params);
        }
        t.levels.push_back(std::move(l0)); t.build_up(); return t;
    }
    // :246-315  open_union_of_paths (== open_many == open_many_single).
    MerkleProof open(const std::vector<size_t>& indices) const {
        if (indices.empty()) throw std::string("open_many: empty indices");
        size_t arity = cfg.arity;
        std::vector<size_t> cur = indices; std::sort(cur.begin(), cur.end()); cur.erase(std::unique(cur.begin(), cur.end()), cur.end());
        MerkleProof pr; pr.arity = arity; pr.indices = cur;
        for (size_t level = 0; level < height(); ++level) {
            const std::vector<Fr>& nodes = levels[level]; size_t len = nodes.size();
            std::map<size_t, std::vector<size_t>> groups;
            for (size_t i : cur) groups[i / arity].push_back(i % arity);
            std::vector<Fr> sib; std::vector<uint8_t> gs;
            for (auto& kv : groups) {
                std::vector<size_t> opened = kv.second; std::sort(opened.begin(), opened.end());
                size_t base = kv.first * arity, end = std::min(base + arity, len), cc = end - base;
                gs.push_back((uint8_t)cc);
                size_t oi = 0;
                for (size_t cpos = 0; cpos < cc; ++cpos) {
                    if (oi < opened.size() && opened[oi] == cpos) ++oi; else sib.push_back(nodes[base + cpos]);
                }
            }
            pr.siblings.push_back(sib); pr.group_sizes.push_back(gs);
            std::vector<size_t> nx; for (size_t i : cur) nx.push_back(i / arity);
            std::sort(nx.begin(), nx.end()); nx.erase(std::unique(nx.begin(), nx.end()), nx.end());
            cur = nx;
        }
        return pr;
    }
};

// merkle/src/lib.rs:587-701  verify_many_ds.
static inline bool verify_many_ds(const Fr& root, const std::vector<size_t>& indices, const std::vector<Fr>& values,
                                  const MerkleProof& proof, uint64_t tree_label, const PoseidonParams& dp) {
    if (indices.empty() || indices.size() != values.size()) return false;
    std::vector<size_t> req = indices; std::sort(req.begin(), req.end()); req.erase(std::unique(req.begin(), req.end()), req.end());
    if (proof.indices != req) return false;
    if (proof.siblings.size() != proof.group_sizes.size()) return false;
    size_t arity = proof.arity;
    if (!ok_width(arity, dp.t)) return false;
    if (arity == 0) throw std::string("verify_many_ds: arity 0 passes the width guard for t = 9 and the reference then divides by it (panic)");
    std::map<size_t, Fr> m; for (size_t k = 0; k < indices.size(); ++k) m[indices[k]] = values[k];
    std::vector<size_t> cur_i = req; std::vector<Fr> cur_v; for (size_t i : cur_i) cur_v.push_back(m[i]);
    for (size_t level = 0; level < proof.siblings.size(); ++level) {
        const std::vector<Fr>& sib = proof.siblings[level]; const std::vector<uint8_t>& gs = proof.group_sizes[level];
        std::map<size_t, std::vector<std::pair<size_t, Fr>>> groups;
        for (size_t k = 0; k < cur_i.size(); ++k) groups[cur_i[k] / arity].push_back({cur_i[k] % arity, cur_v[k]});
        if (groups.size() != gs.size()) return false;
        std::vector<size_t> nx_i; std::vector<Fr> nx_v; size_t off = 0, gi = 0;
        for (auto& kv : groups) {
            size_t cc = gs[gi++];
            if (cc == 0 || cc > arity) return false;
            auto opened = kv.second; std::sort(opened.begin(), opened.end(), [](auto& a, auto& b) { return a.first < b.first; });
            std::vector<Fr> children; size_t oi = 0;
            for (size_t cpos = 0; cpos < cc; ++cpos) {
                if (oi < opened.size() && opened[oi].first == cpos) { children.push_back(opened[oi].second); ++oi; continue; }
                if (off >= sib.size()) return false;
                children.push_back(sib[off++]);
            }
            Fr ds[4]; ds_fields(arity, (uint32_t)level, (uint64_t)kv.first, tree_label, ds);
            nx_i.push_back(kv.first); nx_v.push_back(hash_with_ds_dynamic(ds, 4, children.data(), children.size(), dp));
        }
        if (off != sib.size()) return false;
        cur_i = nx_i; cur_v = nx_v;
    }
    if (cur_v.size() != 1) return false;
    return cur_v[0] == root;
}
// merkle/src/lib.rs:723-773  verify_pairs_ds.
static inline bool verify_pairs_ds(const Fr& root, const std::vector<size_t>& indices, const std::vector<std::pair<Fr, Fr>>& pairs,
                                   const MerkleProof& proof, uint64_t tree_label, const PoseidonParams& dp) {
    if (indices.size() != pairs.size() || indices.empty()) return false;
    size_t arity = proof.arity;
    if (!ok_width(arity, dp.t)) return false;
    std::vector<size_t> req = indices; std::sort(req.begin(), req.end()); req.erase(std::unique(req.begin(), req.end()), req.end());
    std::map<size_t, std::pair<Fr, Fr>> mp; for (size_t k = 0; k < indices.size(); ++k) mp[indices[k]] = pairs[k];
    std::vector<Fr> leaves;
    for (size_t idx : req) {
        Fr ds[4]; ds_fields(arity, LEAF_LEVEL_DS, (uint64_t)idx, tree_label, ds);
        Fr in[2] = {mp[idx].first, mp[idx].second};
        leaves.push_back(hash_with_ds_dynamic(ds, 4, in, 2, dp));
    }
    return verify_many_ds(root, req, leaves, proof, tree_label, dp);
}

// commitment/src/lib.rs:48-51,60-90  MerkleCommitment::commit: arity 16, params seed
// "POSEIDON-T17-X5-SEED", tree_label = cfg.ds_tag.
static inline MerkleTree commitment_commit(const std::vector<Fr>& leaves, uint64_t ds_tag) {
    PoseidonParams p = generate_params_t17_x5(bytes_of("POSEIDON-T17-X5-SEED"));
    MerkleChannelCfg cfg = MerkleChannelCfg::with_params(16, p).with_tree_label(ds_tag);
    return MerkleTree::make(leaves, cfg);
}

}  // namespace oracle

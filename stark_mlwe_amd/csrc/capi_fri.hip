// stark_mlwe_amd/csrc/capi_fri.hip — FRI folding, fri_build_transcript, DEEP-ALI merge, build_f0 and
// the end-to-end deep_fri_prove orchestration (host logic in C++ above the kernels, mirroring
// crates/deep_ali/src/fri.rs and crates/deep_ali/src/lib.rs).  C-ABI in include/stark_mlwe.h.
#include <algorithm>
#include <chrono>
#include <thread>
#include <cstring>
#include <map>
#include "ctx.hpp"
#include "fri_dev.hpp"

using namespace stark;

struct stark_fri_state {
    CtxRef ref_;
    stark_ctx* ctx = nullptr;
    std::vector<size_t> schedule;
    std::vector<fr_t*> f; std::vector<size_t> n;           // L+1 layers (device, pooled)
    std::vector<fr_t> z;                                    // L fold challenges
    std::vector<stark_tree*> trees; std::vector<char> hashed; std::vector<size_t> arity;
    std::vector<fr_t> roots;                                // fetched on first use (one download + one sync for all L+1)
    ~stark_fri_state() { for (auto p : f) if (p) ctx_release(ctx, p); for (auto t : trees) if (t) stark_merkle_free(t); }
};
struct stark_proof { std::vector<uint8_t> bytes; size_t size_estimate = 0; double ms[3] = {0, 0, 0}; };

static inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
static inline int ilog2(size_t x) { return ilog2_ceil(x); }

// fold on `st` with the z-power table zp (m entries, device)
static int32_t fold_launch(stark_ctx* ctx, hipStream_t st, const fr_t* f, size_t n, const fr_t* zp, size_t m, fr_t* out) {
    if (is_pow2(m)) {
        int log_m = ilog2(m), log_g = std::min(log_m, 4);
        uint64_t lanes = (uint64_t)n >> (log_m - log_g);
        hipLaunchKernelGGL(k_fri_fold_pow2<PallasFr>, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, f, (uint64_t)n, zp, log_m, log_g, out);
    } else {
        uint64_t no = n / m;
        hipLaunchKernelGGL(k_fri_fold_any<PallasFr>, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, st, f, no, zp, (uint64_t)m, out);
    }
    STARK_HIP(ctx, hipGetLastError());
    return STARK_OK;
}
// z^0..z^(m-1) (fri.rs:91-93) computed ON the device into a pooled table: no host round trip, nothing to keep alive on the host.
static int32_t zpows_launch(stark_ctx* ctx, hipStream_t st, const fr_t& z, size_t m, fr_t* zp) {
    hipLaunchKernelGGL(k_zpows<PallasFr>, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, st, z, (uint64_t)m, zp);
    STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
}
static int32_t fold_dev(stark_ctx* ctx, const fr_t* f, size_t n, const fr_t& z, size_t m, fr_t* out) {
    if (m < 2) return ctx->fail(STARK_ERR_INVALID_ARG, "m >= 2");                                        // fri.rs:86
    if (n % m) return ctx->fail(STARK_ERR_INVALID_ARG, "layer size must be divisible by m");             // fri.rs:87
    if (!n) return STARK_OK;
    DevBuf zp; STARK_HIP(ctx, zp.alloc(ctx, m * sizeof(fr_t)));
    STARK_TRY(zpows_launch(ctx, ctx->stream, z, m, zp.fr()));
    return fold_launch(ctx, ctx->stream, f, n, zp.fr(), m, out);     // zp returns to the pool: its next user is ordered behind this fold on the same stream
}

// fri_sample_z_ell (fri.rs:59-82): transcript hash on the device, ChaCha12 + candidate test on the host.  The value depends
// only on (seed_z, level, domain_size) (fri.rs:250), so it is computed once per context and key.
static int32_t sample_z(stark_ctx* ctx, uint64_t seed_z, size_t level, size_t domain_size, fr_t* z) {
    const stark_ctx::ZKey key{seed_z, level, domain_size};
    auto it = ctx->z_cache.find(key);
    if (it != ctx->z_cache.end()) { *z = it->second; return STARK_OK; }
    fr_t fused; STARK_TRY(tr_hash_host1(ctx, "FRI/z/l", {host::h_u64(seed_z), host::h_u64(level), host::h_u64(domain_size)}, &fused));
    uint8_t seed[32]; host::h_to_bytes_le(fused, seed);
    host::ChaCha12Rng rng(seed);
    const fr_t one = host::h_one();
    for (size_t tries = 0;;) {
        fr_t cand = host::h_u64(rng.next_u64());
        if (!fr_is_zero(cand) && !fr_eq(fr_pow_u64<PallasFr>(cand, domain_size), one)) { *z = cand; break; }
        if (++tries >= 1000) {
            fr_t fb = host::h_u64(seed_z + (uint64_t)level + 7);
            *z = !fr_eq(fr_pow_u64<PallasFr>(fb, domain_size), one) ? fb : host::h_u64(11); break;
        }
    }
    ctx->z_cache[key] = *z; return STARK_OK;
}

// All L+1 roots with one download and one synchronisation (the query phase needs them on the host; a caller that only builds
// the commitments never pays for it).
static int32_t state_roots(stark_fri_state* S) {
    if (!S->roots.empty()) return STARK_OK;
    stark_ctx* ctx = S->ctx; std::vector<fr_t> r(S->trees.size());
    for (size_t l = 0; l < S->trees.size(); ++l) {
        stark_tree* T = S->trees[l];
        if (T->lens.back() != 1) return ctx->fail(STARK_ERR_INVALID_ARG, "partial (sharded) tree has no root");
        STARK_HIP(ctx, hipMemcpyAsync(&r[l], T->levels.back(), sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    S->roots = r; return STARK_OK;
}

static int32_t fri_build_impl(stark_ctx* ctx, const fr_t* f0_dev, size_t n0, const size_t* schedule, size_t L, uint64_t seed_z, stark_fri_state** out) {
    if (!n0) return ctx->fail(STARK_ERR_INVALID_ARG, "empty layer");
    { size_t n = n0; for (size_t l = 0; l < L; ++l) { if (schedule[l] < 2 || n % schedule[l]) return ctx->fail(STARK_ERR_INVALID_ARG, "schedule not dividing domain size"); n /= schedule[l]; } }   // fri.rs:150
    stark_fri_state* S = new stark_fri_state(); S->ref_.bind(ctx); S->ctx = ctx; S->schedule.assign(schedule, schedule + L);
    auto bail = [&](int32_t rc) { delete S; return rc; };
    // Everything that may upload constants (and synchronise doing so) happens BEFORE any stream is forked: transcript
    // parameters, the leaf template, the Merkle parameters of every layer, the challenges.  After the first call these are all cached.
    S->n.push_back(n0); for (size_t l = 0; l < L; ++l) S->n.push_back(S->n[l] / schedule[l]);
    S->arity.assign(L + 1, 0); S->hashed.assign(L + 1, 0); S->trees.assign(L + 1, nullptr);
    std::vector<stark_params*> mps(L + 1, nullptr);
    { stark_params* tp = nullptr; int32_t rc = ctx_transcript_params(ctx, &tp); if (rc) return bail(rc); }
    for (size_t l = 0; l <= L; ++l) {
        const size_t m_l = l < L ? schedule[l] : 1;
        S->arity[l] = pick_arity_for_layer(S->n[l], m_l); S->hashed[l] = hashed_arity(S->arity[l]) ? 1 : 0;
        int32_t rc = ctx_merkle_params(ctx, host::width_for_arity(S->arity[l]), &mps[l]); if (rc) return bail(rc);     // MerkleChannelCfg::new(arity), fri.rs:277
    }
    size_t zp_total = 0;
    for (size_t l = 0; l < L; ++l) { fr_t z; int32_t rc = sample_z(ctx, seed_z, l, S->n[l], &z); if (rc) return bail(rc); S->z.push_back(z); zp_total += schedule[l]; }
    hipStream_t main_stream = ctx->stream, side = nullptr;
    { int32_t rc = ctx_side_stream(ctx, &side); if (rc) return bail(rc); }
    // layer 0 copy + folds back to back (the challenges do not depend on any commitment: fri.rs:250)
    for (size_t l = 0; l <= L; ++l) { void* q = nullptr; int32_t rc = ctx_alloc(ctx, S->n[l] * sizeof(fr_t), &q); S->f.push_back((fr_t*)q); if (rc) return bail(rc); }
    if (hipMemcpyAsync(S->f[0], f0_dev, n0 * sizeof(fr_t), hipMemcpyDeviceToDevice, main_stream) != hipSuccess) return bail(ctx->fail(STARK_ERR_HIP, "copy f0"));
    DevBuf zp; if (L && zp.alloc(ctx, zp_total * sizeof(fr_t)) != hipSuccess) return bail(ctx->fail(STARK_ERR_OOM, "z powers"));
    { size_t off = 0;
      for (size_t l = 0; l < L; ++l) {
          int32_t rc = zpows_launch(ctx, main_stream, S->z[l], schedule[l], zp.fr() + off); if (rc) return bail(rc);
          rc = fold_launch(ctx, main_stream, S->f[l], S->n[l], zp.fr() + off, schedule[l], S->f[l + 1]); if (rc) return bail(rc);
          off += schedule[l];
      } }
    // Commitments of all L+1 layers (independent jobs).  Layer 0 is ~94 % of the hashing and fills the GPU; the later layers
    // are small and mostly LATENCY-bound (tree tops: one dependent permutation per level), so they go to a side stream and run
    // underneath layer 0 instead of after it.  The streams are passed explicitly; the side stream is forked from and joined back
    // into the main one with events, so nothing here synchronises the host.
    if (hipEventRecord(ctx->ev_fork, main_stream) != hipSuccess || hipStreamWaitEvent(side, ctx->ev_fork, 0) != hipSuccess) return bail(ctx->fail(STARK_ERR_HIP, "fork"));
    auto commit_layer = [&](size_t l, hipStream_t st) -> int32_t {
        const size_t n = S->n[l], m_l = l < L ? schedule[l] : 1, arity = S->arity[l];
        stark_tree* T = nullptr;
        if (S->hashed[l]) {
            void* h = nullptr; STARK_TRY(ctx_alloc(ctx, n * sizeof(fr_t), &h));
            int32_t rc = leaf_pair_hash_on(ctx, st, S->f[l], l < L ? S->f[l + 1] : nullptr, n, m_l, (fr_t*)h);        // fri.rs:283 (s = f_{l+1}[i/m] view)
            if (rc == STARK_OK) rc = merkle_build_on(ctx, st, mps[l], arity, (uint64_t)l, (const fr_t*)h, n, 0, nullptr, 1, 0, 0, 0, /*adopt=*/true, &T);   // the digests become level 0
            if (rc) { if (!T) ctx_release(ctx, h); return rc; }
        } else {
            // commit_pairs(f_l, s_l) (fri.rs:289): s_l is the m-fold replication of f_{l+1} (read as the view f_{l+1}[i / m]), or zeros on the last layer (fri.rs:266)
            STARK_TRY(merkle_build_on(ctx, st, mps[l], arity, (uint64_t)l, S->f[l], n, 1, l < L ? S->f[l + 1] : nullptr, m_l, 0, 0, 0, false, &T));
        }
        S->trees[l] = T;
        return STARK_OK;
    };
    int32_t crc = STARK_OK;
    for (size_t l = L; l >= 1 && crc == STARK_OK; --l) crc = commit_layer(l, side);
    if (crc == STARK_OK) crc = commit_layer(0, main_stream);
    // join: the main stream continues only after the side stream's commitments
    if (hipEventRecord(ctx->ev_fork, side) != hipSuccess || hipStreamWaitEvent(main_stream, ctx->ev_fork, 0) != hipSuccess) { (void)hipStreamSynchronize(side); return bail(ctx->fail(STARK_ERR_HIP, "join")); }
    if (crc != STARK_OK) { (void)hipStreamSynchronize(side); (void)hipStreamSynchronize(main_stream); return bail(crc); }
    *out = S; return STARK_OK;
}

// Two-level power table of a domain generator, cached per (generator, size) — the reference's DomainH (deep_ali/src/lib.rs:109-125).
static int32_t omega_table(stark_ctx* ctx, const fr_t& omega, size_t n_global, PowTable* out) {
    int bits = ilog2(n_global); if (bits < 1) bits = 1;
    for (auto& o : ctx->omega_tabs) if (o.bits == bits && fr_eq(o.omega, omega)) { *out = PowTable{o.lo, o.hi, o.lo_bits}; return STARK_OK; }
    const int lo_bits = (bits + 1) / 2, hi_bits = bits - lo_bits + 1;
    fr_t *lo = nullptr, *hi = nullptr;
    STARK_HIP(ctx, hipMalloc((void**)&lo, ((size_t)1 << lo_bits) * sizeof(fr_t)));
    if (hipMalloc((void**)&hi, ((size_t)1 << hi_bits) * sizeof(fr_t)) != hipSuccess) { (void)hipFree(lo); return ctx->fail(STARK_ERR_OOM, "omega table"); }
    const uint64_t tot = (1ull << lo_bits) + (1ull << hi_bits);
    hipLaunchKernelGGL(k_fill_pow_table<PallasFr>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, lo, hi, lo_bits, hi_bits, omega, host::h_one());
    STARK_HIP(ctx, hipGetLastError());
    if (ctx->omega_tabs.size() >= 16) { STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->omega_tabs[0].lo); (void)hipFree(ctx->omega_tabs[0].hi); ctx->omega_tabs.erase(ctx->omega_tabs.begin()); }
    ctx->omega_tabs.push_back({bits, omega, lo, hi, lo_bits});
    *out = PowTable{lo, hi, lo_bits}; return STARK_OK;
}

// deep_ali_merge_evals_blinded on device pointers (deep_ali/src/lib.rs:60-105).
static int32_t ali_merge_dev_impl(stark_ctx* ctx, const fr_t* a, const fr_t* s, const fr_t* e, const fr_t* t, const fr_t* r_opt, const fr_t& beta,
                                  const fr_t& omega, const fr_t& z, size_t n, fr_t* f0, fr_t* c_star_host,
                                  uint64_t j0 = 0, size_t n_global = 0, bool partial_only = false) {
    // n = number of local elements, holding the global positions j0 .. j0+n-1 of a domain of n_global points (whole vector: j0 = 0, n_global = n)
    if (!n_global) n_global = n;
    if (n_global <= 1 || !n || j0 + n > n_global) return ctx->fail(STARK_ERR_INVALID_ARG, "n > 1");                        // lib.rs:71
    if (fr_eq(fr_pow_u64<PallasFr>(z, n_global), host::h_one())) return ctx->fail(STARK_ERR_INVALID_ARG, "z must be outside H");   // lib.rs:78
    PowTable wp; STARK_TRY(omega_table(ctx, omega, n_global, &wp));
    const unsigned block = 256; uint64_t lanes = (n + ALI_K - 1) / ALI_K; unsigned grid = (unsigned)((lanes + block - 1) / block);
    const uint64_t T = (uint64_t)grid * block;
    fr_t w_step = fr_pow_u64<PallasFr>(omega, T);
    DevBuf sums; if (c_star_host) STARK_HIP(ctx, sums.alloc(ctx, (size_t)grid * sizeof(fr_t)));
    hipLaunchKernelGGL(k_ali_merge<PallasFr>, dim3(grid), dim3(block), 0, ctx->stream, a, s, e, t, r_opt, beta, wp, w_step, fr_inv<PallasFr>(w_step), z, (uint64_t)n, j0, f0, c_star_host ? sums.fr() : (fr_t*)nullptr);
    STARK_HIP(ctx, hipGetLastError());
    if (c_star_host) {
        // c* = phi(z)/Z_H(z) = (1/n) * sum_j phi_j w^j/(z - w^j)   (lib.rs:44 and :94); block partials are reduced on the device
        DevBuf tot; STARK_HIP(ctx, tot.alloc(ctx, sizeof(fr_t)));
        hipLaunchKernelGGL(k_sum_single_block<PallasFr>, dim3(1), dim3(256), 0, ctx->stream, (const fr_t*)sums.fr(), (uint64_t)grid, partial_only ? host::h_one() : fr_inv<PallasFr>(host::h_u64(n_global)), tot.fr());
        STARK_HIP(ctx, hipGetLastError());
        STARK_HIP(ctx, hipMemcpyAsync(c_star_host, tot.p, sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
        STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));      // the caller reads *c_star_host on return
    }
    return STARK_OK;
}

// ali_sample_z_beta_fs (fri.rs:511-533).
static void ali_z_beta_from_fused(const fr_t& fused, size_t n0, const fr_t& roots_seed, fr_t* z, fr_t* beta);
static int32_t ali_sample_z_beta(stark_ctx* ctx, const char* tag, size_t n0, const fr_t& roots_seed, fr_t* z, fr_t* beta) {
    fr_t fused; STARK_TRY(tr_hash_host1(ctx, tag, {roots_seed, host::h_u64(n0)}, &fused));
    ali_z_beta_from_fused(fused, n0, roots_seed, z, beta); return STARK_OK;
}
// the RNG part of ali_sample_z_beta_fs (fri.rs:516-532): beta, then the first candidate outside H
static void ali_z_beta_from_fused(const fr_t& fused, size_t n0, const fr_t& roots_seed, fr_t* z, fr_t* beta) {
    uint8_t seed[32]; host::h_to_bytes_le(fused, seed); host::ChaCha12Rng rng(seed);
    *beta = host::h_u64(rng.next_u64());
    const fr_t one = host::h_one();
    for (size_t tries = 0;;) {
        fr_t cand = host::h_u64(rng.next_u64());
        if (!fr_is_zero(cand) && !fr_eq(fr_pow_u64<PallasFr>(cand, n0), one)) { *z = cand; return; }
        if (++tries >= 1000) {
            fr_t fb = host::h_add(roots_seed, host::h_u64(17));
            *z = !fr_eq(fr_pow_u64<PallasFr>(fb, n0), one) ? fb : host::h_u64(19); return;
        }
    }
}
// DeepAliRealBuilder::build_f0 (fri.rs:535-569), default builder: no blinding, ds_tag "ALI/DEEP".
static int32_t build_f0_dev_impl(stark_ctx* ctx, const fr_t* a, const fr_t* s, const fr_t* e, const fr_t* t, size_t n0, fr_t* f0, fr_t* aux7) {
    if (n0 <= 1) return ctx->fail(STARK_ERR_INVALID_ARG, "n0 > 1");
    // four serial column sponges (fri.rs:551-554): one lane per column, inherently sequential in n0
    DevBuf dig; STARK_HIP(ctx, dig.alloc(ctx, 4 * sizeof(fr_t)));
    const fr_t* cols[4] = {a, s, e, t}; const char* tags[4] = {"ALI/A", "ALI/S", "ALI/E", "ALI/T"};
    STARK_TRY(tr_hash_columns4_dev(ctx, tags, cols, n0, dig.fr()));           // the chains are independent: one launch, four concurrent blocks
    fr_t h[5]; STARK_HIP(ctx, hipMemcpyAsync(h, dig.p, 4 * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h[4] = host::h_u64(n0);
    fr_t seed_f; STARK_TRY(tr_hash_host1(ctx, "ALI/seed", std::vector<fr_t>(h, h + 5), &seed_f));
    fr_t z, beta; STARK_TRY(ali_sample_z_beta(ctx, "ALI/DEEP", n0, seed_f, &z, &beta));
    if (aux7) { for (int c = 0; c < 4; ++c) aux7[c] = h[c]; aux7[4] = seed_f; aux7[5] = z; aux7[6] = beta; }
    fr_t omega = fr_root_of_unity<PallasFr>((unsigned)ilog2(n0));         // FriDomain::new_radix2(n0).omega (fri.rs:53-56); Radix2EvaluationDomain::new rounds n0 up to a power of two, as ilog2 does
    return ali_merge_dev_impl(ctx, a, s, e, t, nullptr, host::h_zero(), omega, z, n0, f0, nullptr);
}

// fri_prove_queries + payload assembly + canonical encoding (fri.rs:355-466, 613-640).
// Sources of the query phase (fri_plan.hpp): the device-resident state of one GPU, and the device transcript hasher.
struct LocalSource : FriSource {
    stark_ctx* ctx; stark_fri_state* S;
    LocalSource(stark_ctx* c, stark_fri_state* s) : ctx(c), S(s) {}
    int32_t layer(size_t l, const std::vector<size_t>& idx, std::vector<fr_t>& outv) override {
        outv.resize(idx.size()); if (idx.empty()) return STARK_OK;
        if (l >= S->f.size()) return ctx->fail(STARK_ERR_INVALID_ARG, "layer out of range");
        for (size_t i : idx) if (i >= S->n[l]) return ctx->fail(STARK_ERR_INVALID_ARG, "layer index out of range");
        DevBuf di, dout; STARK_HIP(ctx, di.alloc(ctx, idx.size() * 8)); STARK_HIP(ctx, dout.alloc(ctx, idx.size() * sizeof(fr_t)));
        std::vector<uint64_t> ix(idx.begin(), idx.end());
        STARK_HIP(ctx, hipMemcpyAsync(di.p, ix.data(), ix.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_gather, dim3((unsigned)((ix.size() + 255) / 256)), dim3(256), 0, ctx->stream, (const fr_t*)S->f[l], (const uint64_t*)di.p, (uint64_t)ix.size(), dout.fr());
        STARK_HIP(ctx, hipGetLastError());
        STARK_HIP(ctx, hipMemcpyAsync(outv.data(), dout.p, ix.size() * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return STARK_OK;
    }
    int32_t digests(size_t tree, size_t level, const std::vector<size_t>& idx, std::vector<fr_t>& out) override {
        if (tree >= S->trees.size()) return ctx->fail(STARK_ERR_INVALID_ARG, "tree out of range");
        out.resize(idx.size()); return stark_merkle_gather(S->trees[tree], (int32_t)level, idx.data(), idx.size(), (uint64_t*)out.data());
    }
};
struct DeviceHasher : TrHasher {
    stark_ctx* ctx; explicit DeviceHasher(stark_ctx* c) : ctx(c) {}
    int32_t hash(const char* tag, const fr_t* fields, size_t k, size_t n, fr_t* out) override {
        if (n == 1) return tr_hash_host1(ctx, tag, std::vector<fr_t>(fields, fields + k), out);
        return stark_tr_hash_fields_tagged(ctx, nullptr, tag, (const uint64_t*)fields, k, n, (uint64_t*)out);
    }
};
static int32_t shape_of_state(stark_ctx* ctx, stark_fri_state* S, size_t n0, FriShape& sh) {
    STARK_TRY(state_roots(S));
    std::string err;
    if (!sh.make(n0, S->schedule.data(), S->schedule.size(), S->roots.data(), err)) return ctx->fail(STARK_ERR_INVALID_ARG, err);
    return STARK_OK;
}
// Transcript hashes of the query phase are pure functions of their inputs: the plan pass and the assembling pass ask for the same ones.
struct MemoHasher : TrHasher {
    TrHasher& inner; std::map<std::string, std::vector<fr_t>> memo;
    explicit MemoHasher(TrHasher& h) : inner(h) {}
    int32_t hash(const char* tag, const fr_t* fields, size_t k, size_t n, fr_t* out) override {
        std::string key(tag); key.push_back('\0'); key.append((const char*)&k, sizeof(k)); key.append((const char*)fields, k * n * sizeof(fr_t));
        auto it = memo.find(key);
        if (it == memo.end()) { std::vector<fr_t> v(n); int32_t rc = inner.hash(tag, fields, k, n, v.data()); if (rc) return rc; it = memo.emplace(std::move(key), std::move(v)).first; }
        memcpy((void*)out, it->second.data(), n * sizeof(fr_t)); return 0;
    }
};
// fri_prove_queries + payload assembly + canonical encoding (fri.rs:355-466, 613-640).  The indices of every opened value depend
// only on the roots, so the query phase first RECORDS what it will read (fri_plan.hpp: the same code against a recording source),
// fetches all of it — a few thousand layer elements and tree nodes spread over every layer and level — with ONE gather launch and
// one download, and then assembles the proof from that list.  (One synchronisation instead of one per opened level.)
static int32_t prove_queries_encode(stark_ctx* ctx, stark_fri_state* S, size_t n0, size_t r, stark_proof* P) {
    FriPlan plan; plan.r = r; STARK_TRY(shape_of_state(ctx, S, n0, plan.shape));
    DeviceHasher H0(ctx); MemoHasher H(H0);
    { int32_t rc = fri_plan_make(plan, H); if (rc == -1) return ctx->fail(STARK_ERR_INVALID_ARG, "query phase: bad index"); if (rc) return rc; }
    const size_t nreq = plan.req.size();
    std::vector<fr_t> vals(nreq);
    if (nreq) {
        // source table: layers first, then the levels of every tree
        std::vector<const fr_t*> base; std::vector<size_t> lens; std::map<std::pair<uint32_t, uint32_t>, uint32_t> tree_slot;
        for (size_t l = 0; l < S->f.size(); ++l) { base.push_back(S->f[l]); lens.push_back(S->n[l]); }
        std::vector<uint32_t> src(nreq); std::vector<uint64_t> idx(nreq);
        for (size_t i = 0; i < nreq; ++i) {
            const FriRequest& q = plan.req[i];
            if (q.kind == 0) { if (q.which >= S->f.size()) return ctx->fail(STARK_ERR_INVALID_ARG, "layer out of range"); src[i] = q.which; }
            else {
                if (q.which >= S->trees.size() || q.level >= S->trees[q.which]->levels.size()) return ctx->fail(STARK_ERR_INVALID_ARG, "tree level out of range");
                auto key = std::make_pair(q.which, q.level); auto it = tree_slot.find(key);
                if (it == tree_slot.end()) { it = tree_slot.emplace(key, (uint32_t)base.size()).first; base.push_back(S->trees[q.which]->levels[q.level]); lens.push_back(S->trees[q.which]->lens[q.level]); }
                src[i] = it->second;
            }
            if (q.index >= lens[src[i]]) return ctx->fail(STARK_ERR_INVALID_ARG, "opening index out of range");
            idx[i] = q.index;
        }
        DevBuf db, ds, di, dout;
        STARK_HIP(ctx, db.alloc(ctx, base.size() * sizeof(void*))); STARK_HIP(ctx, ds.alloc(ctx, nreq * 4)); STARK_HIP(ctx, di.alloc(ctx, nreq * 8)); STARK_HIP(ctx, dout.alloc(ctx, nreq * sizeof(fr_t)));
        STARK_HIP(ctx, hipMemcpyAsync(db.p, base.data(), base.size() * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
        STARK_HIP(ctx, hipMemcpyAsync(ds.p, src.data(), nreq * 4, hipMemcpyHostToDevice, ctx->stream));
        STARK_HIP(ctx, hipMemcpyAsync(di.p, idx.data(), nreq * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_gather_multi, dim3((unsigned)((nreq + 255) / 256)), dim3(256), 0, ctx->stream, (const fr_t* const*)db.p, (const uint32_t*)ds.p, (const uint64_t*)di.p, (uint64_t)nreq, dout.fr());
        STARK_HIP(ctx, hipGetLastError());
        STARK_HIP(ctx, hipMemcpyAsync(vals.data(), dout.p, nreq * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
        STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    ReplaySource rep(vals.data(), vals.size());
    int32_t rc = assemble_proof(plan.shape, r, H, rep, P->bytes, P->size_estimate);
    if (rc == -1 || (rc == 0 && rep.pos != vals.size())) return ctx->fail(STARK_ERR_INVALID_ARG, "query phase: value list does not match the plan");
    return rc;
}

// Query plan of a commit phase whose layers live elsewhere (sharded over ranks): see fri_plan.hpp.
struct stark_fri_plan { CtxRef ref_; stark_ctx* ctx = nullptr; FriPlan plan; };

static int32_t prove_impl(stark_ctx* ctx, const fr_t* a, const fr_t* s, const fr_t* e, const fr_t* t, const fr_t* f0_in, size_t n0,
                          const size_t* schedule, size_t L, size_t r, uint64_t seed_z, stark_proof** out) {
    if (!is_pow2(n0)) return ctx->fail(STARK_ERR_INVALID_ARG, "n0 must be a power of two (radix-2 domain)");
    auto now = [] { return std::chrono::steady_clock::now(); };
    stark_proof* P = new stark_proof(); auto t0 = now();
    DevBuf f0buf; const fr_t* f0 = f0_in;
    if (!f0) {
        if (f0buf.alloc(ctx, n0 * sizeof(fr_t)) != hipSuccess) { delete P; return ctx->fail(STARK_ERR_OOM, "f0"); }
        int32_t rc = build_f0_dev_impl(ctx, a, s, e, t, n0, f0buf.fr(), nullptr); if (rc) { delete P; return rc; }
        f0 = f0buf.fr();
    }
    auto t1 = now();
    stark_fri_state* S = nullptr; { int32_t rc = fri_build_impl(ctx, f0, n0, schedule, L, seed_z, &S); if (rc) { delete P; return rc; } }
    auto t2 = now();
    { int32_t rc = prove_queries_encode(ctx, S, n0, r, P); delete S; if (rc) { delete P; return rc; } }
    auto t3 = now();
    P->ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count(); P->ms[1] = std::chrono::duration<double, std::milli>(t2 - t1).count(); P->ms[2] = std::chrono::duration<double, std::milli>(t3 - t2).count();
    *out = P; return STARK_OK;
}

// B independent proofs of equal shape (stark_deep_fri_prove_batch_dev).  What bounds one prove is the serial column sponge of build_f0
// (fri.rs:548-557: n0/16 dependent permutations per column, one wave each): four waves of the chip are busy for 99 % of the time.  The chains of
// different traces are independent, so all 4 * B of them run in ONE launch; the two Fiat-Shamir hashes per trace (ALI/seed, ALI/DEEP) are
// one launch each for the whole batch; merge, fri_build and the query phase then run trace after trace on the context's stream.
// Every proof is byte-for-byte what stark_deep_fri_prove_dev returns for that trace alone.
static int32_t prove_batch_impl(stark_ctx* ctx, size_t B, const uint64_t* const* a, const uint64_t* const* s, const uint64_t* const* e, const uint64_t* const* t, size_t n0,
                                const size_t* schedule, size_t L, size_t r, uint64_t seed_z, stark_proof** out) {
    if (!is_pow2(n0)) return ctx->fail(STARK_ERR_INVALID_ARG, "n0 must be a power of two (radix-2 domain)");
    if (n0 <= 1) return ctx->fail(STARK_ERR_INVALID_ARG, "n0 > 1");
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_of = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
    for (size_t p = 0; p < B; ++p) out[p] = nullptr;
    auto t0 = now();
    // (1) all column digests: 4 * B serial sponges, concurrently
    std::vector<const fr_t*> ptrs(4 * B);
    for (size_t p = 0; p < B; ++p) { ptrs[4 * p] = as_fr(a[p]); ptrs[4 * p + 1] = as_fr(s[p]); ptrs[4 * p + 2] = as_fr(e[p]); ptrs[4 * p + 3] = as_fr(t[p]); }
    DevBuf dptr, dig, seeds_in, seeds, deep_in, fused;
    STARK_HIP(ctx, dptr.alloc(ctx, ptrs.size() * sizeof(void*))); STARK_HIP(ctx, dig.alloc(ctx, 4 * B * sizeof(fr_t)));
    STARK_HIP(ctx, hipMemcpyAsync(dptr.p, ptrs.data(), ptrs.size() * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
    const char* tags[4] = {"ALI/A", "ALI/S", "ALI/E", "ALI/T"};
    STARK_TRY(tr_hash_columns_batch_dev(ctx, tags, (const fr_t* const*)dptr.p, B, n0, dig.fr()));
    std::vector<fr_t> h(4 * B);
    STARK_HIP(ctx, hipMemcpyAsync(h.data(), dig.p, 4 * B * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // (2) seed_f = H("ALI/seed", [h_a, h_s, h_e, h_t, n0]) and the fused hash of ali_sample_z_beta_fs, one launch each for the batch (fri.rs:556-557, 511-515)
    const fr_t n0f = host::h_u64(n0);
    std::vector<fr_t> in5(5 * B); for (size_t p = 0; p < B; ++p) { for (int c = 0; c < 4; ++c) in5[5 * p + c] = h[4 * p + c]; in5[5 * p + 4] = n0f; }
    STARK_HIP(ctx, seeds_in.alloc(ctx, in5.size() * sizeof(fr_t))); STARK_HIP(ctx, seeds.alloc(ctx, B * sizeof(fr_t)));
    STARK_HIP(ctx, hipMemcpyAsync(seeds_in.p, in5.data(), in5.size() * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    STARK_TRY(tr_hash_dev(ctx, "ALI/seed", seeds_in.fr(), 5, B, seeds.fr()));
    std::vector<fr_t> seed_f(B);
    STARK_HIP(ctx, hipMemcpyAsync(seed_f.data(), seeds.p, B * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<fr_t> in2(2 * B); for (size_t p = 0; p < B; ++p) { in2[2 * p] = seed_f[p]; in2[2 * p + 1] = n0f; }
    STARK_HIP(ctx, deep_in.alloc(ctx, in2.size() * sizeof(fr_t))); STARK_HIP(ctx, fused.alloc(ctx, B * sizeof(fr_t)));
    STARK_HIP(ctx, hipMemcpyAsync(deep_in.p, in2.data(), in2.size() * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    STARK_TRY(tr_hash_dev(ctx, "ALI/DEEP", deep_in.fr(), 2, B, fused.fr()));
    std::vector<fr_t> fu(B);
    STARK_HIP(ctx, hipMemcpyAsync(fu.data(), fused.p, B * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    auto t1 = now();
    // (3) per trace: merge, commit phase, query phase.  These tails are latency-bound (a few ms of small dependent launches each), so up to four of them run
    // side by side: worker contexts of this context (same device, private streams, own pools), one host thread each, traces dealt round-robin.  The inputs
    // are resident and this context's stream is idle (synchronised above), so the workers' streams may read them.
    const fr_t omega = fr_root_of_unity<PallasFr>((unsigned)ilog2(n0));
    const double shared_ms = ms_of(t0, t1);
    const size_t NT = std::min<size_t>(B, 4);
    std::vector<stark_ctx*> cx(NT); for (size_t w = 0; w < NT; ++w) STARK_TRY(ctx_aux(ctx, w, &cx[w]));
    std::vector<int32_t> rcs(NT, STARK_OK);
    auto worker = [&](size_t w) {
        stark_ctx* c = cx[w];
        int32_t rc = ctx_enter(c); if (rc) { rcs[w] = rc; return; }
        DevBuf f0buf; if (f0buf.alloc(c, n0 * sizeof(fr_t)) != hipSuccess) { rcs[w] = c->fail(STARK_ERR_OOM, "f0"); return; }
        for (size_t p = w; p < B; p += NT) {
            auto u0 = now();
            fr_t z, beta; ali_z_beta_from_fused(fu[p], n0, seed_f[p], &z, &beta);
            rc = ali_merge_dev_impl(c, as_fr(a[p]), as_fr(s[p]), as_fr(e[p]), as_fr(t[p]), nullptr, host::h_zero(), omega, z, n0, f0buf.fr(), nullptr); if (rc) { rcs[w] = rc; return; }
            auto u1 = now();
            stark_fri_state* S = nullptr; rc = fri_build_impl(c, f0buf.fr(), n0, schedule, L, seed_z, &S); if (rc) { rcs[w] = rc; return; }
            auto u2 = now();
            stark_proof* P = new stark_proof();
            rc = prove_queries_encode(c, S, n0, r, P); delete S; if (rc) { delete P; rcs[w] = rc; return; }
            auto u3 = now();
            P->ms[0] = shared_ms + ms_of(u0, u1);          // the shared sponge stage (whole batch) + this trace's merge
            P->ms[1] = ms_of(u1, u2); P->ms[2] = ms_of(u2, u3);
            out[p] = P;
        }
        (void)hipStreamSynchronize(c->stream);
    };
    if (NT == 1) worker(0);
    else { std::vector<std::thread> th; for (size_t w = 0; w < NT; ++w) th.emplace_back(worker, w); for (auto& x : th) x.join(); }
    STARK_TRY(ctx_enter(ctx));
    for (size_t w = 0; w < NT; ++w) if (rcs[w]) {
        ctx->err = cx[w]->err;
        for (size_t p = 0; p < B; ++p) if (out[p]) { delete out[p]; out[p] = nullptr; }
        return rcs[w];
    }
    return STARK_OK;
}

extern "C" {

int32_t stark_fri_sample_z(stark_ctx_t* ctx, stark_params_t* tp, uint64_t seed_z, size_t level, size_t domain_size, uint64_t* z4) {
    if (!ctx || !z4) return STARK_ERR_INVALID_ARG; (void)tp;
    fr_t z; STARK_TRY(sample_z(ctx, seed_z, level, domain_size, &z)); store_fr(z4, z); return STARK_OK;
}
int32_t stark_fri_fold_dev(stark_ctx_t* ctx, const uint64_t* f, size_t n, const uint64_t* z4, size_t m, uint64_t* out) {
    if (!ctx || !z4 || (!f && n) || (!out && n)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    return fold_dev(ctx, as_fr(f), n, load_fr(z4), m, as_fr(out));
}
int32_t stark_fri_fold(stark_ctx_t* ctx, const uint64_t* f, size_t n, const uint64_t* z4, size_t m, uint64_t* out) {
    if (!ctx || !z4 || (!f && n) || (!out && n)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (m < 2) return ctx->fail(STARK_ERR_INVALID_ARG, "m >= 2"); if (n % m) return ctx->fail(STARK_ERR_INVALID_ARG, "layer size must be divisible by m");
    DevBuf df, dout; STARK_HIP(ctx, df.alloc(ctx, n * sizeof(fr_t))); STARK_HIP(ctx, dout.alloc(ctx, n / m * sizeof(fr_t)));
    if (n) STARK_HIP(ctx, hipMemcpyAsync(df.p, f, n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    STARK_TRY(fold_dev(ctx, df.fr(), n, load_fr(z4), m, dout.fr()));
    if (n) STARK_HIP(ctx, hipMemcpyAsync(out, dout.p, n / m * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}
int32_t stark_fri_build_dev(stark_ctx_t* ctx, const uint64_t* f0, size_t n0, const size_t* schedule, size_t L, uint64_t seed_z, stark_fri_state_t** out) {
    if (!ctx || !f0 || !out || (!schedule && L)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    return fri_build_impl(ctx, as_fr(f0), n0, schedule, L, seed_z, out);
}
int32_t stark_fri_build(stark_ctx_t* ctx, const uint64_t* f0, size_t n0, const size_t* schedule, size_t L, uint64_t seed_z, stark_fri_state_t** out) {
    if (!ctx || !f0 || !out || (!schedule && L)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    DevBuf d; STARK_HIP(ctx, d.alloc(ctx, n0 * sizeof(fr_t))); STARK_HIP(ctx, hipMemcpyAsync(d.p, f0, n0 * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    STARK_TRY(fri_build_impl(ctx, d.fr(), n0, schedule, L, seed_z, out)); return STARK_OK;
}
int32_t stark_fri_num_layers(stark_fri_state_t* s) { return s ? (int32_t)s->f.size() : STARK_ERR_INVALID_ARG; }
size_t stark_fri_layer_len(stark_fri_state_t* s, int32_t l) { return (s && l >= 0 && (size_t)l < s->n.size()) ? s->n[l] : 0; }
int32_t stark_fri_layer_f(stark_fri_state_t* s, int32_t l, uint64_t* out) {
    if (!s || !out || l < 0 || (size_t)l >= s->f.size()) return STARK_ERR_INVALID_ARG; stark_ctx* ctx = s->ctx;
    STARK_HIP(ctx, hipMemcpyAsync(out, s->f[l], s->n[l] * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}
int32_t stark_fri_layer_root(stark_fri_state_t* s, int32_t l, uint64_t* out4) {
    if (!s || !out4 || l < 0 || (size_t)l >= s->trees.size()) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(s->ctx)); STARK_TRY(state_roots(s)); store_fr(out4, s->roots[l]); return STARK_OK;
}
int32_t stark_fri_layer_z(stark_fri_state_t* s, int32_t l, uint64_t* out4) { if (!s || !out4 || l < 0 || (size_t)l >= s->z.size()) return STARK_ERR_INVALID_ARG; store_fr(out4, s->z[l]); return STARK_OK; }
stark_tree_t* stark_fri_layer_tree(stark_fri_state_t* s, int32_t l) { return (s && l >= 0 && (size_t)l < s->trees.size()) ? s->trees[l] : nullptr; }
int32_t stark_fri_state_free(stark_fri_state_t* s) { if (!s) return STARK_ERR_INVALID_ARG; delete s; return STARK_OK; }   // layers and levels return to the pool (stream-ordered reuse)

int32_t stark_ali_merge_dev(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* r_opt, const uint64_t* beta4,
                            const uint64_t* omega4, const uint64_t* z4, size_t n, uint64_t* f0, uint64_t* c_star4) {
    if (!ctx || !a || !s || !e || !t || !omega4 || !z4 || !f0 || (r_opt && !beta4)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    fr_t cs; STARK_TRY(ali_merge_dev_impl(ctx, as_fr(a), as_fr(s), as_fr(e), as_fr(t), as_fr(r_opt), beta4 ? load_fr(beta4) : host::h_zero(), load_fr(omega4), load_fr(z4), n, as_fr(f0), c_star4 ? &cs : nullptr));
    if (c_star4) store_fr(c_star4, cs); return STARK_OK;
}
int32_t stark_ali_merge(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* r_opt, const uint64_t* beta4,
                        const uint64_t* omega4, const uint64_t* z4, size_t n, uint64_t* f0, uint64_t* c_star4) {
    if (!ctx || !a || !s || !e || !t || !omega4 || !z4 || !f0 || (r_opt && !beta4)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    DevBuf d[6]; const uint64_t* src[5] = {a, s, e, t, r_opt};
    for (int i = 0; i < 5; ++i) if (src[i]) { STARK_HIP(ctx, d[i].alloc(ctx, n * sizeof(fr_t))); STARK_HIP(ctx, hipMemcpyAsync(d[i].p, src[i], n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream)); }
    STARK_HIP(ctx, d[5].alloc(ctx, n * sizeof(fr_t)));
    STARK_TRY(stark_ali_merge_dev(ctx, (const uint64_t*)d[0].p, (const uint64_t*)d[1].p, (const uint64_t*)d[2].p, (const uint64_t*)d[3].p, r_opt ? (const uint64_t*)d[4].p : nullptr, beta4, omega4, z4, n, (uint64_t*)d[5].p, c_star4));
    STARK_HIP(ctx, hipMemcpyAsync(f0, d[5].p, n * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}
int32_t stark_build_f0_dev(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, size_t n0, uint64_t* f0, uint64_t* aux7) {
    if (!ctx || !a || !s || !e || !t || !f0) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    fr_t aux[7]; STARK_TRY(build_f0_dev_impl(ctx, as_fr(a), as_fr(s), as_fr(e), as_fr(t), n0, as_fr(f0), aux7 ? aux : nullptr));
    if (aux7) for (int i = 0; i < 7; ++i) store_fr(aux7 + 4 * i, aux[i]); return STARK_OK;
}
int32_t stark_build_f0(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, size_t n0, uint64_t* f0, uint64_t* aux7) {
    if (!ctx || !a || !s || !e || !t || !f0) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    DevBuf d[5]; const uint64_t* src[4] = {a, s, e, t};
    for (int i = 0; i < 4; ++i) { STARK_HIP(ctx, d[i].alloc(ctx, n0 * sizeof(fr_t))); STARK_HIP(ctx, hipMemcpyAsync(d[i].p, src[i], n0 * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream)); }
    STARK_HIP(ctx, d[4].alloc(ctx, n0 * sizeof(fr_t)));
    STARK_TRY(stark_build_f0_dev(ctx, (const uint64_t*)d[0].p, (const uint64_t*)d[1].p, (const uint64_t*)d[2].p, (const uint64_t*)d[3].p, n0, (uint64_t*)d[4].p, aux7));
    STARK_HIP(ctx, hipMemcpyAsync(f0, d[4].p, n0 * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}

int32_t stark_deep_fri_prove_dev(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* f0, size_t n0,
                                 const size_t* schedule, size_t L, size_t r, uint64_t seed_z, stark_proof_t** out) {
    if (!ctx || !out || (!schedule && L) || (!f0 && (!a || !s || !e || !t))) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    return prove_impl(ctx, as_fr(a), as_fr(s), as_fr(e), as_fr(t), as_fr(f0), n0, schedule, L, r, seed_z, out);
}
int32_t stark_deep_fri_prove(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* f0, size_t n0,
                             const size_t* schedule, size_t L, size_t r, uint64_t seed_z, stark_proof_t** out) {
    if (!ctx || !out || (!schedule && L) || (!f0 && (!a || !s || !e || !t))) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    DevBuf d[5]; const uint64_t* src[5] = {a, s, e, t, f0};
    for (int i = 0; i < 5; ++i) if ((i < 4 && !f0) || (i == 4 && f0)) { STARK_HIP(ctx, d[i].alloc(ctx, n0 * sizeof(fr_t))); STARK_HIP(ctx, hipMemcpyAsync(d[i].p, src[i], n0 * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream)); }
    return stark_deep_fri_prove_dev(ctx, (const uint64_t*)d[0].p, (const uint64_t*)d[1].p, (const uint64_t*)d[2].p, (const uint64_t*)d[3].p, f0 ? (const uint64_t*)d[4].p : nullptr, n0, schedule, L, r, seed_z, out);
}
int32_t stark_deep_fri_prove_batch_dev(stark_ctx_t* ctx, size_t batch, const uint64_t* const* a, const uint64_t* const* s, const uint64_t* const* e, const uint64_t* const* t, size_t n0,
                                       const size_t* schedule, size_t L, size_t r, uint64_t seed_z, stark_proof_t** out) {
    if (!ctx || !out || !batch || !a || !s || !e || !t || (!schedule && L)) return STARK_ERR_INVALID_ARG;
    for (size_t p = 0; p < batch; ++p) if (!a[p] || !s[p] || !e[p] || !t[p]) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    return prove_batch_impl(ctx, batch, a, s, e, t, n0, schedule, L, r, seed_z, out);
}
size_t stark_proof_len(stark_proof_t* p) { return p ? p->bytes.size() : 0; }
int32_t stark_proof_bytes(stark_proof_t* p, uint8_t* out) { if (!p || !out) return STARK_ERR_INVALID_ARG; memcpy(out, p->bytes.data(), p->bytes.size()); return STARK_OK; }
size_t stark_proof_size_estimate(stark_proof_t* p) { return p ? p->size_estimate : 0; }
double stark_proof_stage_ms(stark_proof_t* p, int32_t stage) { return (p && stage >= 0 && stage < 3) ? p->ms[stage] : -1.0; }
int32_t stark_proof_free(stark_proof_t* p) { if (!p) return STARK_ERR_INVALID_ARG; delete p; return STARK_OK; }

// ---- one trace sharded over several GPUs: the pieces the orchestrator (stark_mlwe_amd/dist.py) composes --------
int32_t stark_ali_merge_shard_dev(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* r_opt, const uint64_t* beta4,
                                  const uint64_t* omega4, const uint64_t* z4, size_t n_local, uint64_t j0, size_t n_global, uint64_t* f0, uint64_t* partial4) {
    if (!ctx || !a || !s || !e || !t || !z4 || !f0 || (r_opt && !beta4) || !n_global) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    const fr_t omega = omega4 ? load_fr(omega4) : fr_root_of_unity<PallasFr>((unsigned)ilog2(n_global));               // FriDomain::new_radix2(n).omega, fri.rs:53-56
    fr_t ps; STARK_TRY(ali_merge_dev_impl(ctx, as_fr(a), as_fr(s), as_fr(e), as_fr(t), as_fr(r_opt), beta4 ? load_fr(beta4) : host::h_zero(), omega, load_fr(z4), n_local, as_fr(f0),
                                          partial4 ? &ps : nullptr, j0, n_global, true));
    if (partial4) store_fr(partial4, ps); return STARK_OK;
}
int32_t stark_ali_cstar_from_partials(stark_ctx_t* ctx, const uint64_t* partials, size_t k, size_t n_global, uint64_t* c_star4) {
    if (!ctx || (!partials && k) || !c_star4 || !n_global) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    fr_t acc = host::h_zero(); for (size_t i = 0; i < k; ++i) acc = host::h_add(acc, load_fr(partials + 4 * i));
    store_fr(c_star4, fr_mul<PallasFr>(acc, fr_inv<PallasFr>(host::h_u64(n_global)))); return STARK_OK;                  // c* = (1/n) * sum (lib.rs:44, :94)
}
int32_t stark_ali_challenges(stark_ctx_t* ctx, const uint64_t* digests16, size_t n0, uint64_t* aux12) {
    if (!ctx || !digests16 || !aux12 || n0 <= 1) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    fr_t h[5]; for (int c = 0; c < 4; ++c) h[c] = load_fr(digests16 + 4 * c); h[4] = host::h_u64(n0);
    fr_t seed_f; STARK_TRY(tr_hash_host1(ctx, "ALI/seed", std::vector<fr_t>(h, h + 5), &seed_f));                       // fri.rs:556-557
    fr_t z, beta; STARK_TRY(ali_sample_z_beta(ctx, "ALI/DEEP", n0, seed_f, &z, &beta));
    store_fr(aux12, seed_f); store_fr(aux12 + 4, z); store_fr(aux12 + 8, beta); return STARK_OK;
}
int32_t stark_fri_plan_create(stark_ctx_t* ctx, const uint64_t* roots, size_t n0, const size_t* schedule, size_t L, size_t r, stark_fri_plan_t** out) {
    if (!ctx || !roots || !out || (!schedule && L)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    std::vector<fr_t> rt(L + 1); for (size_t l = 0; l <= L; ++l) rt[l] = load_fr(roots + 4 * l);
    stark_fri_plan* P = new stark_fri_plan(); P->ref_.bind(ctx); P->ctx = ctx; P->plan.r = r;
    std::string err; if (!P->plan.shape.make(n0, schedule, L, rt.data(), err)) { delete P; return ctx->fail(STARK_ERR_INVALID_ARG, err); }
    DeviceHasher H(ctx); int32_t rc = fri_plan_make(P->plan, H);
    if (rc) { delete P; return rc == -1 ? ctx->fail(STARK_ERR_INVALID_ARG, "query plan") : rc; }
    *out = P; return STARK_OK;
}
size_t stark_fri_plan_num_requests(stark_fri_plan_t* p) { return p ? p->plan.req.size() : 0; }
int32_t stark_fri_plan_requests(stark_fri_plan_t* p, uint32_t* kind, uint32_t* which, uint32_t* level, uint64_t* index) {
    if (!p || !kind || !which || !level || !index) return STARK_ERR_INVALID_ARG;
    for (size_t i = 0; i < p->plan.req.size(); ++i) { kind[i] = p->plan.req[i].kind; which[i] = p->plan.req[i].which; level[i] = p->plan.req[i].level; index[i] = p->plan.req[i].index; }
    return STARK_OK;
}
int32_t stark_fri_plan_assemble(stark_fri_plan_t* p, const uint64_t* values, size_t n_values, stark_proof_t** out) {
    if (!p || (!values && n_values) || !out) return STARK_ERR_INVALID_ARG;
    stark_ctx* ctx = p->ctx;
    if (n_values != p->plan.req.size()) return ctx->fail(STARK_ERR_INVALID_ARG, "value count differs from the plan's request count");
    std::vector<fr_t> v(n_values); for (size_t i = 0; i < n_values; ++i) v[i] = load_fr(values + 4 * i);
    ReplaySource src(v.data(), v.size()); DeviceHasher H(ctx);
    stark_proof* P = new stark_proof();
    int32_t rc = assemble_proof(p->plan.shape, p->plan.r, H, src, P->bytes, P->size_estimate);
    if (rc || src.pos != v.size()) { delete P; return ctx->fail(STARK_ERR_INVALID_ARG, "assemble: values do not match the plan"); }
    *out = P; return STARK_OK;
}
int32_t stark_fri_plan_free(stark_fri_plan_t* p) { if (!p) return STARK_ERR_INVALID_ARG; delete p; return STARK_OK; }

}  // extern "C"

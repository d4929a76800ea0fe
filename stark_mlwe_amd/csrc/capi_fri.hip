// stark_mlwe_amd/csrc/capi_fri.hip — FRI folding, fri_build_transcript, DEEP-ALI merge, build_f0 and
// the end-to-end deep_fri_prove orchestration (host logic in C++ above the kernels, mirroring
// crates/deep_ali/src/fri.rs and crates/deep_ali/src/lib.rs).  C-ABI in include/stark_mlwe.h.
#include <algorithm>
#include <chrono>
#include <cstring>
#include "ctx.hpp"
#include "fri_dev.hpp"

using namespace stark;

struct stark_fri_state {
    stark_ctx* ctx = nullptr;
    std::vector<size_t> schedule;
    std::vector<fr_t*> f; std::vector<size_t> n;           // L+1 layers (device)
    std::vector<fr_t> z;                                    // L fold challenges
    std::vector<stark_tree*> trees; std::vector<char> hashed; std::vector<size_t> arity; std::vector<fr_t> roots;
    ~stark_fri_state() { for (auto p : f) if (p) (void)hipFree(p); for (auto t : trees) if (t) stark_merkle_free(t); }
};
struct stark_proof { std::vector<uint8_t> bytes; size_t size_estimate = 0; double ms[3] = {0, 0, 0}; };

static inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
static inline int ilog2(size_t x) { return ilog2_ceil(x); }

// z^0..z^(m-1) on the device (m <= a few hundred: host powers, one small upload).
static int32_t upload_zpows(stark_ctx* ctx, const fr_t& z, size_t m, DevBuf& d) {
    std::vector<fr_t> zp(m); fr_t acc = host::h_one(); for (size_t t = 0; t < m; ++t) { zp[t] = acc; acc = host::h_mul(acc, z); }   // fri.rs:91-93
    STARK_HIP(ctx, d.alloc(m * sizeof(fr_t)));
    STARK_HIP(ctx, hipMemcpyAsync(d.p, zp.data(), m * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // zp is a local: finish the copy before it goes away
    return STARK_OK;
}
static int32_t fold_dev(stark_ctx* ctx, const fr_t* f, size_t n, const fr_t& z, size_t m, fr_t* out) {
    if (m < 2) return ctx->fail(STARK_ERR_INVALID_ARG, "m >= 2");                                        // fri.rs:86
    if (n % m) return ctx->fail(STARK_ERR_INVALID_ARG, "layer size must be divisible by m");             // fri.rs:87
    if (!n) return STARK_OK;
    DevBuf zp; STARK_TRY(upload_zpows(ctx, z, m, zp));
    if (is_pow2(m)) {
        int log_m = ilog2(m), log_g = std::min(log_m, 4);
        uint64_t lanes = (uint64_t)n >> (log_m - log_g);
        hipLaunchKernelGGL(k_fri_fold_pow2<PallasFr>, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, ctx->stream, f, (uint64_t)n, (const fr_t*)zp.fr(), log_m, log_g, out);
    } else {
        uint64_t no = n / m;
        hipLaunchKernelGGL(k_fri_fold_any<PallasFr>, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, ctx->stream, f, no, (const fr_t*)zp.fr(), (uint64_t)m, out);
    }
    STARK_HIP(ctx, hipGetLastError());
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // zp freed on return
    return STARK_OK;
}

// fri_sample_z_ell (fri.rs:59-82): transcript hash on the device, ChaCha12 + candidate test on the host.
static int32_t sample_z(stark_ctx* ctx, uint64_t seed_z, size_t level, size_t domain_size, fr_t* z) {
    fr_t fused; STARK_TRY(tr_hash_host1(ctx, "FRI/z/l", {host::h_u64(seed_z), host::h_u64(level), host::h_u64(domain_size)}, &fused));
    uint8_t seed[32]; host::h_to_bytes_le(fused, seed);
    host::ChaCha12Rng rng(seed);
    const fr_t one = host::h_one();
    for (size_t tries = 0;;) {
        fr_t cand = host::h_u64(rng.next_u64());
        if (!fr_is_zero(cand) && !fr_eq(fr_pow_u64<PallasFr>(cand, domain_size), one)) { *z = cand; return STARK_OK; }
        if (++tries >= 1000) {
            fr_t fb = host::h_u64(seed_z + (uint64_t)level + 7);
            *z = !fr_eq(fr_pow_u64<PallasFr>(fb, domain_size), one) ? fb : host::h_u64(11); return STARK_OK;
        }
    }
}

static int32_t fri_build_impl(stark_ctx* ctx, const fr_t* f0_dev, size_t n0, const size_t* schedule, size_t L, uint64_t seed_z, stark_fri_state** out) {
    if (!n0) return ctx->fail(STARK_ERR_INVALID_ARG, "empty layer");
    { size_t n = n0; for (size_t l = 0; l < L; ++l) { if (schedule[l] < 2 || n % schedule[l]) return ctx->fail(STARK_ERR_INVALID_ARG, "schedule not dividing domain size"); n /= schedule[l]; } }   // fri.rs:150
    stark_fri_state* S = new stark_fri_state(); S->ctx = ctx; S->schedule.assign(schedule, schedule + L);
    auto bail = [&](int32_t rc) { delete S; return rc; };
    stark_params* tp = nullptr; { int32_t rc = ctx_transcript_params(ctx, &tp); if (rc) return bail(rc); }
    // layer 0 copy + folds back to back (the challenges do not depend on any commitment: fri.rs:250)
    fr_t* cur = nullptr; if (hipMalloc((void**)&cur, n0 * sizeof(fr_t)) != hipSuccess) return bail(ctx->fail(STARK_ERR_OOM, "layer 0"));
    S->f.push_back(cur); S->n.push_back(n0);
    if (hipMemcpyAsync(cur, f0_dev, n0 * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) return bail(ctx->fail(STARK_ERR_HIP, "copy f0"));
    for (size_t l = 0; l < L; ++l) {
        fr_t z; { int32_t rc = sample_z(ctx, seed_z, l, S->n[l], &z); if (rc) return bail(rc); }
        S->z.push_back(z);
        size_t nn = S->n[l] / schedule[l]; fr_t* nx = nullptr;
        if (hipMalloc((void**)&nx, nn * sizeof(fr_t)) != hipSuccess) return bail(ctx->fail(STARK_ERR_OOM, "fold layer"));
        S->f.push_back(nx); S->n.push_back(nn);
        int32_t rc = fold_dev(ctx, S->f[l], S->n[l], z, schedule[l], nx); if (rc) return bail(rc);
    }
    // Commitments of all L+1 layers (independent jobs).  Layer 0 is ~94 % of the hashing and fills the GPU; the later layers
    // are small and mostly LATENCY-bound (tree tops: one dependent permutation per level), so they are enqueued first on a
    // side stream and run underneath layer 0 instead of after it.  Temporaries stay alive until both streams have drained
    // (a hipFree in between would synchronise the device and serialise the two again).
    hipStream_t main_stream = ctx->stream, side = nullptr;
    { int32_t rc = ctx_side_stream(ctx, &side); if (rc) return bail(rc); }
    if (hipEventRecord(ctx->ev_fork, main_stream) != hipSuccess || hipStreamWaitEvent(side, ctx->ev_fork, 0) != hipSuccess) return bail(ctx->fail(STARK_ERR_HIP, "fork"));
    std::vector<DevBuf> keep(2 * (L + 1) + 1);
    size_t nkeep = 0;
    S->trees.assign(L + 1, nullptr); S->hashed.assign(L + 1, 0); S->arity.assign(L + 1, 0);
    auto commit_layer = [&](size_t l) -> int32_t {
        size_t n = S->n[l], m_l = l < L ? schedule[l] : 1, arity = pick_arity_for_layer(n, m_l); bool hashed = hashed_arity(arity);
        stark_params* mp = nullptr; STARK_TRY(ctx_merkle_params(ctx, host::width_for_arity(arity), &mp));    // MerkleChannelCfg::new(arity), fri.rs:277
        stark_tree* T = nullptr;
        if (hashed) {
            DevBuf& h = keep[nkeep++]; if (h.alloc(n * sizeof(fr_t)) != hipSuccess) return ctx->fail(STARK_ERR_OOM, "leaf digests");
            STARK_TRY(stark_leaf_pair_hash_dev(ctx, tp, (const uint64_t*)S->f[l], l < L ? (const uint64_t*)S->f[l + 1] : nullptr, n, m_l, (uint64_t*)h.p));     // fri.rs:283 (s = f_{l+1}[i/m] view)
            STARK_TRY(stark_merkle_build_dev(ctx, mp, arity, (uint64_t)l, (const uint64_t*)h.p, n, 0, nullptr, 0, 0, 0, &T));
        } else {
            // commit_pairs(f_l, s_l) (fri.rs:289): s_l is the m-fold replication of f_{l+1}, or zeros on the last layer (fri.rs:266)
            DevBuf& sl = keep[nkeep++]; if (sl.alloc(n * sizeof(fr_t)) != hipSuccess) return ctx->fail(STARK_ERR_OOM, "s layer");
            if (l < L) {
                std::vector<uint64_t> idx(n); for (size_t i = 0; i < n; ++i) idx[i] = i / m_l;
                DevBuf& di = keep[nkeep++]; if (di.alloc(n * 8) != hipSuccess) return ctx->fail(STARK_ERR_OOM, "s idx");
                if (hipMemcpyAsync(di.p, idx.data(), n * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return ctx->fail(STARK_ERR_HIP, "copy idx");
                if (hipStreamSynchronize(ctx->stream) != hipSuccess) return ctx->fail(STARK_ERR_HIP, "idx upload");     // idx is a host temporary
                hipLaunchKernelGGL(k_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const fr_t*)S->f[l + 1], (const uint64_t*)di.p, (uint64_t)n, sl.fr());
            } else if (hipMemsetAsync(sl.p, 0, n * sizeof(fr_t), ctx->stream) != hipSuccess) return ctx->fail(STARK_ERR_HIP, "memset");
            STARK_TRY(stark_merkle_build_dev(ctx, mp, arity, (uint64_t)l, (const uint64_t*)S->f[l], n, 1, (const uint64_t*)sl.p, 0, 0, 0, &T));
        }
        S->trees[l] = T; S->hashed[l] = hashed ? 1 : 0; S->arity[l] = arity;
        return STARK_OK;
    };
    int32_t crc = STARK_OK;
    ctx->stream = side;
    for (size_t l = L; l >= 1 && crc == STARK_OK; --l) crc = commit_layer(l);
    ctx->stream = main_stream;
    if (crc == STARK_OK) crc = commit_layer(0);
    const bool drained = hipStreamSynchronize(side) == hipSuccess && hipStreamSynchronize(main_stream) == hipSuccess;
    if (crc != STARK_OK) return bail(crc);
    if (!drained) return bail(ctx->fail(STARK_ERR_HIP, "sync"));
    for (size_t l = 0; l <= L; ++l) { fr_t root; int32_t rc = stark_merkle_root(S->trees[l], (uint64_t*)&root); if (rc) return bail(rc); S->roots.push_back(root); }
    *out = S; return STARK_OK;
}

// deep_ali_merge_evals_blinded on device pointers (deep_ali/src/lib.rs:60-105).
static int32_t ali_merge_dev_impl(stark_ctx* ctx, const fr_t* a, const fr_t* s, const fr_t* e, const fr_t* t, const fr_t* r_opt, const fr_t& beta,
                                  const fr_t& omega, const fr_t& z, size_t n, fr_t* f0, fr_t* c_star_host,
                                  uint64_t j0 = 0, size_t n_global = 0, bool partial_only = false) {
    // n = number of local elements, holding the global positions j0 .. j0+n-1 of a domain of n_global points (whole vector: j0 = 0, n_global = n)
    if (!n_global) n_global = n;
    if (n_global <= 1 || !n || j0 + n > n_global) return ctx->fail(STARK_ERR_INVALID_ARG, "n > 1");                        // lib.rs:71
    if (fr_eq(fr_pow_u64<PallasFr>(z, n_global), host::h_one())) return ctx->fail(STARK_ERR_INVALID_ARG, "z must be outside H");   // lib.rs:78
    // power table of omega: two levels of 2^ceil(b/2) entries, b = bits of n_global
    int bits = ilog2(n_global); if (bits < 1) bits = 1; int lo_bits = (bits + 1) / 2, hi_bits = bits - lo_bits + 1;
    DevBuf tlo, thi; STARK_HIP(ctx, tlo.alloc(((size_t)1 << lo_bits) * sizeof(fr_t))); STARK_HIP(ctx, thi.alloc(((size_t)1 << hi_bits) * sizeof(fr_t)));
    { uint64_t tot = (1ull << lo_bits) + (1ull << hi_bits);
      hipLaunchKernelGGL(k_fill_pow_table<PallasFr>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, tlo.fr(), thi.fr(), lo_bits, hi_bits, omega, host::h_one());
      STARK_HIP(ctx, hipGetLastError()); }
    PowTable wp{tlo.fr(), thi.fr(), lo_bits};
    const unsigned block = 256; uint64_t lanes = (n + ALI_K - 1) / ALI_K; unsigned grid = (unsigned)((lanes + block - 1) / block);
    const uint64_t T = (uint64_t)grid * block;
    fr_t w_step = fr_pow_u64<PallasFr>(omega, T);
    DevBuf sums; STARK_HIP(ctx, sums.alloc((size_t)grid * sizeof(fr_t)));
    hipLaunchKernelGGL(k_ali_merge<PallasFr>, dim3(grid), dim3(block), 0, ctx->stream, a, s, e, t, r_opt, beta, wp, w_step, fr_inv<PallasFr>(w_step), z, (uint64_t)n, j0, f0, c_star_host ? sums.fr() : (fr_t*)nullptr);
    STARK_HIP(ctx, hipGetLastError());
    if (c_star_host) {
        // c* = phi(z)/Z_H(z) = (1/n) * sum_j phi_j w^j/(z - w^j)   (lib.rs:44 and :94); block partials are reduced on the device
        DevBuf tot; STARK_HIP(ctx, tot.alloc(sizeof(fr_t)));
        hipLaunchKernelGGL(k_sum_single_block<PallasFr>, dim3(1), dim3(256), 0, ctx->stream, (const fr_t*)sums.fr(), (uint64_t)grid, partial_only ? host::h_one() : fr_inv<PallasFr>(host::h_u64(n_global)), tot.fr());
        STARK_HIP(ctx, hipGetLastError());
        STARK_HIP(ctx, hipMemcpyAsync(c_star_host, tot.p, sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}

// ali_sample_z_beta_fs (fri.rs:511-533).
static int32_t ali_sample_z_beta(stark_ctx* ctx, const char* tag, size_t n0, const fr_t& roots_seed, fr_t* z, fr_t* beta) {
    fr_t fused; STARK_TRY(tr_hash_host1(ctx, tag, {roots_seed, host::h_u64(n0)}, &fused));
    uint8_t seed[32]; host::h_to_bytes_le(fused, seed); host::ChaCha12Rng rng(seed);
    *beta = host::h_u64(rng.next_u64());
    const fr_t one = host::h_one();
    for (size_t tries = 0;;) {
        fr_t cand = host::h_u64(rng.next_u64());
        if (!fr_is_zero(cand) && !fr_eq(fr_pow_u64<PallasFr>(cand, n0), one)) { *z = cand; return STARK_OK; }
        if (++tries >= 1000) {
            fr_t fb = host::h_add(roots_seed, host::h_u64(17));
            *z = !fr_eq(fr_pow_u64<PallasFr>(fb, n0), one) ? fb : host::h_u64(19); return STARK_OK;
        }
    }
}
// DeepAliRealBuilder::build_f0 (fri.rs:535-569), default builder: no blinding, ds_tag "ALI/DEEP".
static int32_t build_f0_dev_impl(stark_ctx* ctx, const fr_t* a, const fr_t* s, const fr_t* e, const fr_t* t, size_t n0, fr_t* f0, fr_t* aux7) {
    if (n0 <= 1) return ctx->fail(STARK_ERR_INVALID_ARG, "n0 > 1");
    // four serial column sponges (fri.rs:551-554): one lane per column, inherently sequential in n0
    DevBuf dig; STARK_HIP(ctx, dig.alloc(4 * sizeof(fr_t)));
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
        DevBuf di, dout; STARK_HIP(ctx, di.alloc(idx.size() * 8)); STARK_HIP(ctx, dout.alloc(idx.size() * sizeof(fr_t)));
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
    std::string err;
    if (!sh.make(n0, S->schedule.data(), S->schedule.size(), S->roots.data(), err)) return ctx->fail(STARK_ERR_INVALID_ARG, err);
    return STARK_OK;
}
// fri_prove_queries + payload assembly + canonical encoding (fri.rs:355-466, 613-640): fri_plan.hpp over the local state.
static int32_t prove_queries_encode(stark_ctx* ctx, stark_fri_state* S, size_t n0, size_t r, stark_proof* P) {
    FriShape sh; STARK_TRY(shape_of_state(ctx, S, n0, sh));
    LocalSource src(ctx, S); DeviceHasher H(ctx);
    int32_t rc = assemble_proof(sh, r, H, src, P->bytes, P->size_estimate);
    if (rc == -1) return ctx->fail(STARK_ERR_INVALID_ARG, "query phase: bad index or short value list");
    return rc;
}

// Query plan of a commit phase whose layers live elsewhere (sharded over ranks): see fri_plan.hpp.
struct stark_fri_plan { stark_ctx* ctx = nullptr; FriPlan plan; };

static int32_t prove_impl(stark_ctx* ctx, const fr_t* a, const fr_t* s, const fr_t* e, const fr_t* t, const fr_t* f0_in, size_t n0,
                          const size_t* schedule, size_t L, size_t r, uint64_t seed_z, stark_proof** out) {
    if (!is_pow2(n0)) return ctx->fail(STARK_ERR_INVALID_ARG, "n0 must be a power of two (radix-2 domain)");
    auto now = [] { return std::chrono::steady_clock::now(); };
    stark_proof* P = new stark_proof(); auto t0 = now();
    DevBuf f0buf; const fr_t* f0 = f0_in;
    if (!f0) {
        if (f0buf.alloc(n0 * sizeof(fr_t)) != hipSuccess) { delete P; return ctx->fail(STARK_ERR_OOM, "f0"); }
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

extern "C" {

int32_t stark_fri_sample_z(stark_ctx_t* ctx, stark_params_t* tp, uint64_t seed_z, size_t level, size_t domain_size, uint64_t* z4) {
    if (!ctx || !z4) return STARK_ERR_INVALID_ARG; (void)tp;
    fr_t z; STARK_TRY(sample_z(ctx, seed_z, level, domain_size, &z)); store_fr(z4, z); return STARK_OK;
}
int32_t stark_fri_fold_dev(stark_ctx_t* ctx, const uint64_t* f, size_t n, const uint64_t* z4, size_t m, uint64_t* out) {
    if (!ctx || !z4 || (!f && n) || (!out && n)) return STARK_ERR_INVALID_ARG;
    return fold_dev(ctx, as_fr(f), n, load_fr(z4), m, as_fr(out));
}
int32_t stark_fri_fold(stark_ctx_t* ctx, const uint64_t* f, size_t n, const uint64_t* z4, size_t m, uint64_t* out) {
    if (!ctx || !z4 || (!f && n) || (!out && n)) return STARK_ERR_INVALID_ARG;
    if (m < 2) return ctx->fail(STARK_ERR_INVALID_ARG, "m >= 2"); if (n % m) return ctx->fail(STARK_ERR_INVALID_ARG, "layer size must be divisible by m");
    DevBuf df, dout; STARK_HIP(ctx, df.alloc(n * sizeof(fr_t))); STARK_HIP(ctx, dout.alloc(n / m * sizeof(fr_t)));
    if (n) STARK_HIP(ctx, hipMemcpyAsync(df.p, f, n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    STARK_TRY(fold_dev(ctx, df.fr(), n, load_fr(z4), m, dout.fr()));
    if (n) STARK_HIP(ctx, hipMemcpyAsync(out, dout.p, n / m * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}
int32_t stark_fri_build_dev(stark_ctx_t* ctx, const uint64_t* f0, size_t n0, const size_t* schedule, size_t L, uint64_t seed_z, stark_fri_state_t** out) {
    if (!ctx || !f0 || !out || (!schedule && L)) return STARK_ERR_INVALID_ARG;
    return fri_build_impl(ctx, as_fr(f0), n0, schedule, L, seed_z, out);
}
int32_t stark_fri_build(stark_ctx_t* ctx, const uint64_t* f0, size_t n0, const size_t* schedule, size_t L, uint64_t seed_z, stark_fri_state_t** out) {
    if (!ctx || !f0 || !out || (!schedule && L)) return STARK_ERR_INVALID_ARG;
    DevBuf d; STARK_HIP(ctx, d.alloc(n0 * sizeof(fr_t))); STARK_HIP(ctx, hipMemcpyAsync(d.p, f0, n0 * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    STARK_TRY(fri_build_impl(ctx, d.fr(), n0, schedule, L, seed_z, out)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}
int32_t stark_fri_num_layers(stark_fri_state_t* s) { return s ? (int32_t)s->f.size() : STARK_ERR_INVALID_ARG; }
size_t stark_fri_layer_len(stark_fri_state_t* s, int32_t l) { return (s && l >= 0 && (size_t)l < s->n.size()) ? s->n[l] : 0; }
int32_t stark_fri_layer_f(stark_fri_state_t* s, int32_t l, uint64_t* out) {
    if (!s || !out || l < 0 || (size_t)l >= s->f.size()) return STARK_ERR_INVALID_ARG; stark_ctx* ctx = s->ctx;
    STARK_HIP(ctx, hipMemcpyAsync(out, s->f[l], s->n[l] * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}
int32_t stark_fri_layer_root(stark_fri_state_t* s, int32_t l, uint64_t* out4) { if (!s || !out4 || l < 0 || (size_t)l >= s->roots.size()) return STARK_ERR_INVALID_ARG; store_fr(out4, s->roots[l]); return STARK_OK; }
int32_t stark_fri_layer_z(stark_fri_state_t* s, int32_t l, uint64_t* out4) { if (!s || !out4 || l < 0 || (size_t)l >= s->z.size()) return STARK_ERR_INVALID_ARG; store_fr(out4, s->z[l]); return STARK_OK; }
stark_tree_t* stark_fri_layer_tree(stark_fri_state_t* s, int32_t l) { return (s && l >= 0 && (size_t)l < s->trees.size()) ? s->trees[l] : nullptr; }
int32_t stark_fri_state_free(stark_fri_state_t* s) { if (!s) return STARK_ERR_INVALID_ARG; (void)hipStreamSynchronize(s->ctx->stream); delete s; return STARK_OK; }

int32_t stark_ali_merge_dev(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* r_opt, const uint64_t* beta4,
                            const uint64_t* omega4, const uint64_t* z4, size_t n, uint64_t* f0, uint64_t* c_star4) {
    if (!ctx || !a || !s || !e || !t || !omega4 || !z4 || !f0 || (r_opt && !beta4)) return STARK_ERR_INVALID_ARG;
    fr_t cs; STARK_TRY(ali_merge_dev_impl(ctx, as_fr(a), as_fr(s), as_fr(e), as_fr(t), as_fr(r_opt), beta4 ? load_fr(beta4) : host::h_zero(), load_fr(omega4), load_fr(z4), n, as_fr(f0), c_star4 ? &cs : nullptr));
    if (c_star4) store_fr(c_star4, cs); return STARK_OK;
}
int32_t stark_ali_merge(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* r_opt, const uint64_t* beta4,
                        const uint64_t* omega4, const uint64_t* z4, size_t n, uint64_t* f0, uint64_t* c_star4) {
    if (!ctx || !a || !s || !e || !t || !omega4 || !z4 || !f0 || (r_opt && !beta4)) return STARK_ERR_INVALID_ARG;
    DevBuf d[6]; const uint64_t* src[5] = {a, s, e, t, r_opt};
    for (int i = 0; i < 5; ++i) if (src[i]) { STARK_HIP(ctx, d[i].alloc(n * sizeof(fr_t))); STARK_HIP(ctx, hipMemcpyAsync(d[i].p, src[i], n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream)); }
    STARK_HIP(ctx, d[5].alloc(n * sizeof(fr_t)));
    STARK_TRY(stark_ali_merge_dev(ctx, (const uint64_t*)d[0].p, (const uint64_t*)d[1].p, (const uint64_t*)d[2].p, (const uint64_t*)d[3].p, r_opt ? (const uint64_t*)d[4].p : nullptr, beta4, omega4, z4, n, (uint64_t*)d[5].p, c_star4));
    STARK_HIP(ctx, hipMemcpyAsync(f0, d[5].p, n * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}
int32_t stark_build_f0_dev(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, size_t n0, uint64_t* f0, uint64_t* aux7) {
    if (!ctx || !a || !s || !e || !t || !f0) return STARK_ERR_INVALID_ARG;
    fr_t aux[7]; STARK_TRY(build_f0_dev_impl(ctx, as_fr(a), as_fr(s), as_fr(e), as_fr(t), n0, as_fr(f0), aux7 ? aux : nullptr));
    if (aux7) for (int i = 0; i < 7; ++i) store_fr(aux7 + 4 * i, aux[i]); return STARK_OK;
}
int32_t stark_build_f0(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, size_t n0, uint64_t* f0, uint64_t* aux7) {
    if (!ctx || !a || !s || !e || !t || !f0) return STARK_ERR_INVALID_ARG;
    DevBuf d[5]; const uint64_t* src[4] = {a, s, e, t};
    for (int i = 0; i < 4; ++i) { STARK_HIP(ctx, d[i].alloc(n0 * sizeof(fr_t))); STARK_HIP(ctx, hipMemcpyAsync(d[i].p, src[i], n0 * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream)); }
    STARK_HIP(ctx, d[4].alloc(n0 * sizeof(fr_t)));
    STARK_TRY(stark_build_f0_dev(ctx, (const uint64_t*)d[0].p, (const uint64_t*)d[1].p, (const uint64_t*)d[2].p, (const uint64_t*)d[3].p, n0, (uint64_t*)d[4].p, aux7));
    STARK_HIP(ctx, hipMemcpyAsync(f0, d[4].p, n0 * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}

int32_t stark_deep_fri_prove_dev(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* f0, size_t n0,
                                 const size_t* schedule, size_t L, size_t r, uint64_t seed_z, stark_proof_t** out) {
    if (!ctx || !out || (!schedule && L) || (!f0 && (!a || !s || !e || !t))) return STARK_ERR_INVALID_ARG;
    return prove_impl(ctx, as_fr(a), as_fr(s), as_fr(e), as_fr(t), as_fr(f0), n0, schedule, L, r, seed_z, out);
}
int32_t stark_deep_fri_prove(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* f0, size_t n0,
                             const size_t* schedule, size_t L, size_t r, uint64_t seed_z, stark_proof_t** out) {
    if (!ctx || !out || (!schedule && L) || (!f0 && (!a || !s || !e || !t))) return STARK_ERR_INVALID_ARG;
    DevBuf d[5]; const uint64_t* src[5] = {a, s, e, t, f0};
    for (int i = 0; i < 5; ++i) if ((i < 4 && !f0) || (i == 4 && f0)) { STARK_HIP(ctx, d[i].alloc(n0 * sizeof(fr_t))); STARK_HIP(ctx, hipMemcpyAsync(d[i].p, src[i], n0 * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream)); }
    return stark_deep_fri_prove_dev(ctx, (const uint64_t*)d[0].p, (const uint64_t*)d[1].p, (const uint64_t*)d[2].p, (const uint64_t*)d[3].p, f0 ? (const uint64_t*)d[4].p : nullptr, n0, schedule, L, r, seed_z, out);
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
    const fr_t omega = omega4 ? load_fr(omega4) : fr_root_of_unity<PallasFr>((unsigned)ilog2(n_global));               // FriDomain::new_radix2(n).omega, fri.rs:53-56
    fr_t ps; STARK_TRY(ali_merge_dev_impl(ctx, as_fr(a), as_fr(s), as_fr(e), as_fr(t), as_fr(r_opt), beta4 ? load_fr(beta4) : host::h_zero(), omega, load_fr(z4), n_local, as_fr(f0),
                                          partial4 ? &ps : nullptr, j0, n_global, true));
    if (partial4) store_fr(partial4, ps); return STARK_OK;
}
int32_t stark_ali_cstar_from_partials(stark_ctx_t* ctx, const uint64_t* partials, size_t k, size_t n_global, uint64_t* c_star4) {
    if (!ctx || (!partials && k) || !c_star4 || !n_global) return STARK_ERR_INVALID_ARG;
    fr_t acc = host::h_zero(); for (size_t i = 0; i < k; ++i) acc = host::h_add(acc, load_fr(partials + 4 * i));
    store_fr(c_star4, fr_mul<PallasFr>(acc, fr_inv<PallasFr>(host::h_u64(n_global)))); return STARK_OK;                  // c* = (1/n) * sum (lib.rs:44, :94)
}
int32_t stark_ali_challenges(stark_ctx_t* ctx, const uint64_t* digests16, size_t n0, uint64_t* aux12) {
    if (!ctx || !digests16 || !aux12 || n0 <= 1) return STARK_ERR_INVALID_ARG;
    fr_t h[5]; for (int c = 0; c < 4; ++c) h[c] = load_fr(digests16 + 4 * c); h[4] = host::h_u64(n0);
    fr_t seed_f; STARK_TRY(tr_hash_host1(ctx, "ALI/seed", std::vector<fr_t>(h, h + 5), &seed_f));                       // fri.rs:556-557
    fr_t z, beta; STARK_TRY(ali_sample_z_beta(ctx, "ALI/DEEP", n0, seed_f, &z, &beta));
    store_fr(aux12, seed_f); store_fr(aux12 + 4, z); store_fr(aux12 + 8, beta); return STARK_OK;
}
int32_t stark_fri_plan_create(stark_ctx_t* ctx, const uint64_t* roots, size_t n0, const size_t* schedule, size_t L, size_t r, stark_fri_plan_t** out) {
    if (!ctx || !roots || !out || (!schedule && L)) return STARK_ERR_INVALID_ARG;
    std::vector<fr_t> rt(L + 1); for (size_t l = 0; l <= L; ++l) rt[l] = load_fr(roots + 4 * l);
    stark_fri_plan* P = new stark_fri_plan(); P->ctx = ctx; P->plan.r = r;
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

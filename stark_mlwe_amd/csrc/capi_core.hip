// stark_mlwe_amd/csrc/capi_core.hip — context, memory, Poseidon constants, Poseidon / Merkle entry points.
// C-ABI declared in include/stark_mlwe.h.  No CPU compute fallback anywhere in this file: every
// bulk operation is a kernel launch on the context's stream.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include "ctx.hpp"
#include "poseidon_dev.hpp"
#include "poseidon_pair.hpp"
#include "poseidon_coop.hpp"
#include "poseidon_chain.hpp"
#include "poseidon_wave.hpp"
#include "fri_dev.hpp"

using namespace stark;

static const size_t kMaxLds = 160 * 1024;
static inline int poseidon_block(int t) { return (size_t)t * 32 * 64 <= kMaxLds ? 64 : 32; }
static inline size_t poseidon_lds(int t, int block) { return (size_t)t * 32 * block; }
// the wave-pair kernels (poseidon_pair.hpp) serve the hot widths; the "poseidon_lane_only" option selects the one-lane-per-sponge form (diagnostic)
static inline bool use_pair(const stark_ctx* ctx, int t) { return !ctx->opt_poseidon_lane_only && (t == 9 || t == 17); }
// Small batches of t = 17 sponges take the five-wave latency form of poseidon_chain.hpp (one workgroup per sponge, one resident per CU): Merkle levels of up
// to 256 nodes (two permutations: 155 us against 290 us on one wave each; equal from 512 nodes on), leaf layers of up to 2048 leaves (one permutation:
// 80 us per 256 leaves against the 0.77 ms a launch of the wave-pair throughput kernel takes whatever its size).  tools/latency_timing.py;
// option "sponge_one_wave" keeps them on the one-wave / wave-pair kernels (comparison).
constexpr size_t kChainMaxNodes = 256, kChainMaxLeaves = 2048, kCoopMaxLeaves = 4096, kCoopMaxNodes = 4096;      // leaf layers above kChainMaxLeaves: one wave per leaf up to kCoopMaxLeaves
static inline bool use_chain(const stark_ctx* ctx, const PoseidonDev& d, size_t n, size_t n_max) {
    return !ctx->opt_poseidon_lane_only && !ctx->opt_sponge_one_wave && d.t == 17 && d.rf == 8 && d.rp == 64 && d.chain_a && n <= n_max;
}
static inline row::Consts row_consts_of(const stark_ctx* ctx) {
    const RowConstsHost h = row_consts_host(); row::Consts RK; for (int i = 0; i < 9; ++i) RK.ni[i] = h.ni[i]; for (int i = 0; i < 5; ++i) RK.t[i] = h.t[i]; RK.dbg = (uint32_t)ctx->opt_sponge_debug; return RK;
}

namespace stark {

int32_t ctx_scratch(stark_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->scratch_bytes) {
        if (ctx->scratch) { STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->scratch); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
        STARK_HIP(ctx, hipMalloc(&ctx->scratch, bytes)); ctx->scratch_bytes = bytes;
    }
    *out = ctx->scratch; return STARK_OK;
}

// ---- caching device allocator (stark_ctx::pool_free) ----------------------------------------------------------
static inline size_t pool_round(size_t bytes) {
    if (bytes < 256) return 256;
    if (bytes <= (1u << 20)) { size_t r = 256; while (r < bytes) r <<= 1; return r; }       // small: powers of two
    return (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);                             // large: whole MiB (layer / level sizes repeat exactly)
}
int32_t ctx_alloc(stark_ctx* ctx, size_t bytes, void** out) {
    const size_t sz = pool_round(bytes);
    auto it = ctx->pool_free.find(sz);
    if (it != ctx->pool_free.end() && !it->second.empty()) {
        void* p = it->second.back(); it->second.pop_back(); ctx->pool_cached_bytes -= sz;
        ctx->pool_live[p] = sz; *out = p; return STARK_OK;
    }
    void* p = nullptr; hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {                                       // out of memory: give the cached blocks back and retry once
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->stream);
        for (auto& kv : ctx->pool_free) { for (void* q : kv.second) (void)hipFree(q); kv.second.clear(); }
        ctx->pool_cached_bytes = 0;
        e = hipMalloc(&p, sz);
        if (e != hipSuccess) { (void)hipGetLastError(); return ctx->fail(STARK_ERR_OOM, "device allocation of " + std::to_string(sz) + " bytes failed"); }
    }
    ctx->pool_live[p] = sz; *out = p; return STARK_OK;
}
void ctx_release(stark_ctx* ctx, void* p) {
    if (!p || !ctx) return;
    auto it = ctx->pool_live.find(p);
    if (it == ctx->pool_live.end()) { (void)hipFree(p); return; }            // not ours (defensive)
    const size_t sz = it->second; ctx->pool_live.erase(it);
    ctx->pool_free[sz].push_back(p); ctx->pool_cached_bytes += sz;
}
int32_t hash_ds_scattered(stark_ctx* ctx, stark_params* p, int mode, size_t arity, size_t chunk, uint32_t level, uint64_t label, const uint64_t* positions_dev,
                          const fr_t* in0, const fr_t* in1, size_t n_hashes, fr_t* out);
int32_t ctx_enter(stark_ctx* ctx) {
    if (!ctx) return STARK_ERR_INVALID_ARG;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != ctx->device) STARK_HIP(ctx, hipSetDevice(ctx->device));
    return STARK_OK;
}

int32_t ctx_side_stream(stark_ctx* ctx, hipStream_t* out) {
    if (!ctx->side_stream) { STARK_HIP(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking)); STARK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming)); }
    *out = ctx->side_stream; return STARK_OK;
}

static int32_t params_finish(stark_ctx* ctx, stark_params* P) {
    if (host::rp_for_width(P->ref.t) < 0) return ctx->fail(STARK_ERR_UNSUPPORTED, "Poseidon width must be one of 9,17,33,65,129");
    P->kc = host::make_kernel_consts(P->ref);
    if (!P->kc.ok) return ctx->fail(STARK_ERR_UNSUPPORTED, "MDS matrix has a singular leading minor: LU/sparse kernel form unavailable");
    const host::KernelConsts& k = P->kc;
    std::vector<fr_t> blob; auto put = [&](const std::vector<fr_t>& v) { size_t off = blob.size(); blob.insert(blob.end(), v.begin(), v.end()); return off; };
    size_t o_rcf = put(k.rc_full), o_rcp = put(k.rc_partial), o_lu = put(k.lu), o_pre = put(k.lu_pre), o_row0 = put(k.row0), o_sp = put(k.sparse), o_mds = put(k.mds), o_mpre = put(k.mds_pre), o_gam = put(k.gamma);
    // radix-2^29 multiplier tables (fr29.hpp), appended to the same device blob as raw words
    const size_t o_29 = blob.size();
    { std::vector<uint32_t> w29; for (auto* v : {&k.lu29, &k.lu_pre29, &k.row0_29, &k.sparse29, &k.gamma29, &k.mds29, &k.mds_pre29, &k.chain_a, &k.chain_g, &k.chain_w}) w29.insert(w29.end(), v->begin(), v->end());
      while (w29.size() % 8) w29.push_back(0);
      blob.resize(o_29 + w29.size() / 8); memcpy((void*)(blob.data() + o_29), w29.data(), w29.size() * 4); }
    // int8 MFMA fragments of the dense matrices (t = 17), raw bytes in the same blob
    const size_t o_frag = blob.size(), frag_elems = (k.mds_frag.size() + sizeof(fr_t) - 1) / sizeof(fr_t);
    if (!k.mds_frag.empty()) { blob.resize(o_frag + 2 * frag_elems); memcpy((void*)(blob.data() + o_frag), k.mds_frag.data(), k.mds_frag.size()); memcpy((void*)(blob.data() + o_frag + frag_elems), k.mds_pre_frag.data(), k.mds_pre_frag.size()); }
    STARK_HIP(ctx, hipMalloc((void**)&P->blob, blob.size() * sizeof(fr_t)));
    STARK_HIP(ctx, hipMemcpyAsync(P->blob, blob.data(), blob.size() * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    P->dev.t = k.t; P->dev.rf = k.rf; P->dev.rp = k.rp;
    P->dev.rc_full = P->blob + o_rcf; P->dev.rc_partial = P->blob + o_rcp; P->dev.lu = P->blob + o_lu; P->dev.lu_pre = P->blob + o_pre;
    P->dev.row0 = P->blob + o_row0; P->dev.sparse = P->blob + o_sp; P->dev.mds = P->blob + o_mds; P->dev.mds_pre = P->blob + o_mpre; P->dev.gamma = P->blob + o_gam;
    { const uint32_t* b29 = reinterpret_cast<const uint32_t*>(P->blob + o_29);
      P->dev.lu29 = b29; P->dev.lu_pre29 = b29 + k.lu29.size(); P->dev.row0_29 = P->dev.lu_pre29 + k.lu_pre29.size();
      P->dev.sparse29 = P->dev.row0_29 + k.row0_29.size(); P->dev.gamma29 = P->dev.sparse29 + k.sparse29.size();
      P->dev.mds29 = P->dev.gamma29 + k.gamma29.size(); P->dev.mds_pre29 = P->dev.mds29 + k.mds29.size();
      const uint32_t* ch = P->dev.mds_pre29 + k.mds_pre29.size();
      P->dev.chain_a = k.chain_a.empty() ? nullptr : ch; P->dev.chain_g = k.chain_a.empty() ? nullptr : ch + k.chain_a.size(); P->dev.chain_w = k.chain_a.empty() ? nullptr : ch + k.chain_a.size() + k.chain_g.size(); }
    P->dev.mds_frag = k.mds_frag.empty() ? nullptr : (const void*)(P->blob + o_frag); P->dev.mds_pre_frag = k.mds_frag.empty() ? nullptr : (const void*)(P->blob + o_frag + frag_elems);
    return STARK_OK;
}
static int32_t params_from_consts(stark_ctx* ctx, const host::PoseidonConsts& c, stark_params** out) {
    stark_params* P = new stark_params(); P->ctx = ctx; P->ref = c;
    int32_t rc = params_finish(ctx, P);
    if (rc != STARK_OK) { if (P->blob) (void)hipFree(P->blob); delete P; return rc; }
    *out = P; return STARK_OK;
}
int32_t ctx_transcript_params(stark_ctx* ctx, stark_params** out) {
    if (!ctx->tparams) STARK_TRY(params_from_consts(ctx, host::consts_transcript(), &ctx->tparams));
    if (out) *out = ctx->tparams;
    return STARK_OK;
}
int32_t ctx_merkle_params(stark_ctx* ctx, int t, stark_params** out) {
    auto it = ctx->merkle_params.find(t);
    if (it == ctx->merkle_params.end()) {
        if (host::rp_for_width(t) < 0) return ctx->fail(STARK_ERR_UNSUPPORTED, "unsupported Poseidon width");
        stark_params* P = nullptr; STARK_TRY(params_from_consts(ctx, host::consts_for_width(t), &P));
        ctx->merkle_params[t] = P; *out = P; return STARK_OK;
    }
    *out = it->second; return STARK_OK;
}

// Transcript framing for `tr_hash_fields_tagged(tag, xs)` (fri.rs:28-35 over transcript/src/lib.rs:55-101):
// absorbed stream = [AB, w("FRI/FS"), AB, words(tag).., xs.., CH, AB, w("out")], capacity lane = INIT.
static int32_t tr_frame(stark_ctx* ctx, const char* label, const char* tag, const char* out_label, fr_t** dev, int* np, int* ns) {
    std::string key = std::string(label) + "\x01" + tag + "\x01" + out_label;
    auto it = ctx->tr_frames.find(key);
    if (it == ctx->tr_frames.end()) {
        std::vector<fr_t> fr; const fr_t AB = host::h_tag("FSv1-ABSORB-BYTES"), CH = host::h_tag("FSv1-CHALLENGE");
        fr.push_back(AB); for (auto& w : host::h_words(label)) fr.push_back(w);
        fr.push_back(AB); for (auto& w : host::h_words(tag)) fr.push_back(w);
        int p = (int)fr.size();
        fr.push_back(CH); fr.push_back(AB); for (auto& w : host::h_words(out_label)) fr.push_back(w);
        int s = (int)fr.size() - p;
        fr_t* d = nullptr; STARK_HIP(ctx, hipMalloc((void**)&d, fr.size() * sizeof(fr_t)));
        STARK_HIP(ctx, hipMemcpyAsync(d, fr.data(), fr.size() * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
        STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->tr_frames[key] = d; ctx->tr_frame_dims[key] = {p, s};
        it = ctx->tr_frames.find(key);
    }
    *dev = it->second; *np = ctx->tr_frame_dims[key].first; *ns = ctx->tr_frame_dims[key].second; return STARK_OK;
}
static int32_t launch_column_sponges(stark_ctx* ctx, stark_params* tp, const TrMultiJob& J, unsigned nblocks, fr_t* out_dev);
int32_t tr_hash_dev(stark_ctx* ctx, const char* tag, const fr_t* fields_dev, size_t k, size_t n, fr_t* out_dev) {
    hipStream_t st = ctx->stream;
    stark_params* tp = nullptr; STARK_TRY(ctx_transcript_params(ctx, &tp));
    fr_t* frame = nullptr; int np = 0, ns = 0; STARK_TRY(tr_frame(ctx, "FRI/FS", tag, "out", &frame, &np, &ns));
    if (n == 0) return STARK_OK;
    TrJob J; J.prefix = frame; J.np = np; J.suffix = frame + np; J.ns = ns; J.cap = host::h_tag("FSv1-TRANSCRIPT-INIT"); J.k = k; J.n = n;
    if (!ctx->opt_poseidon_lane_only && !ctx->opt_sponge_one_wave && tp->dev.chain_a && n <= 512) {
        // up to two resident workgroups per CU: five waves per sponge (poseidon_chain.hpp), 72 us per permutation against 142 us on one wave —
        // a column digest of a sharded prove, a long Fiat-Shamir input, and the short challenge hashes of the query phase alike
        TrMultiJob M; M.cap = J.cap; M.batch = nullptr; M.stride = k;
        for (int c = 0; c < 4; ++c) { M.prefix[c] = frame; M.np[c] = np; M.suffix[c] = frame + np; M.ns[c] = ns; M.fields[c] = fields_dev; M.k[c] = k; }
        return launch_column_sponges(ctx, tp, M, (unsigned)n, out_dev);
    }
    if (!ctx->opt_poseidon_lane_only && n <= 4096) {
        // few (or one, possibly very long) sponges: one wave per sponge, latency-oriented (poseidon_coop.hpp)
        hipLaunchKernelGGL(k_tr_hash_coop, dim3((unsigned)n), dim3(64), coop_lds_bytes(17), st, tp->dev, J, fields_dev, out_dev);
        STARK_HIP(ctx, hipGetLastError());
        return STARK_OK;
    }
    const int block = 64; const size_t lds = poseidon_lds(17, block);
    hipLaunchKernelGGL(k_tr_hash, dim3((unsigned)((n + block - 1) / block)), dim3(block), lds, st, tp->dev, J, fields_dev, out_dev);
    STARK_HIP(ctx, hipGetLastError());
    return STARK_OK;
}
// Long serial sponges: five waves per chain (poseidon_chain.hpp) unless the option "sponge_one_wave" asks for the round-2 one-wave form.
static int32_t launch_column_sponges(stark_ctx* ctx, stark_params* tp, const TrMultiJob& J, unsigned nblocks, fr_t* out_dev) {
    if (tp->dev.chain_a && !ctx->opt_sponge_one_wave) {
        const row::Consts RK = row_consts_of(ctx);
        hipLaunchKernelGGL(k_tr_hash_chain, dim3(nblocks), dim3(320), chain_lds_bytes(), ctx->stream, tp->dev, J, RK, out_dev);
    } else {
        hipLaunchKernelGGL(k_tr_hash_coop_multi, dim3(nblocks), dim3(64), coop_lds_bytes(17), ctx->stream, tp->dev, J, out_dev);
    }
    STARK_HIP(ctx, hipGetLastError());
    return STARK_OK;
}
// The four column sponges of DeepAliRealBuilder::build_f0 (fri.rs:551-554) as one launch of four blocks.
int32_t tr_hash_columns4_dev(stark_ctx* ctx, const char* const tags[4], const fr_t* const cols[4], size_t n0, fr_t* out4_dev) {
    stark_params* tp = nullptr; STARK_TRY(ctx_transcript_params(ctx, &tp));
    TrMultiJob J; J.cap = host::h_tag("FSv1-TRANSCRIPT-INIT"); J.batch = nullptr; J.stride = 0;
    for (int c = 0; c < 4; ++c) {
        fr_t* frame = nullptr; int np = 0, ns = 0; STARK_TRY(tr_frame(ctx, "FRI/FS", tags[c], "out", &frame, &np, &ns));
        J.prefix[c] = frame; J.np[c] = np; J.suffix[c] = frame + np; J.ns[c] = ns; J.fields[c] = cols[c]; J.k[c] = n0;
    }
    STARK_TRY(launch_column_sponges(ctx, tp, J, 4, out4_dev));
    return STARK_OK;
}
// The same for B independent traces: 4 * B chains, one block each, in one launch.  ptrs_dev[4 * p + c] = column c of trace p (device array of device pointers).
int32_t tr_hash_columns_batch_dev(stark_ctx* ctx, const char* const tags[4], const fr_t* const* ptrs_dev, size_t batch, size_t n0, fr_t* out_dev) {
    stark_params* tp = nullptr; STARK_TRY(ctx_transcript_params(ctx, &tp));
    TrMultiJob J; J.cap = host::h_tag("FSv1-TRANSCRIPT-INIT"); J.batch = ptrs_dev; J.stride = 0;
    for (int c = 0; c < 4; ++c) {
        fr_t* frame = nullptr; int np = 0, ns = 0; STARK_TRY(tr_frame(ctx, "FRI/FS", tags[c], "out", &frame, &np, &ns));
        J.prefix[c] = frame; J.np[c] = np; J.suffix[c] = frame + np; J.ns[c] = ns; J.fields[c] = nullptr; J.k[c] = n0;
    }
    STARK_TRY(launch_column_sponges(ctx, tp, J, (unsigned)(4 * batch), out_dev));
    return STARK_OK;
}
int32_t tr_hash_host1(stark_ctx* ctx, const char* tag, const std::vector<fr_t>& fields, fr_t* out) {
    DevBuf in, o; STARK_HIP(ctx, in.alloc(ctx, fields.size() * sizeof(fr_t))); STARK_HIP(ctx, o.alloc(ctx, sizeof(fr_t)));
    if (!fields.empty()) STARK_HIP(ctx, hipMemcpyAsync(in.p, fields.data(), fields.size() * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    STARK_TRY(tr_hash_dev(ctx, tag, in.fr(), fields.size(), 1, o.fr()));
    STARK_HIP(ctx, hipMemcpyAsync(out, o.p, sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}

}  // namespace stark

// leaf template of hash_leaf_pair (fri.rs:38-44)
static int32_t ctx_leaf_init(stark_ctx* ctx, fr_t** out) {
    if (!ctx->leaf_init) {
        const fr_t AB = host::h_tag("FSv1-ABSORB-BYTES"), CH = host::h_tag("FSv1-CHALLENGE");
        fr_t init[17]; for (auto& x : init) x = host::h_zero();
        init[0] = AB; init[1] = host::h_words("FRI/leaf/poseidon")[0]; init[2] = AB; init[3] = host::h_words("FRI/leaf")[0];
        /* lanes 4,5 = (f, s) */ init[6] = CH; init[7] = AB; init[8] = host::h_words("leaf")[0]; init[16] = host::h_tag("FSv1-TRANSCRIPT-INIT");
        // closed form of round 0 for the wave-pair kernel: K_i = sum_{j != 4,5} M[i][j] * (init_j + rc0_j)^5, then columns 4 and 5 of M
        stark_params* tp = nullptr; STARK_TRY(ctx_transcript_params(ctx, &tp));
        const host::PoseidonConsts& c = tp->ref;
        fr_t blob[17 + 51 + 40];                       // + columns 4, 5 of M as 34 x 9 words in radix 2^29 (306 words = 38.25 elements)
        for (auto& x : blob) x = host::h_zero();
        for (int j = 0; j < 17; ++j) blob[j] = init[j];
        fr_t x[17]; for (int j = 0; j < 17; ++j) x[j] = fr_pow5<PallasFr>(host::h_add(init[j], c.rc_full[j]));
        for (int i = 0; i < 17; ++i) {
            fr_t k = host::h_zero();
            for (int j = 0; j < 17; ++j) if (j != 4 && j != 5) k = host::h_add(k, host::h_mul(c.mds[(size_t)i * 17 + j], x[j]));
            blob[17 + i] = k; blob[34 + i] = c.mds[(size_t)i * 17 + 4]; blob[51 + i] = c.mds[(size_t)i * 17 + 5];
            uint32_t* m45 = reinterpret_cast<uint32_t*>(&blob[68]);
            const fr_t k20 = fr_from_u64<PallasFr>(1ull << FR29_SBOX_SHIFT);     // the S-box outputs x4, x5 arrive divided by 2^20 (fr_pow5_r29)
            fr29_const_from<PallasFr>(host::h_mul(c.mds[(size_t)i * 17 + 4], k20), m45 + 9 * i); fr29_const_from<PallasFr>(host::h_mul(c.mds[(size_t)i * 17 + 5], k20), m45 + 9 * (17 + i));
        }
        STARK_HIP(ctx, hipMalloc((void**)&ctx->leaf_init, sizeof(blob)));
        STARK_HIP(ctx, hipMemcpyAsync(ctx->leaf_init, blob, sizeof(blob), hipMemcpyHostToDevice, ctx->stream));
        STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *out = ctx->leaf_init; return STARK_OK;
}

extern "C" {

int32_t stark_version(void) { return 1; }

int32_t stark_ctx_create(int32_t device, void* stream, stark_ctx_t** out) {
    if (!out) return STARK_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return STARK_ERR_HIP;   // no device => no product path
    if (hipSetDevice(device) != hipSuccess) return STARK_ERR_HIP;
    stark_ctx* c = new stark_ctx(); c->device = device;
    if (stream == STARK_STREAM_PRIVATE) { if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return STARK_ERR_HIP; } c->own_stream = true; }
    else { c->stream = (hipStream_t)stream; c->own_stream = false; }         // NULL = the device's legacy default stream (ordered against torch's default stream and every blocking stream)
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { delete c; return STARK_ERR_HIP; }
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->num_cus = cus; }
    stark::ntt_set_attrs();
    // allow the full 160 KiB of LDS per workgroup for the kernels that stage through it
    (void)hipFuncSetAttribute((const void*)k_leaf_pair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)k_hash_ds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)k_permute_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)k_tr_hash, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)k_hash_stream, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)k_leaf_pair2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)k_hash_ds2<17>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)k_hash_ds2<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)k_tr_hash_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)k_hash_ds_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)k_leaf_pair_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    *out = c; return STARK_OK;
}
}  // extern "C"
// everything the context owns; runs when the context has been destroyed AND its last handle is gone
static void ctx_teardown(stark_ctx* ctx) {
    for (stark_ctx* a : ctx->aux) if (a) ctx_teardown(a);            // worker contexts hand out no handles of their own
    ctx->aux.clear();
    (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream);
    stark::comm_destroy(ctx);
    stark::ntt_plans_free(ctx);
    if (ctx->tparams) stark_poseidon_params_free(ctx->tparams);
    for (auto& kv : ctx->merkle_params) stark_poseidon_params_free(kv.second);
    for (auto& kv : ctx->tr_frames) (void)hipFree(kv.second);
    if (ctx->leaf_init) (void)hipFree(ctx->leaf_init);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    for (auto& o : ctx->omega_tabs) { (void)hipFree(o.lo); (void)hipFree(o.hi); }
    for (auto& kv : ctx->pool_free) for (void* q : kv.second) (void)hipFree(q);
    for (auto& kv : ctx->pool_live) (void)hipFree(kv.first);                 // stark_alloc blocks the caller never freed (handles are gone by now)
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0); if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}
namespace stark {
void ctx_ref(stark_ctx* c) { ++c->live_handles; }
void ctx_unref(stark_ctx* c) { if (--c->live_handles == 0 && c->destroy_pending) ctx_teardown(c); }
}
extern "C" {
// With handles still alive (trees, FRI states, plans, transcripts, parameter sets made for the caller) the context is only MARKED here: it stops
// accepting new work through stark_ctx_* but stays valid for those handles, and the last of them to be freed tears it down.
int32_t stark_ctx_destroy(stark_ctx_t* ctx) {
    if (!ctx) return STARK_ERR_INVALID_ARG;
    if (ctx->destroy_pending) return STARK_ERR_INVALID_ARG;                  // destroyed twice
    STARK_TRY(ctx_enter(ctx));
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->live_handles > 0) { ctx->destroy_pending = true; return STARK_OK; }
    ctx_teardown(ctx); return STARK_OK;
}
int32_t stark_ctx_sync(stark_ctx_t* ctx) { if (!ctx) return STARK_ERR_INVALID_ARG; STARK_TRY(ctx_enter(ctx)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK; }
int32_t stark_ctx_trim(stark_ctx_t* ctx) {
    if (!ctx) return STARK_ERR_INVALID_ARG; STARK_TRY(ctx_enter(ctx));
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& kv : ctx->pool_free) { for (void* q : kv.second) (void)hipFree(q); kv.second.clear(); }
    ctx->pool_cached_bytes = 0;
    for (stark_ctx* a : ctx->aux) if (a) (void)stark_ctx_trim(a);
    stark::ntt_plans_free(ctx);                  // NTT plans with their direct twiddle / coset tables (up to 3*n*32 B per plan) are rebuilt on demand
    if (ctx->scratch) { (void)hipFree(ctx->scratch); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
    return STARK_OK;
}
size_t stark_ctx_cached_bytes(stark_ctx_t* ctx) { return ctx ? ctx->pool_cached_bytes : 0; }
}  // extern "C"
namespace stark {
int32_t ctx_aux(stark_ctx* ctx, size_t k, stark_ctx** out) {
    while (ctx->aux.size() <= k) {
        stark_ctx* a = nullptr; int32_t rc = stark_ctx_create(ctx->device, STARK_STREAM_PRIVATE, &a);
        if (rc) return ctx->fail(rc, "worker context");
        ctx->aux.push_back(a);
    }
    stark_ctx* a = ctx->aux[k];
    a->opt_ntt_direct_max_log = ctx->opt_ntt_direct_max_log; a->opt_ntt_merged_coset = ctx->opt_ntt_merged_coset; a->opt_ntt_log_tile = ctx->opt_ntt_log_tile; a->opt_ntt_log_tile_forced = ctx->opt_ntt_log_tile_forced;
    a->opt_ntt_min_waves = ctx->opt_ntt_min_waves; a->opt_poseidon_lane_only = ctx->opt_poseidon_lane_only; a->opt_sponge_one_wave = ctx->opt_sponge_one_wave;
    *out = a; return STARK_OK;
}
}
extern "C" {
int32_t stark_ctx_set_option(stark_ctx_t* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    const std::string k(key);
    if (k == "ntt_direct_max_log") { if (value < 0 || value > 30) return ctx->fail(STARK_ERR_INVALID_ARG, "ntt_direct_max_log: 0..30"); ctx->opt_ntt_direct_max_log = (int)value; }
    else if (k == "ntt_log_tile") { if (value != -1 && (value < 8 || value > 12)) return ctx->fail(STARK_ERR_INVALID_ARG, "ntt_log_tile: 8..12, or -1 for the default"); ctx->opt_ntt_log_tile_forced = value != -1; ctx->opt_ntt_log_tile = value == -1 ? 11 : (int)value; }
    else if (k == "ntt_min_waves") { if (value != 2 && value != 4) return ctx->fail(STARK_ERR_INVALID_ARG, "ntt_min_waves: 2 or 4"); ctx->opt_ntt_min_waves = (int)value; }
    else if (k == "ntt_merged_coset") ctx->opt_ntt_merged_coset = value != 0;
    else if (k == "poseidon_lane_only") ctx->opt_poseidon_lane_only = value != 0;
    else if (k == "sponge_one_wave") ctx->opt_sponge_one_wave = value != 0;
    else if (k == "sponge_debug") ctx->opt_sponge_debug = (int)value;
    else return ctx->fail(STARK_ERR_INVALID_ARG, "unknown option '" + k + "' (ntt_direct_max_log, ntt_merged_coset, ntt_log_tile, ntt_min_waves, poseidon_lane_only, sponge_one_wave)");
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    stark::ntt_plans_free(ctx);                  // plans (and their direct tables) are rebuilt lazily under the new options
    return STARK_OK;
}
const char* stark_last_error(stark_ctx_t* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int32_t stark_malloc(stark_ctx_t* ctx, size_t bytes, void** dptr) { if (!ctx || !dptr) return STARK_ERR_INVALID_ARG; STARK_TRY(ctx_enter(ctx)); STARK_HIP(ctx, hipMalloc(dptr, bytes ? bytes : 32)); return STARK_OK; }
int32_t stark_free(stark_ctx_t* ctx, void* dptr) { if (!ctx) return STARK_ERR_INVALID_ARG; STARK_TRY(ctx_enter(ctx)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); STARK_HIP(ctx, hipFree(dptr)); return STARK_OK; }
int32_t stark_memcpy_h2d(stark_ctx_t* ctx, void* d, const void* s, size_t bytes) {
    if (!ctx) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    STARK_HIP(ctx, hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK; }
int32_t stark_memcpy_d2h(stark_ctx_t* ctx, void* d, const void* s, size_t bytes) {
    if (!ctx) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    STARK_HIP(ctx, hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK; }
int32_t stark_timer_start(stark_ctx_t* ctx) { if (!ctx) return STARK_ERR_INVALID_ARG; STARK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream)); return STARK_OK; }
int32_t stark_timer_stop_ms(stark_ctx_t* ctx, float* ms) {
    if (!ctx || !ms) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    STARK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream)); STARK_HIP(ctx, hipEventSynchronize(ctx->ev1)); STARK_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1)); return STARK_OK; }

// ---- diagnostics: the integer-VALU "speed of light" of this device, measured live -----------------------
}  // extern "C"
// 8 independent v_mad_u64_u32 chains per lane (the primitive of every field product here); 4 waves per SIMD.
#define STARK_DIAG_ITERS 2048
static __global__ void __launch_bounds__(1024) k_diag_mac_rate(uint32_t* out, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 9 + 4, a5 = a0 * 11 + 5, a6 = a0 * 13 + 6, a7 = a0 * 15 + 7;
    uint64_t d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7; const uint32_t b = seed | 1;
#define STARK_DIAG_MAD(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d##i) : "v"(a##i), "v"(b) : "vcc");
    for (int i = 0; i < STARK_DIAG_ITERS; ++i) { STARK_DIAG_MAD(0) STARK_DIAG_MAD(1) STARK_DIAG_MAD(2) STARK_DIAG_MAD(3) STARK_DIAG_MAD(4) STARK_DIAG_MAD(5) STARK_DIAG_MAD(6) STARK_DIAG_MAD(7) }
#undef STARK_DIAG_MAD
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7);
}
extern "C" {
int32_t stark_diag_mac_rate(stark_ctx_t* ctx, double* lane_macs_per_s) {
    if (!ctx || !lane_macs_per_s) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    hipDeviceProp_t prop; STARK_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    const int blocks = prop.multiProcessorCount * 2, threads = 1024;        // 2 x 16 waves per CU = 8 waves per SIMD resident, issue-bound either way
    DevBuf o; STARK_HIP(ctx, o.alloc(ctx, (size_t)blocks * threads * 4));
    hipLaunchKernelGGL(k_diag_mac_rate, dim3(blocks), dim3(threads), 0, ctx->stream, (uint32_t*)o.p, 12345u);       // warm-up (clocks, code object)
    STARK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(k_diag_mac_rate, dim3(blocks), dim3(threads), 0, ctx->stream, (uint32_t*)o.p, 12345u);
    STARK_HIP(ctx, hipGetLastError());
    STARK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream)); STARK_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0; STARK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *lane_macs_per_s = (double)blocks * threads * 8.0 * STARK_DIAG_ITERS / (ms * 1e-3);
    return STARK_OK;
}

// ---- constants ---------------------------------------------------------------------------------------
int32_t stark_poseidon_params_upload(stark_ctx_t* ctx, int32_t t, int32_t rf, int32_t rp, const uint64_t* mds, const uint64_t* rc_full, const uint64_t* rc_partial, stark_params_t** out) {
    if (!ctx || !mds || !rc_full || !rc_partial || !out || t < 2 || rf <= 0 || (rf & 1) || rp <= 0) return ctx ? ctx->fail(STARK_ERR_INVALID_ARG, "bad params") : STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    host::PoseidonConsts c; c.t = t; c.rf = rf; c.rp = rp;
    c.mds.resize((size_t)t * t); c.rc_full.resize((size_t)rf * t); c.rc_partial.resize(rp);
    for (size_t i = 0; i < c.mds.size(); ++i) c.mds[i] = load_fr(mds + 4 * i);
    for (size_t i = 0; i < c.rc_full.size(); ++i) c.rc_full[i] = load_fr(rc_full + 4 * i);
    for (size_t i = 0; i < c.rc_partial.size(); ++i) c.rc_partial[i] = load_fr(rc_partial + 4 * i);
    { int32_t rc = params_from_consts(ctx, c, out); if (rc == STARK_OK) (*out)->ref_.bind(ctx); return rc; }   // a set handed to the caller keeps the context alive
}
int32_t stark_poseidon_params_for_width(stark_ctx_t* ctx, int32_t t, stark_params_t** out) {
    if (!ctx || !out) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (host::rp_for_width(t) < 0) return ctx->fail(STARK_ERR_UNSUPPORTED, "unsupported Poseidon width t; supported t in {9,17,33,65,129}");   // poseidon/src/lib.rs:127
    { int32_t rc = params_from_consts(ctx, host::consts_for_width(t), out); if (rc == STARK_OK) (*out)->ref_.bind(ctx); return rc; }   // a set handed to the caller keeps the context alive
}
int32_t stark_poseidon_params_t17_seed(stark_ctx_t* ctx, const uint8_t* seed, size_t n, stark_params_t** out) {
    if (!ctx || !out || (!seed && n)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    { int32_t rc = params_from_consts(ctx, host::derive_consts(std::string((const char*)seed, n), 17, 8, 64), out); if (rc == STARK_OK) (*out)->ref_.bind(ctx); return rc; }   // a set handed to the caller keeps the context alive
}
int32_t stark_poseidon_params_export(stark_params_t* p, int32_t* t, int32_t* rf, int32_t* rp, uint64_t* mds, uint64_t* rc_full, uint64_t* rc_partial) {
    if (!p) return STARK_ERR_INVALID_ARG;
    if (t) *t = p->ref.t; if (rf) *rf = p->ref.rf; if (rp) *rp = p->ref.rp;
    if (mds) for (size_t i = 0; i < p->ref.mds.size(); ++i) store_fr(mds + 4 * i, p->ref.mds[i]);
    if (rc_full) for (size_t i = 0; i < p->ref.rc_full.size(); ++i) store_fr(rc_full + 4 * i, p->ref.rc_full[i]);
    if (rc_partial) for (size_t i = 0; i < p->ref.rc_partial.size(); ++i) store_fr(rc_partial + 4 * i, p->ref.rc_partial[i]);
    return STARK_OK;
}
int32_t stark_poseidon_params_free(stark_params_t* p) { if (!p) return STARK_ERR_INVALID_ARG; if (p->blob) (void)hipFree(p->blob); delete p; return STARK_OK; }

// ---- Poseidon ----------------------------------------------------------------------------------------
int32_t stark_poseidon_permute_batch_dev(stark_ctx_t* ctx, stark_params_t* p, uint64_t* states, size_t n) {
    if (!ctx || !p || (!states && n)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (!n) return STARK_OK;
    const int block = poseidon_block(p->dev.t);
    hipLaunchKernelGGL(k_permute_batch, dim3((unsigned)((n + block - 1) / block)), dim3(block), poseidon_lds(p->dev.t, block), ctx->stream, p->dev, as_fr(states), n);
    STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
}
int32_t stark_poseidon_permute_batch(stark_ctx_t* ctx, stark_params_t* p, uint64_t* states, size_t n) {
    if (!ctx || !p || (!states && n)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    size_t bytes = n * p->dev.t * sizeof(fr_t); DevBuf d; STARK_HIP(ctx, d.alloc(ctx, bytes));
    STARK_HIP(ctx, hipMemcpyAsync(d.p, states, bytes, hipMemcpyHostToDevice, ctx->stream));
    STARK_TRY(stark_poseidon_permute_batch_dev(ctx, p, (uint64_t*)d.p, n));
    STARK_HIP(ctx, hipMemcpyAsync(states, d.p, bytes, hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}
static int32_t hash_stream(stark_ctx_t* ctx, stark_params_t* p, int mode, const uint64_t* a, size_t na, const uint64_t* b, size_t nb, const fr_t& tag, size_t n, uint64_t* out) {
    DevBuf da, db, dout; STARK_HIP(ctx, da.alloc(ctx, n * na * sizeof(fr_t))); STARK_HIP(ctx, db.alloc(ctx, n * nb * sizeof(fr_t))); STARK_HIP(ctx, dout.alloc(ctx, n * sizeof(fr_t)));
    if (n * na) STARK_HIP(ctx, hipMemcpyAsync(da.p, a, n * na * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    if (n * nb) STARK_HIP(ctx, hipMemcpyAsync(db.p, b, n * nb * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    const int block = poseidon_block(p->dev.t);
    hipLaunchKernelGGL(k_hash_stream, dim3((unsigned)((n + block - 1) / block)), dim3(block), poseidon_lds(p->dev.t, block), ctx->stream, p->dev, mode, da.fr(), na, db.fr(), nb, tag, n, dout.fr());
    STARK_HIP(ctx, hipGetLastError());
    STARK_HIP(ctx, hipMemcpyAsync(out, dout.p, n * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}
int32_t stark_poseidon_hash_with_ds_dynamic(stark_ctx_t* ctx, stark_params_t* p, const uint64_t* ds, size_t nds, const uint64_t* in, size_t cnt, size_t n, uint64_t* out) {
    if (!ctx || !p || !out || (!ds && nds) || (!in && cnt)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (!n) return STARK_OK;
    return hash_stream(ctx, p, 0, ds, nds, in, cnt, host::h_zero(), n, out);
}
int32_t stark_poseidon_hash_with_ds(stark_ctx_t* ctx, stark_params_t* p, const uint64_t* in, size_t cnt, const uint64_t* ds_tag, uint64_t* out) {
    if (!ctx || !p || !out || !ds_tag || (!in && cnt)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (p->dev.t != 17) return ctx->fail(STARK_ERR_INVALID_ARG, "hash_with_ds is the fixed t=17 sponge");
    return hash_stream(ctx, p, 1, nullptr, 0, in, cnt, load_fr(ds_tag), 1, out);
}
static int32_t launch_hash_ds(stark_ctx_t* ctx, hipStream_t st, stark_params_t* p, int mode, size_t arity, uint32_t level, uint64_t pos0, uint64_t label,
                              const fr_t* in0, const fr_t* in1, size_t n_in, fr_t* out, size_t cp_div = 1, const uint64_t* pos_list = nullptr, size_t chunk = 0) {
    DsJob J; J.pos_list = pos_list; J.arity_f = host::h_u64(arity); J.level_f = host::h_u64(level); J.label_f = host::h_u64(label); J.pos0 = pos0; J.arity = arity; J.n_in = n_in; J.mode = mode; J.cp_div = cp_div ? cp_div : 1;
    if (chunk) J.arity = chunk;          // the verifier's groups: DS field `arity` as given, `chunk` children per hash (a short last chunk of the proof's level)
    J.n_out = mode == 1 ? n_in : (n_in + J.arity - 1) / J.arity;
    if (!J.n_out) return STARK_OK;
    if (use_chain(ctx, p->dev, J.n_out, kChainMaxNodes)) {
        hipLaunchKernelGGL(k_hash_ds_chain, dim3((unsigned)J.n_out), dim3(320), chain_lds_bytes(), st, p->dev, J, row_consts_of(ctx), in0, in1, out);
        STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
    }
    if (use_pair(ctx, p->dev.t) && J.n_out <= kCoopMaxNodes) {
        // small level: one wave per node (latency form); a batch of 64 nodes per wave pair only pays off above that
        if (p->dev.t == 17) hipLaunchKernelGGL(k_hash_ds_coop<17>, dim3((unsigned)J.n_out), dim3(64), coop_lds_bytes(17), st, p->dev, J, in0, in1, out);
        else hipLaunchKernelGGL(k_hash_ds_coop<9>, dim3((unsigned)J.n_out), dim3(64), coop_lds_bytes(9), st, p->dev, J, in0, in1, out);
        STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
    }
    if (use_pair(ctx, p->dev.t)) {
        if (p->dev.t == 17) hipLaunchKernelGGL(k_hash_ds2<17>, dim3((unsigned)((J.n_out + 63) / 64)), dim3(128), pair_lds_bytes(17), st, p->dev, J, in0, in1, out);
        else hipLaunchKernelGGL(k_hash_ds2<9>, dim3((unsigned)((J.n_out + 63) / 64)), dim3(128), pair_lds_bytes(9), st, p->dev, J, in0, in1, out);
        STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
    }
    if (!ctx->opt_poseidon_lane_only && (p->dev.t == 33 || p->dev.t == 65 || p->dev.t == 129) && J.n_out <= 0x7fffffffu) {
        // wide states (arity 32 / 64 / 128): one wave per node (poseidon_wave.hpp) — a lane per node walks a 17 ms permutation alone
        const dim3 grid((unsigned)J.n_out), blk(64); const size_t lds = wave_lds_bytes(p->dev.t);
        if (p->dev.t == 33) hipLaunchKernelGGL(k_hash_ds_wave<33>, grid, blk, lds, st, p->dev, J, in0, in1, out);
        else if (p->dev.t == 65) hipLaunchKernelGGL(k_hash_ds_wave<65>, grid, blk, lds, st, p->dev, J, in0, in1, out);
        else hipLaunchKernelGGL(k_hash_ds_wave<129>, grid, blk, lds, st, p->dev, J, in0, in1, out);
        STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
    }
    const int block = poseidon_block(p->dev.t);
    hipLaunchKernelGGL(k_hash_ds, dim3((unsigned)((J.n_out + block - 1) / block)), dim3(block), poseidon_lds(p->dev.t, block), st, p->dev, J, in0, in1, out);
    STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
}
}  // extern "C"
// DS hashes with scattered positions (the verifier's union-of-paths levels): hash k = H([arity, level, positions[k], label] || chunk children)
int32_t stark::hash_ds_scattered(stark_ctx* ctx, stark_params* p, int mode, size_t arity, size_t chunk, uint32_t level, uint64_t label, const uint64_t* positions_dev,
                                 const fr_t* in0, const fr_t* in1, size_t n_hashes, fr_t* out) {
    return launch_hash_ds(ctx, ctx->stream, p, mode, arity, level, 0, label, in0, in1, mode == 1 ? n_hashes : n_hashes * chunk, out, 1, positions_dev, mode == 1 ? 0 : chunk);
}
extern "C" {
int32_t stark_poseidon_hash_ds_batch_dev(stark_ctx_t* ctx, stark_params_t* p, size_t arity, uint32_t level, uint64_t pos0, uint64_t label, const uint64_t* in, size_t n_in, uint64_t* out) {
    if (!ctx || !p || !in || !out || arity == 0) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (host::width_for_arity(arity) != p->dev.t) return ctx->fail(STARK_ERR_INVALID_ARG, "arity incompatible with Poseidon width");
    STARK_TRY(ctx_enter(ctx));
    return launch_hash_ds(ctx, ctx->stream, p, 0, arity, level, pos0, label, as_fr(in), nullptr, n_in, as_fr(out));
}
int32_t stark_poseidon_hash_ds_batch(stark_ctx_t* ctx, stark_params_t* p, size_t arity, uint32_t level, uint64_t pos0, uint64_t label, const uint64_t* in, size_t n_in, uint64_t* out) {
    if (!ctx || !p || !in || !out || arity == 0) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    size_t n_out = (n_in + arity - 1) / arity; DevBuf di, dout; STARK_HIP(ctx, di.alloc(ctx, n_in * sizeof(fr_t))); STARK_HIP(ctx, dout.alloc(ctx, n_out * sizeof(fr_t)));
    STARK_HIP(ctx, hipMemcpyAsync(di.p, in, n_in * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    STARK_TRY(stark_poseidon_hash_ds_batch_dev(ctx, p, arity, level, pos0, label, (const uint64_t*)di.p, n_in, (uint64_t*)dout.p));
    STARK_HIP(ctx, hipMemcpyAsync(out, dout.p, n_out * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}
}  // extern "C"
namespace stark {
// hash_leaf_pair is the FIXED transcript permutation (fri.rs:39: transcript::default_params()): a caller's handle must hold
// those very constants, anything else would silently mix two parameter sets (round 0 is folded into the context's template).
static bool same_consts(const host::PoseidonConsts& a, const host::PoseidonConsts& b) {
    if (a.t != b.t || a.rf != b.rf || a.rp != b.rp || a.mds.size() != b.mds.size() || a.rc_full.size() != b.rc_full.size() || a.rc_partial.size() != b.rc_partial.size()) return false;
    for (size_t i = 0; i < a.mds.size(); ++i) if (!fr_eq(a.mds[i], b.mds[i])) return false;
    for (size_t i = 0; i < a.rc_full.size(); ++i) if (!fr_eq(a.rc_full[i], b.rc_full[i])) return false;
    for (size_t i = 0; i < a.rc_partial.size(); ++i) if (!fr_eq(a.rc_partial[i], b.rc_partial[i])) return false;
    return true;
}
int32_t leaf_pair_hash_on(stark_ctx* ctx, hipStream_t st, const fr_t* f, const fr_t* f_next, size_t n, size_t m, fr_t* h) {
    if (!n) return STARK_OK;
    stark_params* tp = nullptr; STARK_TRY(ctx_transcript_params(ctx, &tp));
    fr_t* init = nullptr; STARK_TRY(ctx_leaf_init(ctx, &init));
    if (use_chain(ctx, tp->dev, n, kChainMaxLeaves)) {
        hipLaunchKernelGGL(k_leaf_pair_chain, dim3((unsigned)n), dim3(320), chain_lds_bytes(), st, tp->dev, row_consts_of(ctx), (const fr_t*)init, f, f_next, m, h);
        STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
    }
    if (use_pair(ctx, 17) && !ctx->opt_sponge_one_wave && n <= kCoopMaxLeaves) {
        hipLaunchKernelGGL(k_leaf_pair_coop, dim3((unsigned)n), dim3(64), coop_lds_bytes(17), st, tp->dev, (const fr_t*)init, f, f_next, m, h);
        STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
    }
    if (use_pair(ctx, 17)) {
        hipLaunchKernelGGL(k_leaf_pair2, dim3((unsigned)((n + 63) / 64)), dim3(128), pair_lds_bytes(17), st, tp->dev, init + 17, f, f_next, n, m, h);
        STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
    }
    const int block = 64;
    hipLaunchKernelGGL(k_leaf_pair, dim3((unsigned)((n + block - 1) / block)), dim3(block), poseidon_lds(17, block), st, tp->dev, init, f, f_next, n, m, h);
    STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
}
}  // namespace stark
extern "C" {
int32_t stark_leaf_pair_hash_dev(stark_ctx_t* ctx, stark_params_t* tp, const uint64_t* f, const uint64_t* f_next, size_t n, size_t m, uint64_t* h) {
    if (!ctx || (!f && n) || (!h && n) || m == 0) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (tp) {    // NULL = the transcript parameters (the only valid choice); a handle is accepted when it holds the same constants
        if (tp->dev.t != 17) return ctx->fail(STARK_ERR_INVALID_ARG, "leaf hash uses the t=17 transcript permutation");
        stark_params* mine = nullptr; STARK_TRY(ctx_transcript_params(ctx, &mine));
        if (tp != mine && !same_consts(tp->ref, mine->ref)) return ctx->fail(STARK_ERR_INVALID_ARG, "hash_leaf_pair is defined over transcript::default_params(); the handle holds other constants");
    }
    return leaf_pair_hash_on(ctx, ctx->stream, as_fr(f), as_fr(f_next), n, m, as_fr(h));
}
int32_t stark_leaf_pair_hash(stark_ctx_t* ctx, stark_params_t* tp, const uint64_t* f, const uint64_t* f_next, size_t n, size_t m, uint64_t* h) {
    if (!ctx || (!f && n) || (!h && n) || m == 0) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    size_t nn = f_next ? (n + m - 1) / m : 0; DevBuf df, dn, dh;
    STARK_HIP(ctx, df.alloc(ctx, n * sizeof(fr_t))); STARK_HIP(ctx, dn.alloc(ctx, nn * sizeof(fr_t))); STARK_HIP(ctx, dh.alloc(ctx, n * sizeof(fr_t)));
    if (n) STARK_HIP(ctx, hipMemcpyAsync(df.p, f, n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    if (nn) STARK_HIP(ctx, hipMemcpyAsync(dn.p, f_next, nn * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    STARK_TRY(stark_leaf_pair_hash_dev(ctx, tp, (const uint64_t*)df.p, f_next ? (const uint64_t*)dn.p : nullptr, n, m, (uint64_t*)dh.p));
    if (n) STARK_HIP(ctx, hipMemcpyAsync(h, dh.p, n * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}
int32_t stark_tr_hash_fields_tagged_dev(stark_ctx_t* ctx, stark_params_t* tp, const char* tag, const uint64_t* fields, size_t k, size_t n, uint64_t* out) {
    if (!ctx || !tag || (!fields && k && n) || (!out && n)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    (void)tp;   // the transcript permutation is fixed (transcript/src/lib.rs:44-46); the handle is accepted for API symmetry
    return tr_hash_dev(ctx, tag, as_fr(fields), k, n, as_fr(out));
}
int32_t stark_tr_hash_fields_tagged(stark_ctx_t* ctx, stark_params_t* tp, const char* tag, const uint64_t* fields, size_t k, size_t n, uint64_t* out) {
    if (!ctx || !tag || (!fields && k && n) || (!out && n)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    DevBuf di, dout; STARK_HIP(ctx, di.alloc(ctx, n * k * sizeof(fr_t))); STARK_HIP(ctx, dout.alloc(ctx, n * sizeof(fr_t)));
    if (n * k) STARK_HIP(ctx, hipMemcpyAsync(di.p, fields, n * k * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    STARK_TRY(stark_tr_hash_fields_tagged_dev(ctx, tp, tag, (const uint64_t*)di.p, k, n, (uint64_t*)dout.p));
    if (n) STARK_HIP(ctx, hipMemcpyAsync(out, dout.p, n * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}

// ---- Merkle ------------------------------------------------------------------------------------------
}  // extern "C"
namespace stark {
// MerkleTree::new / new_pairs on `st`.  pairs: leaves are (f_i, cp[i / cp_div]) pairs (cp == nullptr: zeros).  adopt: `leaves` is a
// pooled block (ctx_alloc) whose ownership moves into the tree as level 0 (no copy); otherwise level 0 is a copy.
int32_t merkle_build_on(stark_ctx* ctx, hipStream_t st, stark_params* p, size_t arity, uint64_t label, const fr_t* leaves, size_t n, int pairs, const fr_t* cp, size_t cp_div,
                        uint64_t first_pos, uint32_t level0, size_t stop_at_len, bool adopt, stark_tree** out) {
    if (n == 0) return ctx->fail(STARK_ERR_INVALID_ARG, "no leaves");                                                 // merkle/src/lib.rs:148
    if (host::width_for_arity(arity) != p->dev.t) return ctx->fail(STARK_ERR_INVALID_ARG, "arity incompatible with Poseidon width");   // :155-161
    if (arity == 1 && n > 1) return ctx->fail(STARK_ERR_UNSUPPORTED, "arity 1 with more than one leaf never terminates in the reference");
    stark_tree* T = new stark_tree(); T->ref_.bind(ctx); T->ctx = ctx; T->p = p; T->arity = arity; T->label = label;
    auto bail = [&](int32_t rc) { delete T; return rc; };
    fr_t* l0 = nullptr;
    if (adopt && !pairs) l0 = const_cast<fr_t*>(leaves);
    else { void* q = nullptr; int32_t rc = ctx_alloc(ctx, n * sizeof(fr_t), &q); if (rc) return bail(rc); l0 = (fr_t*)q; }
    T->levels.push_back(l0); T->lens.push_back(n); T->owned.push_back(1);
    if (pairs) { int32_t rc = launch_hash_ds(ctx, st, p, 1, arity, 0xFFFFFFFFu, first_pos, label, leaves, cp, n, l0, cp_div); if (rc) return bail(rc); }
    else if (!adopt && hipMemcpyAsync(l0, leaves, n * sizeof(fr_t), hipMemcpyDeviceToDevice, st) != hipSuccess) return bail(ctx->fail(STARK_ERR_HIP, "copy leaves"));
    uint32_t level = level0; uint64_t pos = first_pos; size_t stop = stop_at_len > 0 ? stop_at_len : 1;
    while (T->lens.back() > stop) {
        size_t len = T->lens.back(), nn = (len + arity - 1) / arity;
        if (pos % arity) return bail(ctx->fail(STARK_ERR_INVALID_ARG, "shard offset not aligned to the arity"));
        pos /= arity;
        void* nx = nullptr; { int32_t rc = ctx_alloc(ctx, nn * sizeof(fr_t), &nx); if (rc) return bail(rc); }
        T->levels.push_back((fr_t*)nx); T->lens.push_back(nn); T->owned.push_back(1);
        int32_t rc = launch_hash_ds(ctx, st, p, 0, arity, level, pos, label, T->levels[T->levels.size() - 2], nullptr, len, (fr_t*)nx); if (rc) return bail(rc);
        level += 1;
    }
    *out = T; return STARK_OK;
}
}  // namespace stark
extern "C" {
int32_t stark_merkle_build_dev(stark_ctx_t* ctx, stark_params_t* p, size_t arity, uint64_t label, const uint64_t* leaves, size_t n, int32_t pairs, const uint64_t* cp,
                               uint64_t first_pos, uint32_t level0, int32_t stop_at_len, stark_tree_t** out) {
    if (!ctx || !p || !leaves || !out || arity == 0 || (pairs && !cp)) return ctx ? ctx->fail(STARK_ERR_INVALID_ARG, "bad merkle args") : STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    return merkle_build_on(ctx, ctx->stream, p, arity, label, as_fr(leaves), n, pairs, as_fr(cp), 1, first_pos, level0, stop_at_len > 0 ? (size_t)stop_at_len : 0, false, out);
}
int32_t stark_merkle_build(stark_ctx_t* ctx, stark_params_t* p, size_t arity, uint64_t label, const uint64_t* leaves, size_t n, int32_t pairs, const uint64_t* cp, stark_tree_t** out) {
    if (!ctx || !p || !leaves || !out || (pairs && !cp)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (n == 0) return ctx->fail(STARK_ERR_INVALID_ARG, "no leaves");
    DevBuf dl, dc; STARK_HIP(ctx, dl.alloc(ctx, n * sizeof(fr_t))); STARK_HIP(ctx, hipMemcpyAsync(dl.p, leaves, n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    if (pairs) { STARK_HIP(ctx, dc.alloc(ctx, n * sizeof(fr_t))); STARK_HIP(ctx, hipMemcpyAsync(dc.p, cp, n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream)); }
    STARK_TRY(stark_merkle_build_dev(ctx, p, arity, label, (const uint64_t*)dl.p, n, pairs, pairs ? (const uint64_t*)dc.p : nullptr, 0, 0, 0, out));
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}
int32_t stark_merkle_num_levels(stark_tree_t* t) { return t ? (int32_t)t->levels.size() : STARK_ERR_INVALID_ARG; }
size_t stark_merkle_level_len(stark_tree_t* t, int32_t lvl) { return (t && lvl >= 0 && (size_t)lvl < t->lens.size()) ? t->lens[lvl] : 0; }
const uint64_t* stark_merkle_level_dev(stark_tree_t* t, int32_t lvl) { return (t && lvl >= 0 && (size_t)lvl < t->levels.size()) ? (const uint64_t*)t->levels[lvl] : nullptr; }
int32_t stark_merkle_level(stark_tree_t* t, int32_t lvl, uint64_t* out) {
    if (!t || !out || lvl < 0 || (size_t)lvl >= t->levels.size()) return STARK_ERR_INVALID_ARG;
    stark_ctx* ctx = t->ctx; STARK_TRY(ctx_enter(ctx));
    STARK_HIP(ctx, hipMemcpyAsync(out, t->levels[lvl], t->lens[lvl] * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}
int32_t stark_merkle_root(stark_tree_t* t, uint64_t* out4) {
    if (!t || !out4) return STARK_ERR_INVALID_ARG;
    if (t->lens.back() != 1) return t->ctx->fail(STARK_ERR_INVALID_ARG, "partial (sharded) tree has no root");
    return stark_merkle_level(t, (int32_t)t->levels.size() - 1, out4);
}
int32_t stark_merkle_gather(stark_tree_t* t, int32_t lvl, const size_t* idx, size_t k, uint64_t* out) {
    if (!t || (!idx && k) || (!out && k) || lvl < 0 || (size_t)lvl >= t->levels.size()) return STARK_ERR_INVALID_ARG;
    stark_ctx* ctx = t->ctx; if (!k) return STARK_OK;
    STARK_TRY(ctx_enter(ctx));
    for (size_t i = 0; i < k; ++i) if (idx[i] >= t->lens[lvl]) return ctx->fail(STARK_ERR_INVALID_ARG, "gather index out of range");
    DevBuf di, dout; STARK_HIP(ctx, di.alloc(ctx, k * 8)); STARK_HIP(ctx, dout.alloc(ctx, k * sizeof(fr_t)));
    std::vector<uint64_t> ix(idx, idx + k);
    STARK_HIP(ctx, hipMemcpyAsync(di.p, ix.data(), k * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_gather, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, ctx->stream, t->levels[lvl], (const uint64_t*)di.p, (uint64_t)k, dout.fr());
    STARK_HIP(ctx, hipGetLastError());
    STARK_HIP(ctx, hipMemcpyAsync(out, dout.p, k * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}
int32_t stark_merkle_free(stark_tree_t* t) { if (!t) return STARK_ERR_INVALID_ARG; delete t; return STARK_OK; }   // levels go back to the context's pool (stream-ordered reuse: no device sync)

}  // extern "C"

// open_union_of_paths (merkle/src/lib.rs:246-315): host index logic (fri_plan.hpp) + device gathers of the siblings.
namespace stark {
struct TreeSource : FriSource {
    stark_tree* t; explicit TreeSource(stark_tree* t_) : t(t_) {}
    int32_t layer(size_t, const std::vector<size_t>&, std::vector<fr_t>&) override { return STARK_ERR_INVALID_ARG; }
    int32_t digests(size_t, size_t level, const std::vector<size_t>& idx, std::vector<fr_t>& out) override {
        out.resize(idx.size()); return stark_merkle_gather(t, (int32_t)level, idx.data(), idx.size(), (uint64_t*)out.data());
    }
};
int32_t merkle_open_host(stark_tree* t, const std::vector<size_t>& indices, MerkleProofHost& pr) {
    stark_ctx* ctx = t->ctx;
    if (indices.empty()) return ctx->fail(STARK_ERR_INVALID_ARG, "open_many: empty indices");                           // :247
    if (t->lens.back() != 1) return ctx->fail(STARK_ERR_INVALID_ARG, "cannot open a partial tree");
    for (size_t i : indices) if (i >= t->lens[0]) return ctx->fail(STARK_ERR_INVALID_ARG, "leaf index out of range");
    TreeSource src(t);
    return merkle_open_from(src, 0, t->lens, t->arity, indices, pr);
}
}  // namespace stark

extern "C" int32_t stark_merkle_open(stark_tree_t* t, const size_t* idx, size_t k, uint8_t* buf, size_t cap, size_t* len) {
    if (!t || !len || (!idx && k)) return STARK_ERR_INVALID_ARG;
    MerkleProofHost pr; STARK_TRY(merkle_open_host(t, std::vector<size_t>(idx, idx + k), pr));
    std::vector<uint8_t> b; enc_mproof(b, pr);
    *len = b.size();
    if (buf) { if (cap < b.size()) return t->ctx->fail(STARK_ERR_INVALID_ARG, "buffer too small"); memcpy(buf, b.data(), b.size()); }
    return STARK_OK;
}

#include "sumcheck_impl.hpp"

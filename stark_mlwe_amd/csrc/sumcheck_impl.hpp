// stark_mlwe_amd/csrc/sumcheck_impl.hpp — (included at the end of capi_core.hip: the Poseidon kernels it launches are defined once, in that translation unit)
// "next" row N4: the sum-check consumer of the Merkle / Poseidon / transcript kernels
// (crates/channel/src/lib.rs).  prove_plain / verify_plain (:1045-1128) and the Merkle-folded prove_mf / verify_mf (:1130-1240):
//   * MerkleCommitment::commit (commitment/src/lib.rs:85-90: arity 16, parameters "POSEIDON-T17-X5-SEED") of the witness and of every
//     shrinking folded layer = the level-batched Merkle kernels of the FRI path (merkle_build_on);
//   * round coefficients c0 = sum a_j, c1 = sum (b_j - a_j) (:406-416) and the fold (1-r) a + r b (:456-462) = streaming kernels;
//   * the Fiat-Shamir channel (:7-117) = a DEVICE-RESIDENT transcript: absorbs are queued on the host and run as ONE launch of the
//     one-wave sponge (poseidon_coop.hpp) when a challenge is drawn — the only host round trips are the challenges themselves,
//     which the protocol serialises anyway (r_i depends on the previous root, transcript/src/lib.rs:92-101).
// Proof bytes = bincode 1.x layout of the reference's serde structs ProofPlain / ProofMF (:925-979), what its bench measures.
#pragma once
#include <set>
#include "fri_verify.hpp"

using namespace stark;

struct stark_proof { std::vector<uint8_t> bytes; size_t size_estimate = 0; double ms[3] = {0, 0, 0}; };   // same object as capi_fri.hip's

namespace {

// ---- kernels ------------------------------------------------------------------------------------------------------------
// block partials of (c0, c1) over pairs (a, b) = (layer[2j], layer[2j+1]); out[2*block], out[2*block+1]
__global__ void __launch_bounds__(256) k_sc_coeffs(const fr_t* __restrict__ layer, uint64_t npairs, fr_t* __restrict__ out) {
    __shared__ uint4 red[2 * 2 * 4];
    fr_t c0 = fr_zero<PF>(), c1 = fr_zero<PF>();
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < npairs; j += (uint64_t)gridDim.x * blockDim.x) {
        const fr_t a = ldg(layer + 2 * j), b = ldg(layer + 2 * j + 1);
        c0 = fr_add<PF>(c0, a); c1 = fr_add<PF>(c1, fr_sub<PF>(b, a));
    }
    for (int sft = 1; sft < 64; sft <<= 1) { c0 = fr_add<PF>(c0, shfl_xor_fr(c0, sft)); c1 = fr_add<PF>(c1, shfl_xor_fr(c1, sft)); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[4 * wave] = make_uint4(c0.v[0], c0.v[1], c0.v[2], c0.v[3]); red[4 * wave + 1] = make_uint4(c0.v[4], c0.v[5], c0.v[6], c0.v[7]);
                     red[4 * wave + 2] = make_uint4(c1.v[0], c1.v[1], c1.v[2], c1.v[3]); red[4 * wave + 3] = make_uint4(c1.v[4], c1.v[5], c1.v[6], c1.v[7]); }
    __syncthreads();
    if (threadIdx.x == 0) {
        fr_t t0 = fr_zero<PF>(), t1 = fr_zero<PF>();
        for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) {
            fr_t x, y; const uint4 a = red[4 * wv], b = red[4 * wv + 1], c = red[4 * wv + 2], d = red[4 * wv + 3];
            x.v[0] = a.x; x.v[1] = a.y; x.v[2] = a.z; x.v[3] = a.w; x.v[4] = b.x; x.v[5] = b.y; x.v[6] = b.z; x.v[7] = b.w;
            y.v[0] = c.x; y.v[1] = c.y; y.v[2] = c.z; y.v[3] = c.w; y.v[4] = d.x; y.v[5] = d.y; y.v[6] = d.z; y.v[7] = d.w;
            t0 = fr_add<PF>(t0, x); t1 = fr_add<PF>(t1, y);
        }
        stg(out + 2 * blockIdx.x, t0); stg(out + 2 * blockIdx.x + 1, t1);
    }
}
// final reduction of the block partials (one block): out[0] = sum partial c0, out[1] = sum partial c1
__global__ void __launch_bounds__(64) k_sc_coeffs_final(const fr_t* __restrict__ part, uint64_t nblocks, fr_t* __restrict__ out) {
    fr_t c0 = fr_zero<PF>(), c1 = fr_zero<PF>();
    for (uint64_t i = threadIdx.x; i < nblocks; i += 64) { c0 = fr_add<PF>(c0, ldg(part + 2 * i)); c1 = fr_add<PF>(c1, ldg(part + 2 * i + 1)); }
    for (int sft = 1; sft < 64; sft <<= 1) { c0 = fr_add<PF>(c0, shfl_xor_fr(c0, sft)); c1 = fr_add<PF>(c1, shfl_xor_fr(c1, sft)); }
    if (threadIdx.x == 0) { stg(out, c0); stg(out + 1, c1); }
}
// next[j] = (1 - r) * layer[2j] + r * layer[2j+1]  =  a + r * (b - a)
__global__ void __launch_bounds__(256) k_sc_fold(const fr_t* __restrict__ layer, uint64_t npairs, fr_t r, fr_t* __restrict__ next) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= npairs) return;
    const fr_t a = ldg(layer + 2 * j), b = ldg(layer + 2 * j + 1);
    stg(next + j, fr_add<PF>(a, fr_mul<PF>(r, fr_sub<PF>(b, a))));
}
// The streaming transcript (transcript/src/lib.rs:79-101) on one wave: state[17] and the rate cursor live in device memory between
// launches; absorbs `n` queued fields with the lazy permute-on-full rule, then (finish) permutes and squeezes state[0].
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) k_tr_stream(PoseidonDev P, fr_t* __restrict__ state, uint32_t* __restrict__ pos_io,
                                                                                           const fr_t* __restrict__ fields, uint64_t n, int finish, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    CoopLds L = coop_setup<17>(lds, P);
    const int lane = threadIdx.x;
    fr_t s = lane < 17 ? ldg(state + lane) : fr_zero<PF>();
    uint32_t pos = *pos_io;
    for (uint64_t i = 0; i < n;) {
        if (pos == 16) { s = coop_permute<17>(s, P, L, lane); pos = 0; }                 // only before absorbing more (lazy)
        const uint64_t take = (16 - pos) < (n - i) ? (16 - pos) : (n - i);
        if ((uint32_t)lane >= pos && (uint64_t)lane < pos + take) s = fr_add<PF>(s, ldg(fields + i + (lane - pos)));
        pos += (uint32_t)take; i += take;
    }
    if (finish) { s = coop_permute<17>(s, P, L, lane); pos = 0; }
    if (lane < 17) stg(state + lane, s);
    if (lane == 0) { *pos_io = pos; if (out) stg(out, s); }
}

// The same on the five waves of poseidon_chain.hpp (72 us per permutation instead of 142 us): the stored cursor becomes `pos` leading no-op elements of
// the stream, so that the block boundaries — and with them the lazy permutations — fall where k_tr_stream puts them.
__global__ void __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(1, 2))) k_tr_stream_chain(PoseidonDev P, row::Consts RK, fr_t* __restrict__ state, uint32_t* __restrict__ pos_io,
                                                                                                  const fr_t* __restrict__ fields, uint64_t n, int finish, fr_t* __restrict__ out) {
    extern __shared__ uint4 lds[];
    const uint32_t pos = *pos_io;                                                           // every thread reads it before thread 0 writes it back (barriers in between)
    const size_t total = (size_t)pos + n;
    chain_sponge_ex(P, RK, lds, total, fr_zero<PF>(), [&](size_t q) -> fr_t { return q < pos ? fr_zero<PF>() : ldg(fields + (q - pos)); },
                    finish ? out : (fr_t*)nullptr, state, finish != 0, state);
    if (threadIdx.x == 0) *pos_io = finish ? 0u : (total ? (uint32_t)(total - 16 * ((total - 1) / 16)) : 0u);
}

// ---- device-resident transcript -----------------------------------------------------------------------------------------
struct DevTranscript {
    stark_ctx* ctx; DevBuf state, posb, out; std::vector<fr_t> pending; stark_params* tp = nullptr;
    explicit DevTranscript(stark_ctx* c) : ctx(c) {}
    int32_t init(const uint8_t* label, size_t n) {                                        // Transcript::new (:55-65)
        STARK_TRY(ctx_transcript_params(ctx, &tp));
        STARK_HIP(ctx, state.alloc(ctx, 17 * sizeof(fr_t))); STARK_HIP(ctx, posb.alloc(ctx, 4)); STARK_HIP(ctx, out.alloc(ctx, sizeof(fr_t)));
        fr_t st[17]; for (auto& x : st) x = host::h_zero(); st[16] = host::h_tag("FSv1-TRANSCRIPT-INIT");
        STARK_HIP(ctx, hipMemcpyAsync(state.p, st, sizeof(st), hipMemcpyHostToDevice, ctx->stream));
        STARK_HIP(ctx, hipMemsetAsync(posb.p, 0, 4, ctx->stream));
        STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));                                 // st is a stack temporary
        absorb_bytes(label, n); return STARK_OK;
    }
    void absorb_field(const fr_t& x) { pending.push_back(x); }
    void absorb_bytes(const uint8_t* b, size_t n) {                                        // :67-73: marker, then 31-byte words
        pending.push_back(host::h_tag("FSv1-ABSORB-BYTES"));
        for (size_t o = 0; o < n; o += 31) pending.push_back(host::h_from_le_bytes_mod_order(b + o, std::min<size_t>(31, n - o)));
    }
    void absorb_str(const char* s) { absorb_bytes((const uint8_t*)s, strlen(s)); }
    void absorb_u64(uint64_t x) { uint8_t b[8]; for (int j = 0; j < 8; ++j) b[j] = (uint8_t)(x >> (8 * j)); absorb_bytes(b, 8); }
    int32_t run(bool finish, fr_t* result) {
        DevBuf f; const size_t n = pending.size();
        if (n) { STARK_HIP(ctx, f.alloc(ctx, n * sizeof(fr_t))); STARK_HIP(ctx, hipMemcpyAsync(f.p, pending.data(), n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream)); }
        if (use_chain(ctx, tp->dev, 1, 1)) {
            (void)hipFuncSetAttribute((const void*)k_tr_stream_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
            hipLaunchKernelGGL(k_tr_stream_chain, dim3(1), dim3(320), chain_lds_bytes(), ctx->stream, tp->dev, row_consts_of(ctx), state.fr(), (uint32_t*)posb.p, (const fr_t*)f.fr(), (uint64_t)n, finish ? 1 : 0, finish ? out.fr() : (fr_t*)nullptr);
        } else
        hipLaunchKernelGGL(k_tr_stream, dim3(1), dim3(64), coop_lds_bytes(17), ctx->stream, tp->dev, state.fr(), (uint32_t*)posb.p, (const fr_t*)f.fr(), (uint64_t)n, finish ? 1 : 0, finish ? out.fr() : (fr_t*)nullptr);
        STARK_HIP(ctx, hipGetLastError());
        if (finish && result) STARK_HIP(ctx, hipMemcpyAsync(result, out.p, sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
        STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));                                 // `pending` is host memory; a challenge is needed on the host anyway
        pending.clear(); return STARK_OK;
    }
    int32_t challenge(const uint8_t* label, size_t n, fr_t* r) {                           // :92-101
        pending.push_back(host::h_tag("FSv1-CHALLENGE")); absorb_bytes(label, n);
        return run(true, r);
    }
};

// ---- bincode layout of ProofPlain / ProofMF (channel/src/lib.rs:925-979) ---------------------------------------------------
struct BinW {
    std::vector<uint8_t>& b; explicit BinW(std::vector<uint8_t>& v) : b(v) {}
    void u64(uint64_t x) { enc_u64(b, x); }
    void fb(const fr_t& x) { u64(32); enc_fr(b, x); }                                      // FBytes: serde_bytes Vec<u8> of the 32-byte compressed element
    void idxs(const std::vector<size_t>& v) { u64(v.size()); for (size_t x : v) u64(x); }
    void fvec(const std::vector<fr_t>& v) { u64(v.size()); for (auto& x : v) fb(x); }
    void mproof(const MerkleProofHost& p) {                                                // MerkleProofBytes { arity, group_sizes, indices, siblings }
        u64(p.arity);
        u64(p.group_sizes.size()); for (auto& l : p.group_sizes) { u64(l.size()); for (uint8_t x : l) b.push_back(x); }
        idxs(p.indices);
        u64(p.siblings.size()); for (auto& l : p.siblings) fvec(l);
    }
};
struct BinR {
    ByteReader R; explicit BinR(const uint8_t* p, size_t n) : R(p, n) {}
    fr_t fb() { if (R.u64() != 32) R.ok = false; return R.fr(); }
    bool idxs(std::vector<size_t>& v) { size_t k = R.len(8); v.resize(k); for (size_t i = 0; i < k; ++i) v[i] = (size_t)R.u64(); return R.ok; }
    bool fvec(std::vector<fr_t>& v) { size_t k = R.len(40); v.resize(k); for (size_t i = 0; i < k && R.ok; ++i) v[i] = fb(); return R.ok; }
    bool mproof(MerkleProofHost& p) {
        p.arity = (size_t)R.u64();
        size_t g = R.len(8); p.group_sizes.assign(g, {}); for (size_t i = 0; i < g && R.ok; ++i) { size_t k = R.len(1); p.group_sizes[i].resize(k); for (size_t j = 0; j < k; ++j) p.group_sizes[i][j] = R.u8(); }
        if (!idxs(p.indices)) return false;
        size_t a = R.len(8); p.siblings.assign(a, {}); for (size_t i = 0; i < a && R.ok; ++i) fvec(p.siblings[i]);
        return R.ok;
    }
};
struct RoundMFHost { fr_t c0, c1, next_root; std::vector<size_t> cur_indices, next_indices; std::vector<fr_t> cur_values, next_values; MerkleProofHost cur_proof, next_proof; };

static inline std::vector<uint8_t> lab_idx(const char* base, uint64_t i) { std::vector<uint8_t> l((const uint8_t*)base, (const uint8_t*)base + strlen(base)); for (int j = 0; j < 8; ++j) l.push_back((uint8_t)(i >> (8 * j))); return l; }
static void send_digest(DevTranscript& T, const char* label, const fr_t& d) { T.absorb_str("CHAN/SEND/DIGEST"); T.absorb_str(label); T.absorb_field(d); }   // :22-26
static void send_opening(DevTranscript& T, const std::vector<size_t>& idx, const std::vector<fr_t>& vals, const MerkleProofHost& pr) {                  // :32-62
    T.absorb_str("CHAN/SEND/OPEN");
    for (size_t i : idx) T.absorb_u64((uint64_t)i);
    for (auto& v : vals) T.absorb_field(v);
    T.absorb_str("PROOF/ARITY"); T.absorb_u64((uint64_t)pr.arity);
    T.absorb_str("PROOF/GROUP_SIZES");
    for (auto& l : pr.group_sizes) { T.absorb_u64((uint64_t)l.size()); for (uint8_t sz : l) T.absorb_bytes(&sz, 1); }
    T.absorb_str("PROOF/SIBLINGS");
    for (auto& l : pr.siblings) { T.absorb_u64((uint64_t)l.size()); for (auto& s : l) T.absorb_field(s); }
}

// MerkleCommitment's parameters (commitment/src/lib.rs:48-51), cached per context next to the other parameter sets (key -17)
static int32_t commit_params(stark_ctx* ctx, stark_params** out) {
    auto it = ctx->merkle_params.find(-17);
    if (it != ctx->merkle_params.end()) { *out = it->second; return STARK_OK; }
    stark_params* P = nullptr; const char* seed = "POSEIDON-T17-X5-SEED";
    STARK_TRY(stark_poseidon_params_t17_seed(ctx, (const uint8_t*)seed, strlen(seed), &P));
    ctx->merkle_params[-17] = P; *out = P; return STARK_OK;
}
// (c0, c1) of a layer, on the host (the protocol sends them)
static int32_t round_coeffs(stark_ctx* ctx, const fr_t* layer, size_t len, fr_t* c0, fr_t* c1) {
    const uint64_t np = len / 2; const unsigned grid = (unsigned)std::min<uint64_t>((np + 255) / 256, 1024);
    DevBuf part, res; STARK_HIP(ctx, part.alloc(ctx, (size_t)grid * 2 * sizeof(fr_t))); STARK_HIP(ctx, res.alloc(ctx, 2 * sizeof(fr_t)));
    hipLaunchKernelGGL(k_sc_coeffs, dim3(grid), dim3(256), 0, ctx->stream, layer, np, part.fr());
    hipLaunchKernelGGL(k_sc_coeffs_final, dim3(1), dim3(64), 0, ctx->stream, (const fr_t*)part.fr(), (uint64_t)grid, res.fr());
    STARK_HIP(ctx, hipGetLastError());
    fr_t h[2]; STARK_HIP(ctx, hipMemcpyAsync(h, res.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *c0 = h[0]; *c1 = h[1]; return STARK_OK;
}
static int32_t fold(stark_ctx* ctx, const fr_t* layer, size_t len, const fr_t& r, fr_t* next) {
    const uint64_t np = len / 2;
    hipLaunchKernelGGL(k_sc_fold, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, ctx->stream, layer, np, r, next);
    STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
}
static int32_t read_elems(stark_ctx* ctx, const fr_t* dev, const std::vector<size_t>& idx, std::vector<fr_t>& out) {
    out.resize(idx.size()); if (idx.empty()) return STARK_OK;
    DevBuf di, dout; STARK_HIP(ctx, di.alloc(ctx, idx.size() * 8)); STARK_HIP(ctx, dout.alloc(ctx, idx.size() * sizeof(fr_t)));
    std::vector<uint64_t> ix(idx.begin(), idx.end());
    STARK_HIP(ctx, hipMemcpyAsync(di.p, ix.data(), ix.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_gather, dim3((unsigned)((ix.size() + 255) / 256)), dim3(256), 0, ctx->stream, dev, (const uint64_t*)di.p, (uint64_t)ix.size(), dout.fr());
    STARK_HIP(ctx, hipGetLastError());
    STARK_HIP(ctx, hipMemcpyAsync(out.data(), dout.p, ix.size() * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}
static int32_t tree_root(stark_ctx* ctx, stark_tree* t, fr_t* r) {
    STARK_HIP(ctx, hipMemcpyAsync(r, t->levels.back(), sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}
struct TreeHolder { stark_tree* t = nullptr; ~TreeHolder() { if (t) stark_merkle_free(t); } };

// prove_plain (:1045-1076) on a device-resident witness of 2^k elements
static int32_t prove_plain_impl(stark_ctx* ctx, const fr_t* witness, size_t k, uint64_t tree_label, stark_proof** out) {
    if (k > 40) return ctx->fail(STARK_ERR_INVALID_ARG, "k too large");
    const size_t n = (size_t)1 << k;
    stark_params* cp = nullptr; STARK_TRY(commit_params(ctx, &cp));
    DevTranscript T(ctx); STARK_TRY(T.init((const uint8_t*)"E2E/PLAIN", 9));
    TreeHolder tree; STARK_TRY(merkle_build_on(ctx, ctx->stream, cp, 16, tree_label, witness, n, 0, nullptr, 1, 0, 0, 0, false, &tree.t));   // MerkleProver::commit_vector (:172-179)
    fr_t root; STARK_TRY(tree_root(ctx, tree.t, &root));
    send_digest(T, "commit/root", root);
    DevBuf bufA, bufB; STARK_HIP(ctx, bufA.alloc(ctx, std::max<size_t>(n / 2, 1) * sizeof(fr_t))); STARK_HIP(ctx, bufB.alloc(ctx, std::max<size_t>(n / 4, 1) * sizeof(fr_t)));
    const fr_t* layer = witness; size_t len = n;
    stark_proof* P = new stark_proof(); BinW W(P->bytes);
    auto bail = [&](int32_t rc) { delete P; return rc; };
    std::vector<std::pair<fr_t, fr_t>> rounds;
    fr_t c0, c1;
    if (k == 0) { std::vector<fr_t> v; int32_t rc = read_elems(ctx, witness, {0}, v); if (rc) return bail(rc); T.absorb_str("SUMCHECK/CLAIM"); T.absorb_field(v[0]); }
    for (size_t i = 0; i < k; ++i) {
        { int32_t rc = round_coeffs(ctx, layer, len, &c0, &c1); if (rc) return bail(rc); }
        if (i == 0) { T.absorb_str("SUMCHECK/CLAIM"); T.absorb_field(host::h_add(host::h_add(c0, c0), c1)); }      // send_claim (:434-446): s = sum of the table = 2 c0 + c1
        T.absorb_str("SUMCHECK/ROUND"); T.absorb_u64((uint64_t)i);                                                // round (:448-472)
        T.absorb_str("COEFF/c0"); T.absorb_field(c0); T.absorb_str("COEFF/c1"); T.absorb_field(c1);
        fr_t r; { auto lb = lab_idx("sumcheck/r", i); int32_t rc = T.challenge(lb.data(), lb.size(), &r); if (rc) return bail(rc); }
        fr_t* nx = (i & 1) ? bufB.fr() : bufA.fr();
        { int32_t rc = fold(ctx, layer, len, r, nx); if (rc) return bail(rc); }
        layer = nx; len /= 2; rounds.push_back({c0, c1});
    }
    std::vector<fr_t> fin; { int32_t rc = read_elems(ctx, layer, {0}, fin); if (rc) return bail(rc); }
    T.absorb_str("SUMCHECK/FINAL/EVAL"); T.absorb_field(fin[0]);                                                   // :474-484 (bound; no further challenge is drawn)
    W.fb(root); W.u64(rounds.size()); for (auto& r : rounds) { W.fb(r.first); W.fb(r.second); }
    P->bytes.push_back(0);                                                                                         // extra_openings: None
    W.fb(fin[0]); P->size_estimate = P->bytes.size();
    *out = P; return STARK_OK;
}

// mf_round_challenge_from_root (:592-598): a fresh transcript per round
static int32_t mf_round_challenge(stark_ctx* ctx, size_t round_idx, const fr_t& prev_root, fr_t* r) {
    DevTranscript T(ctx); STARK_TRY(T.init((const uint8_t*)"SUMCHECK-MF/ROUND-CHAL", 22));
    T.absorb_str("SUMCHECK/MF/R"); T.absorb_u64((uint64_t)round_idx); T.absorb_field(prev_root);
    return T.challenge((const uint8_t*)"r_i", 3, r);
}
static inline size_t mf_query_index(const fr_t& r, size_t half) {                            // :667-676
    const fr_t c = fr_to_canonical<PallasFr>(r); uint64_t acc = 0;
    for (int i = 0; i < 4; ++i) acc ^= (uint64_t)c.v[2 * i] | ((uint64_t)c.v[2 * i + 1] << 32);
    return (size_t)(acc % (uint64_t)half);
}
// prove_mf (:1130-1172)
static int32_t prove_mf_impl(stark_ctx* ctx, const fr_t* witness, size_t k, uint64_t tree_label, size_t qpr, stark_proof** out) {
    if (k > 40) return ctx->fail(STARK_ERR_INVALID_ARG, "k too large");
    const size_t n = (size_t)1 << k;
    stark_params* cp = nullptr; STARK_TRY(commit_params(ctx, &cp));
    DevTranscript T(ctx); STARK_TRY(T.init((const uint8_t*)"E2E/MF", 6));
    TreeHolder cur_tree; STARK_TRY(merkle_build_on(ctx, ctx->stream, cp, 16, tree_label, witness, n, 0, nullptr, 1, 0, 0, 0, false, &cur_tree.t));     // SumCheckMFProver::new (:601-622)
    fr_t cur_root; STARK_TRY(tree_root(ctx, cur_tree.t, &cur_root));
    send_digest(T, "sumcheck-mf/root/0", cur_root);
    const fr_t initial_root = cur_root;
    DevBuf bufA, bufB; STARK_HIP(ctx, bufA.alloc(ctx, std::max<size_t>(n / 2, 1) * sizeof(fr_t))); STARK_HIP(ctx, bufB.alloc(ctx, std::max<size_t>(n / 4, 1) * sizeof(fr_t)));
    const fr_t* layer = witness; size_t len = n;
    std::vector<RoundMFHost> rounds; fr_t c0, c1;
    if (k == 0) { std::vector<fr_t> v; STARK_TRY(read_elems(ctx, witness, {0}, v)); T.absorb_str("SUMCHECK/MF/CLAIM"); T.absorb_field(v[0]); }
    for (size_t i = 0; i < k; ++i) {                                                        // round (:631-737)
        RoundMFHost R;
        STARK_TRY(round_coeffs(ctx, layer, len, &c0, &c1));
        if (i == 0) { T.absorb_str("SUMCHECK/MF/CLAIM"); T.absorb_field(host::h_add(host::h_add(c0, c0), c1)); }     // send_claim (:624-629)
        T.absorb_str("SUMCHECK/MF/ROUND"); T.absorb_u64((uint64_t)i);
        T.absorb_str("COEFF/c0"); T.absorb_field(c0); T.absorb_str("COEFF/c1"); T.absorb_field(c1);
        fr_t r; STARK_TRY(mf_round_challenge(ctx, i, cur_root, &r));
        const size_t half = len / 2;
        fr_t* nx = (i & 1) ? bufB.fr() : bufA.fr();
        STARK_TRY(fold(ctx, layer, len, r, nx));
        TreeHolder next_tree; STARK_TRY(merkle_build_on(ctx, ctx->stream, cp, 16, tree_label, nx, half, 0, nullptr, 1, 0, 0, 0, false, &next_tree.t));
        fr_t next_root; STARK_TRY(tree_root(ctx, next_tree.t, &next_root));
        send_digest(T, "sumcheck-mf/root/next", next_root);
        const size_t q_target = std::min(std::max(qpr, (size_t)1), half);                   // :656
        std::set<size_t> qs; size_t attempt = 0, j = 0; const size_t max_attempts = std::max(q_target * 16, (size_t)16);
        while (qs.size() < q_target && attempt < max_attempts) {
            std::vector<uint8_t> ql((const uint8_t*)"sumcheck-mf/q", (const uint8_t*)"sumcheck-mf/q" + 13);
            for (int b = 0; b < 8; ++b) ql.push_back((uint8_t)((uint64_t)i >> (8 * b))); for (int b = 0; b < 8; ++b) ql.push_back((uint8_t)((uint64_t)j >> (8 * b)));
            fr_t rr; STARK_TRY(T.challenge(ql.data(), ql.size(), &rr));
            if (half > 0) qs.insert(mf_query_index(rr, half));
            ++j; ++attempt;
        }
        if (qs.size() < q_target) for (size_t idx = 0; idx < half && qs.size() < q_target; ++idx) qs.insert(idx);      // :683-690
        std::vector<size_t> queries(qs.begin(), qs.end());
        for (size_t jj : queries) { R.cur_indices.push_back(2 * jj); R.cur_indices.push_back(2 * jj + 1); }
        STARK_TRY(read_elems(ctx, layer, R.cur_indices, R.cur_values));
        STARK_TRY(merkle_open_host(cur_tree.t, R.cur_indices, R.cur_proof));
        R.next_indices = queries; STARK_TRY(read_elems(ctx, nx, queries, R.next_values));
        STARK_TRY(merkle_open_host(next_tree.t, R.next_indices, R.next_proof));
        send_opening(T, R.cur_indices, R.cur_values, R.cur_proof);
        send_opening(T, R.next_indices, R.next_values, R.next_proof);
        R.c0 = c0; R.c1 = c1; R.next_root = next_root;
        rounds.push_back(std::move(R));
        std::swap(cur_tree.t, next_tree.t); cur_root = next_root; layer = nx; len = half;
    }
    std::vector<fr_t> fin; STARK_TRY(read_elems(ctx, layer, {0}, fin));
    T.absorb_str("SUMCHECK/MF/FINAL/EVAL"); T.absorb_field(fin[0]);          // finalize_eval (:732-738): bound, nothing is drawn after it (queued, never launched)
    stark_proof* P = new stark_proof(); BinW W(P->bytes);
    W.fb(initial_root); W.u64(rounds.size());
    for (auto& R : rounds) { W.fb(R.c0); W.fb(R.c1); W.fb(R.next_root); W.idxs(R.cur_indices); W.fvec(R.cur_values); W.mproof(R.cur_proof); W.idxs(R.next_indices); W.fvec(R.next_values); W.mproof(R.next_proof); }
    W.fb(fin[0]); P->size_estimate = P->bytes.size();
    *out = P; return STARK_OK;
}

// verifier hashing with MerkleCommitment's parameters (a t = 17 set that is NOT poseidon_params_for_arity(16))
struct CommitVerifyHasher : VerifyHasher {
    stark_ctx* ctx; stark_params* cp; CommitVerifyHasher(stark_ctx* c, stark_params* p) : ctx(c), cp(p) {}
    int32_t leaf_pairs(const fr_t*, const fr_t*, size_t, fr_t*) override { return STARK_ERR_UNSUPPORTED; }
    int32_t ds_pair_leaves(size_t, uint64_t, const uint64_t*, const fr_t*, const fr_t*, size_t, fr_t*) override { return STARK_ERR_UNSUPPORTED; }
    int32_t ds_nodes(size_t arity, size_t chunk, uint32_t level, uint64_t label, const uint64_t* positions, const fr_t* children, size_t n, fr_t* out) override {
        if (!n) return STARK_OK;
        DevBuf dp, dc, dout; STARK_HIP(ctx, dp.alloc(ctx, n * 8)); STARK_HIP(ctx, dc.alloc(ctx, n * chunk * sizeof(fr_t))); STARK_HIP(ctx, dout.alloc(ctx, n * sizeof(fr_t)));
        STARK_HIP(ctx, hipMemcpyAsync(dp.p, positions, n * 8, hipMemcpyHostToDevice, ctx->stream)); STARK_HIP(ctx, hipMemcpyAsync(dc.p, children, n * chunk * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
        STARK_TRY(hash_ds_scattered(ctx, cp, 0, arity, chunk, level, label, (const uint64_t*)dp.p, dc.fr(), nullptr, n, dout.fr()));
        STARK_HIP(ctx, hipMemcpyAsync(out, dout.p, n * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
    }
};

// verify_plain (:1080-1128).  A failed check answers `false` (inside the reference it is a failed assert_eq!, i.e. a panic).
static int32_t verify_plain_impl(stark_ctx* ctx, const uint8_t* bytes, size_t len, bool& ok) {
    ok = false;
    BinR D(bytes, len); const fr_t root = D.fb(); const size_t nr = D.R.len(80);
    std::vector<std::pair<fr_t, fr_t>> rounds(nr); for (size_t i = 0; i < nr && D.R.ok; ++i) { rounds[i].first = D.fb(); rounds[i].second = D.fb(); }
    if (D.R.u8() != 0) D.R.ok = false;
    const fr_t final_eval = D.fb();
    if (!D.R.ok || D.R.left()) return STARK_OK;
    if (rounds.empty()) return STARK_OK;                                                     // :1100-1102
    DevTranscript T(ctx); STARK_TRY(T.init((const uint8_t*)"E2E/PLAIN", 9));
    send_digest(T, "commit/root", root);
    fr_t running = host::h_add(host::h_add(rounds[0].first, rounds[0].first), rounds[0].second);
    T.absorb_str("SUMCHECK/CLAIM"); T.absorb_field(running);
    for (size_t i = 0; i < rounds.size(); ++i) {
        const fr_t& c0 = rounds[i].first; const fr_t& c1 = rounds[i].second;
        T.absorb_str("SUMCHECK/ROUND"); T.absorb_u64((uint64_t)i); T.absorb_str("COEFF/c0"); T.absorb_field(c0); T.absorb_str("COEFF/c1"); T.absorb_field(c1);
        if (!fr_eq(host::h_add(host::h_add(c0, c0), c1), running)) return STARK_OK;          // :511-512
        fr_t r; { auto lb = lab_idx("sumcheck/r", i); STARK_TRY(T.challenge(lb.data(), lb.size(), &r)); }
        running = host::h_add(c0, host::h_mul(c1, r));
    }
    ok = fr_eq(final_eval, running); return STARK_OK;                                        // :528
}
// verify_mf (:1176-1240)
static int32_t verify_mf_impl(stark_ctx* ctx, uint64_t tree_label, const uint8_t* bytes, size_t len, bool& ok) {
    ok = false;
    BinR D(bytes, len); const fr_t initial_root = D.fb(); const size_t nr = D.R.len(120);
    std::vector<RoundMFHost> rounds(nr);
    for (size_t i = 0; i < nr && D.R.ok; ++i) { RoundMFHost& R = rounds[i]; R.c0 = D.fb(); R.c1 = D.fb(); R.next_root = D.fb(); D.idxs(R.cur_indices); D.fvec(R.cur_values); D.mproof(R.cur_proof); D.idxs(R.next_indices); D.fvec(R.next_values); D.mproof(R.next_proof); }
    const fr_t final_eval = D.fb();
    if (!D.R.ok || D.R.left()) return STARK_OK;
    stark_params* cp = nullptr; STARK_TRY(commit_params(ctx, &cp));
    CommitVerifyHasher H(ctx, cp);
    bool have = false; fr_t running = host::h_zero(), prev_root = initial_root;
    for (size_t i = 0; i < rounds.size(); ++i) {
        const RoundMFHost& R = rounds[i];
        const fr_t twoc = host::h_add(host::h_add(R.c0, R.c0), R.c1);
        if (have && !fr_eq(twoc, running)) return STARK_OK;                                  // start_round (:803-804)
        fr_t r; STARK_TRY(mf_round_challenge(ctx, i, prev_root, &r));                        // derive_round_challenge (:807-810)
        bool good = false;                                                                   // verify_fold_openings (:821-869)
        STARK_TRY(verify_many_ds_host(H, 16, prev_root, R.cur_indices, R.cur_values, R.cur_proof, tree_label, good)); if (!good) return STARK_OK;
        STARK_TRY(verify_many_ds_host(H, 16, R.next_root, R.next_indices, R.next_values, R.next_proof, tree_label, good)); if (!good) return STARK_OK;
        if (R.cur_indices.size() != R.cur_values.size() || R.next_indices.size() != R.next_values.size()) return STARK_OK;
        std::map<size_t, std::pair<std::pair<bool, fr_t>, std::pair<bool, fr_t>>> pairs;
        for (size_t t = 0; t < R.cur_indices.size(); ++t) { const size_t ix = R.cur_indices[t]; auto& e = pairs[ix / 2]; if (ix % 2 == 0) e.first = {true, R.cur_values[t]}; else e.second = {true, R.cur_values[t]}; }
        for (size_t t = 0; t < R.next_indices.size(); ++t) {
            auto it = pairs.find(R.next_indices[t]);
            if (it == pairs.end() || !it->second.first.first || !it->second.second.first) return STARK_OK;
            const fr_t a = it->second.first.second, b = it->second.second.second;
            if (!fr_eq(host::h_add(a, host::h_mul(r, host::h_sub(b, a))), R.next_values[t])) return STARK_OK;
        }
        running = host::h_add(R.c0, host::h_mul(R.c1, r)); have = true; prev_root = R.next_root;
    }
    ok = !have || fr_eq(final_eval, running); return STARK_OK;                               // :1237-1238
}

}  // namespace

// ---- the streaming transcript as an object of the ABI (transcript/src/lib.rs:48-117) ------------------------------------------
struct stark_transcript { CtxRef ref_; DevTranscript T; explicit stark_transcript(stark_ctx* c) : T(c) { ref_.bind(c); } };

extern "C" {

// Transcript::new(label, default_params()) — the state lives on the device; absorbs are queued and run with the next challenge.
int32_t stark_transcript_new(stark_ctx_t* ctx, const uint8_t* label, size_t label_len, stark_transcript_t** out) {
    if (!ctx || !out || (!label && label_len)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    stark_transcript* t = new stark_transcript(ctx);
    int32_t rc = t->T.init(label, label_len); if (rc) { delete t; return rc; }
    *out = t; return STARK_OK;
}
int32_t stark_transcript_absorb_bytes(stark_transcript_t* t, const uint8_t* bytes, size_t n) { if (!t || (!bytes && n)) return STARK_ERR_INVALID_ARG; t->T.absorb_bytes(bytes, n); return STARK_OK; }
int32_t stark_transcript_absorb_fields(stark_transcript_t* t, const uint64_t* fields, size_t n) {
    if (!t || (!fields && n)) return STARK_ERR_INVALID_ARG;
    for (size_t i = 0; i < n; ++i) t->T.absorb_field(load_fr(fields + 4 * i));
    return STARK_OK;
}
int32_t stark_transcript_challenge(stark_transcript_t* t, const uint8_t* label, size_t label_len, uint64_t* out4) {
    if (!t || !out4 || (!label && label_len)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(t->T.ctx));
    fr_t r; STARK_TRY(t->T.challenge(label, label_len, &r)); store_fr(out4, r); return STARK_OK;
}
// Transcript::challenges(label, n): challenge(label || le64(i)) for i < n (:103-112)
int32_t stark_transcript_challenges(stark_transcript_t* t, const uint8_t* label, size_t label_len, size_t n, uint64_t* out) {
    if (!t || (!out && n) || (!label && label_len)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(t->T.ctx));
    for (size_t i = 0; i < n; ++i) {
        std::vector<uint8_t> tag(label, label + label_len); for (int j = 0; j < 8; ++j) tag.push_back((uint8_t)((uint64_t)i >> (8 * j)));
        fr_t r; STARK_TRY(t->T.challenge(tag.data(), tag.size(), &r)); store_fr(out + 4 * i, r);
    }
    return STARK_OK;
}
int32_t stark_transcript_free(stark_transcript_t* t) { if (!t) return STARK_ERR_INVALID_ARG; delete t; return STARK_OK; }

// The reference's bench inputs (channel/benches/end_to_end.rs:249-253): `ncols` vectors of n elements drawn one after the other
// from ONE StdRng::seed_from_u64(seed) with ark-ff's Fp::rand (rand_core 0.6.4 PCG32 seed expansion, ChaCha12, rejection sampling of
// 255-bit candidates; the accepted limbs ARE the Montgomery representation).  Host-only: no context, no device.
int32_t stark_ref_bench_inputs(uint64_t seed, size_t n, size_t ncols, uint64_t* out) {
    if (!out && n * ncols) return STARK_ERR_INVALID_ARG;
    uint8_t key[32]; uint64_t state = seed;
    for (int c = 0; c < 8; ++c) {                                                          // SeedableRng::seed_from_u64
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        const uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
        const uint32_t x = (xs >> rot) | (xs << ((32 - rot) & 31));
        key[4 * c] = (uint8_t)x; key[4 * c + 1] = (uint8_t)(x >> 8); key[4 * c + 2] = (uint8_t)(x >> 16); key[4 * c + 3] = (uint8_t)(x >> 24);
    }
    host::ChaCha12Rng rng(key);
    for (size_t i = 0; i < n * ncols; ++i) {
        for (;;) {                                                                         // Fp::rand: 4 limbs, top bit cleared, accept below the modulus
            uint64_t l[4]; for (int j = 0; j < 4; ++j) l[j] = rng.next_u64();
            l[3] &= 0x7FFFFFFFFFFFFFFFull;
            uint32_t t[9]; for (int j = 0; j < 4; ++j) { t[2 * j] = (uint32_t)l[j]; t[2 * j + 1] = (uint32_t)(l[j] >> 32); } t[8] = 0;
            if (fr_geq_p<PallasFr>(t)) continue;
            for (int j = 0; j < 4; ++j) out[4 * i + j] = l[j];
            break;
        }
    }
    return STARK_OK;
}

// CommitmentScheme for MerkleCommitment (commitment/src/lib.rs:80-114): arity 16, tree_label = cfg.ds_tag, parameters "POSEIDON-T17-X5-SEED".
int32_t stark_commitment_commit(stark_ctx_t* ctx, uint64_t ds_tag, const uint64_t* leaves, size_t n, stark_tree_t** out) {
    if (!ctx || !leaves || !out) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    stark_params* cp = nullptr; STARK_TRY(commit_params(ctx, &cp));
    DevBuf d; STARK_HIP(ctx, d.alloc(ctx, std::max<size_t>(n, 1) * sizeof(fr_t)));
    if (n) STARK_HIP(ctx, hipMemcpyAsync(d.p, leaves, n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    STARK_TRY(merkle_build_on(ctx, ctx->stream, cp, 16, ds_tag, d.fr(), n, 0, nullptr, 1, 0, 0, 0, false, out));     // commit (:85-90); open = stark_merkle_open (:92-94)
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}
// verify (:96-113): verify_many_ds with the static t = 17 parameters lifted to the dynamic form
int32_t stark_commitment_verify(stark_ctx_t* ctx, uint64_t ds_tag, const uint64_t* root4, const size_t* indices, size_t k, const uint64_t* values, const uint8_t* proof, size_t len, int32_t* accepted) {
    if (!ctx || !root4 || (!indices && k) || (!values && k) || (!proof && len) || !accepted) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    *accepted = 0;
    ByteReader R(proof, len); MerkleProofHost pr; if (!dec_mproof(R, pr) || R.left()) return STARK_OK;
    stark_params* cp = nullptr; STARK_TRY(commit_params(ctx, &cp));
    std::vector<size_t> ix(indices, indices + k); std::vector<fr_t> v(k); for (size_t i = 0; i < k; ++i) v[i] = load_fr(values + 4 * i);
    CommitVerifyHasher H(ctx, cp); bool ok = false;
    STARK_TRY(verify_many_ds_host(H, 16, load_fr(root4), ix, v, pr, ds_tag, ok));
    *accepted = ok ? 1 : 0; return STARK_OK;
}

// Mle::evaluate(r) (channel/src/lib.rs:279-295): k folds layer[i] = (1 - r_j) layer[2i] + r_j layer[2i+1]; table of 2^k elements (host).
int32_t stark_mle_evaluate(stark_ctx_t* ctx, const uint64_t* table, size_t k, const uint64_t* r, uint64_t* out4) {
    if (!ctx || !table || (!r && k) || !out4 || k > 40) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    const size_t n = (size_t)1 << k;
    DevBuf a, b; STARK_HIP(ctx, a.alloc(ctx, n * sizeof(fr_t))); STARK_HIP(ctx, b.alloc(ctx, std::max<size_t>(n / 2, 1) * sizeof(fr_t)));
    STARK_HIP(ctx, hipMemcpyAsync(a.p, table, n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    fr_t* cur = a.fr(); fr_t* nxt = b.fr(); size_t len = n;
    for (size_t j = 0; j < k; ++j) { STARK_TRY(fold(ctx, cur, len, load_fr(r + 4 * j), nxt)); std::swap(cur, nxt); len /= 2; }
    fr_t v; STARK_HIP(ctx, hipMemcpyAsync(&v, cur, sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    store_fr(out4, v); return STARK_OK;
}

int32_t stark_sumcheck_prove_plain_dev(stark_ctx_t* ctx, const uint64_t* witness, size_t k, uint64_t tree_label, stark_proof_t** out) {
    if (!ctx || !witness || !out) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    return prove_plain_impl(ctx, as_fr(witness), k, tree_label, out);
}
int32_t stark_sumcheck_prove_mf_dev(stark_ctx_t* ctx, const uint64_t* witness, size_t k, uint64_t tree_label, size_t queries_per_round, stark_proof_t** out) {
    if (!ctx || !witness || !out) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    return prove_mf_impl(ctx, as_fr(witness), k, tree_label, queries_per_round, out);
}
int32_t stark_sumcheck_prove_plain(stark_ctx_t* ctx, const uint64_t* witness, size_t k, uint64_t tree_label, stark_proof_t** out) {
    if (!ctx || !witness || !out || k > 40) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    const size_t n = (size_t)1 << k; DevBuf d; STARK_HIP(ctx, d.alloc(ctx, n * sizeof(fr_t)));
    STARK_HIP(ctx, hipMemcpyAsync(d.p, witness, n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    return prove_plain_impl(ctx, d.fr(), k, tree_label, out);
}
int32_t stark_sumcheck_prove_mf(stark_ctx_t* ctx, const uint64_t* witness, size_t k, uint64_t tree_label, size_t queries_per_round, stark_proof_t** out) {
    if (!ctx || !witness || !out || k > 40) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    const size_t n = (size_t)1 << k; DevBuf d; STARK_HIP(ctx, d.alloc(ctx, n * sizeof(fr_t)));
    STARK_HIP(ctx, hipMemcpyAsync(d.p, witness, n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    return prove_mf_impl(ctx, d.fr(), k, tree_label, queries_per_round, out);
}
int32_t stark_sumcheck_verify_plain(stark_ctx_t* ctx, size_t k, uint64_t tree_label, const uint8_t* proof, size_t len, int32_t* accepted) {
    if (!ctx || (!proof && len) || !accepted) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx)); (void)k; (void)tree_label;     // verify_plain reads neither vk.k (it walks proof.rounds) nor the tree label
    bool ok = false; STARK_TRY(verify_plain_impl(ctx, proof, len, ok)); *accepted = ok ? 1 : 0; return STARK_OK;
}
int32_t stark_sumcheck_verify_mf(stark_ctx_t* ctx, size_t k, uint64_t tree_label, size_t queries_per_round, const uint8_t* proof, size_t len, int32_t* accepted) {
    if (!ctx || (!proof && len) || !accepted) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx)); (void)k; (void)queries_per_round;
    bool ok = false; STARK_TRY(verify_mf_impl(ctx, tree_label, proof, len, ok)); *accepted = ok ? 1 : 0; return STARK_OK;
}

}  // extern "C"

// stark_mlwe_amd/csrc/capi_verify.hip — the verifier entry points of the C-ABI (next row N3): host logic of fri_verify.hpp with
// every hash batched onto the GPU kernels the prover uses (leaf-pair kernel, DS node kernels with scattered positions).
//   stark_deep_fri_verify            deep_fri_verify                    crates/deep_ali/src/fri.rs:643-762
//   stark_merkle_verify_many_ds      MerkleProver::verify_single        crates/merkle/src/lib.rs:587-722, 800-812
//   stark_merkle_verify_pairs_ds     MerkleProver::verify_pairs         crates/merkle/src/lib.rs:723-773, 841-855
#include <cstring>
#include "ctx.hpp"
#include "fri_verify.hpp"

using namespace stark;

namespace stark {
int32_t hash_ds_scattered(stark_ctx* ctx, stark_params* p, int mode, size_t arity, size_t chunk, uint32_t level, uint64_t label, const uint64_t* positions_dev,
                          const fr_t* in0, const fr_t* in1, size_t n_hashes, fr_t* out);   // capi_core.hip
}

namespace {
struct GpuVerifyHasher : VerifyHasher {
    stark_ctx* ctx; explicit GpuVerifyHasher(stark_ctx* c) : ctx(c) {}
    // host vectors -> pooled device buffers -> kernel -> host (a few hundred hashes per call; one synchronisation each)
    int32_t up(DevBuf& d, const void* src, size_t bytes) {
        STARK_HIP(ctx, d.alloc(ctx, bytes));
        if (bytes) STARK_HIP(ctx, hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return STARK_OK;
    }
    int32_t down(fr_t* dst, const DevBuf& d, size_t n) {
        STARK_HIP(ctx, hipMemcpyAsync(dst, d.p, n * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
        STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
    }
    int32_t leaf_pairs(const fr_t* f, const fr_t* s, size_t n, fr_t* out) override {
        if (!n) return STARK_OK;
        DevBuf df, ds, dh; STARK_TRY(up(df, f, n * sizeof(fr_t))); STARK_TRY(up(ds, s, n * sizeof(fr_t))); STARK_HIP(ctx, dh.alloc(ctx, n * sizeof(fr_t)));
        STARK_TRY(leaf_pair_hash_on(ctx, ctx->stream, df.fr(), ds.fr(), n, 1, dh.fr()));
        return down(out, dh, n);
    }
    int32_t ds_nodes(size_t arity, size_t chunk, uint32_t level, uint64_t label, const uint64_t* positions, const fr_t* children, size_t n, fr_t* out) override {
        if (!n) return STARK_OK;
        stark_params* mp = nullptr; STARK_TRY(ctx_merkle_params(ctx, host::width_for_arity(arity), &mp));
        DevBuf dp, dc, dout; STARK_TRY(up(dp, positions, n * 8)); STARK_TRY(up(dc, children, n * chunk * sizeof(fr_t))); STARK_HIP(ctx, dout.alloc(ctx, n * sizeof(fr_t)));
        STARK_TRY(hash_ds_scattered(ctx, mp, 0, arity, chunk, level, label, (const uint64_t*)dp.p, dc.fr(), nullptr, n, dout.fr()));
        return down(out, dout, n);
    }
    int32_t ds_pair_leaves(size_t arity, uint64_t label, const uint64_t* positions, const fr_t* f, const fr_t* cp, size_t n, fr_t* out) override {
        if (!n) return STARK_OK;
        stark_params* mp = nullptr; STARK_TRY(ctx_merkle_params(ctx, host::width_for_arity(arity), &mp));
        DevBuf dp, df, dc, dout; STARK_TRY(up(dp, positions, n * 8)); STARK_TRY(up(df, f, n * sizeof(fr_t))); STARK_TRY(up(dc, cp, n * sizeof(fr_t))); STARK_HIP(ctx, dout.alloc(ctx, n * sizeof(fr_t)));
        STARK_TRY(hash_ds_scattered(ctx, mp, 1, arity, arity, 0xFFFFFFFFu, label, (const uint64_t*)dp.p, df.fr(), dc.fr(), n, dout.fr()));
        return down(out, dout, n);
    }
};
}  // namespace

extern "C" {

int32_t stark_deep_fri_verify(stark_ctx_t* ctx, const uint8_t* proof, size_t len, const size_t* schedule, size_t L, size_t r, uint64_t seed_z, int32_t* accepted) {
    if (!ctx || (!proof && len) || (!schedule && L) || !accepted) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    (void)seed_z;   // DeepFriParams.seed_z is carried for signature parity: the reference's verifier never reads it (fri.rs:643-762)
    *accepted = 0;
    DeepFriProofHost P; if (!decode_proof(proof, len, P)) return STARK_OK;         // not a well-formed proof: reject
    GpuVerifyHasher H(ctx); bool ok = false;
    STARK_TRY(deep_fri_verify_host(H, P, schedule, L, r, ok));
    *accepted = ok ? 1 : 0; return STARK_OK;
}

static int32_t merkle_verify(stark_ctx_t* ctx, int pairs, size_t cfg_arity, uint64_t tree_label, const uint64_t* root4, const size_t* idx, size_t k, const uint64_t* values, const uint64_t* cp,
                             const uint8_t* proof, size_t len, int32_t* accepted) {
    if (!ctx || !root4 || (!idx && k) || (!values && k) || (pairs && !cp && k) || (!proof && len) || !accepted) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    *accepted = 0;
    if (host::width_for_arity(cfg_arity) < 0 || cfg_arity == 0) return ctx->fail(STARK_ERR_UNSUPPORTED, "unsupported Merkle arity; max supported = 128");   // MerkleChannelCfg::new (poseidon/src/lib.rs:164)
    ByteReader R(proof, len); MerkleProofHost pr; if (!dec_mproof(R, pr) || R.left()) return STARK_OK;
    std::vector<size_t> ix(idx, idx + k); std::vector<fr_t> v(k), c(pairs ? k : 0);
    for (size_t i = 0; i < k; ++i) { v[i] = load_fr(values + 4 * i); if (pairs) c[i] = load_fr(cp + 4 * i); }
    GpuVerifyHasher H(ctx); bool ok = false;
    STARK_TRY(pairs ? verify_pairs_ds_host(H, cfg_arity, load_fr(root4), ix, v, c, pr, tree_label, ok) : verify_many_ds_host(H, cfg_arity, load_fr(root4), ix, v, pr, tree_label, ok));
    *accepted = ok ? 1 : 0; return STARK_OK;
}
int32_t stark_merkle_verify_many_ds(stark_ctx_t* ctx, size_t cfg_arity, uint64_t tree_label, const uint64_t* root4, const size_t* indices, size_t k, const uint64_t* values,
                                    const uint8_t* proof, size_t len, int32_t* accepted) {
    return merkle_verify(ctx, 0, cfg_arity, tree_label, root4, indices, k, values, nullptr, proof, len, accepted);
}
int32_t stark_merkle_verify_pairs_ds(stark_ctx_t* ctx, size_t cfg_arity, uint64_t tree_label, const uint64_t* root4, const size_t* indices, size_t k, const uint64_t* f_vals, const uint64_t* cp_vals,
                                     const uint8_t* proof, size_t len, int32_t* accepted) {
    return merkle_verify(ctx, 1, cfg_arity, tree_label, root4, indices, k, f_vals, cp_vals, proof, len, accepted);
}

}  // extern "C"

// tests/capi_client.cpp — a COMPILED caller of the C-ABI (include/stark_mlwe.h), linked against libstark_mlwe_hip.so: what the
// reference's Rust FFI would do, in C++ (no Python, no torch, no oracle).  Exit code 0 = every check passed;
// exit code 3 = no HIP device (the library refused to create a context: there is no CPU fallback), anything else = failure.
//   g++ -std=c++17 -I include tests/capi_client.cpp -L stark_mlwe_amd -lstark_mlwe_hip -Wl,-rpath,$PWD/stark_mlwe_amd -o capi_client
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "stark_mlwe.h"

#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, ctx ? stark_last_error(ctx) : ""); return 1; } } while (0)
#define OK(call) CHECK((call) == STARK_OK)

static uint64_t mix64(uint64_t x) { x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31); }
// the synthetic generator of DESIGN.md "Synthetic inputs" (host copy): limbs are the stored Montgomery representation
static void synth(uint64_t seed, uint64_t col, size_t n, std::vector<uint64_t>& out) {
    out.resize(4 * n);
    for (size_t i = 0; i < n; ++i) for (uint64_t j = 0; j < 4; ++j) { uint64_t v = mix64(seed + (col << 56) + 4 * i + j); if (j == 3) v &= 0x3FFFFFFFFFFFFFFFull; out[4 * i + j] = v; }
}

int main() {
    stark_ctx_t* ctx = nullptr;
    std::printf("stark_version %d\n", stark_version());
    int32_t rc = stark_ctx_create(0, nullptr, &ctx);
    if (rc == STARK_ERR_HIP) { std::printf("no HIP device: stark_ctx_create -> STARK_ERR_HIP (no CPU fallback)\n"); return 3; }
    CHECK(rc == STARK_OK && ctx);

    // ---- deep_fri_prove -> deep_fri_verify round trip (fri.rs:601-762) -----------------------------------------------
    const size_t n0 = 1 << 10, schedule[2] = {16, 8}, L = 2, r = 8; const uint64_t seed_z = 0xDEEFBAADull;
    std::vector<uint64_t> a, s, e, t; synth(0x5EED, 0, n0, a); synth(0x5EED, 1, n0, s); synth(0x5EED, 2, n0, e); synth(0x5EED, 3, n0, t);
    stark_proof_t* pr = nullptr;
    OK(stark_deep_fri_prove(ctx, a.data(), s.data(), e.data(), t.data(), nullptr, n0, schedule, L, r, seed_z, &pr));
    std::vector<uint8_t> proof(stark_proof_len(pr)); OK(stark_proof_bytes(pr, proof.data()));
    const size_t est = stark_proof_size_estimate(pr); OK(stark_proof_free(pr));
    CHECK(proof.size() > 1000 && est > 1000 && est < proof.size());
    int32_t acc = -1;
    OK(stark_deep_fri_verify(ctx, proof.data(), proof.size(), schedule, L, r, seed_z, &acc)); CHECK(acc == 1);
    proof[proof.size() / 2] ^= 0x20;
    OK(stark_deep_fri_verify(ctx, proof.data(), proof.size(), schedule, L, r, seed_z, &acc)); CHECK(acc == 0);
    proof[proof.size() / 2] ^= 0x20;
    OK(stark_deep_fri_verify(ctx, proof.data(), proof.size(), schedule, L, r + 1, seed_z, &acc)); CHECK(acc == 0);
    // the same inputs again: identical bytes (deterministic Fiat-Shamir)
    OK(stark_deep_fri_prove(ctx, a.data(), s.data(), e.data(), t.data(), nullptr, n0, schedule, L, r, seed_z, &pr));
    std::vector<uint8_t> proof2(stark_proof_len(pr)); OK(stark_proof_bytes(pr, proof2.data())); OK(stark_proof_free(pr));
    CHECK(proof2 == proof);

    // ---- MerkleTree::new -> open -> verify_single (merkle/src/lib.rs:1053-1136) ---------------------------------------
    stark_params_t* p17 = nullptr; OK(stark_poseidon_params_for_width(ctx, 17, &p17));
    std::vector<uint64_t> leaves; synth(7, 1, 300, leaves);
    stark_tree_t* tree = nullptr; OK(stark_merkle_build(ctx, p17, 16, 42, leaves.data(), 300, 0, nullptr, &tree));
    CHECK(stark_merkle_num_levels(tree) == 4 && stark_merkle_level_len(tree, 1) == 19 && stark_merkle_level_len(tree, 3) == 1);
    uint64_t root[4]; OK(stark_merkle_root(tree, root));
    const size_t idx[3] = {5, 17, 299}; size_t plen = 0;
    OK(stark_merkle_open(tree, idx, 3, nullptr, 0, &plen)); std::vector<uint8_t> mp(plen); OK(stark_merkle_open(tree, idx, 3, mp.data(), plen, &plen));
    uint64_t vals[12]; for (int k = 0; k < 3; ++k) std::memcpy(vals + 4 * k, leaves.data() + 4 * idx[k], 32);
    OK(stark_merkle_verify_many_ds(ctx, 16, 42, root, idx, 3, vals, mp.data(), plen, &acc)); CHECK(acc == 1);
    vals[5] ^= 1; OK(stark_merkle_verify_many_ds(ctx, 16, 42, root, idx, 3, vals, mp.data(), plen, &acc)); CHECK(acc == 0); vals[5] ^= 1;
    OK(stark_merkle_verify_many_ds(ctx, 16, 43, root, idx, 3, vals, mp.data(), plen, &acc)); CHECK(acc == 0);
    OK(stark_merkle_free(tree));
    // error behaviour: the reference panics on an empty tree (merkle/src/lib.rs:148); the ABI returns a status
    CHECK(stark_merkle_build(ctx, p17, 16, 0, leaves.data(), 0, 0, nullptr, &tree) == STARK_ERR_INVALID_ARG);
    CHECK(stark_merkle_build(ctx, p17, 8, 0, leaves.data(), 8, 0, nullptr, &tree) == STARK_ERR_INVALID_ARG);      // arity 8 needs t = 9 (:155-161)

    // ---- fft / ifft (crates/fft/src/lib.rs:6-54), both fields --------------------------------------------------------
    for (int field = 0; field < 2; ++field) {
        std::vector<uint64_t> x; synth(11, 7, 1 << 12, x); std::vector<uint64_t> y = x;
        OK(stark_ntt(ctx, field, y.data(), 12, 0, nullptr)); CHECK(y != x);
        OK(stark_ntt(ctx, field, y.data(), 12, 1, nullptr)); CHECK(y == x);
    }
    // ---- fri_fold_layer error behaviour (fri.rs:86-87) --------------------------------------------------------------
    { std::vector<uint64_t> f; synth(3, 0, 64, f); uint64_t z[4] = {3, 0, 0, 0}; std::vector<uint64_t> out(4 * 64);
      CHECK(stark_fri_fold(ctx, f.data(), 64, z, 1, out.data()) == STARK_ERR_INVALID_ARG);
      CHECK(stark_fri_fold(ctx, f.data(), 60, z, 16, out.data()) == STARK_ERR_INVALID_ARG);
      OK(stark_fri_fold(ctx, f.data(), 64, z, 16, out.data())); }
    OK(stark_poseidon_params_free(p17));
    OK(stark_ctx_destroy(ctx)); ctx = nullptr;
    std::printf("capi_client: all checks passed (proof %zu bytes, size estimate %zu)\n", proof.size(), est);
    return 0;
}

// oracle/fingerprint.cpp — TEST INFRASTRUCTURE ONLY.
//
// Re-runs the reference's Criterion harness recipe for the "paper" preset
// (crates/channel/benches/end_to_end.rs:187-270): rng_seed chain 1337 -> seed*1103515245+12345 per
// (preset,k), inputs a,s,e,t = F::rand from StdRng::seed_from_u64(rng_seed), schedule [16,16,8],
// r = 32, seed_z = 0xDEEFBAAD, DeepAliRealBuilder; prints deep_fri_proof_size_bytes so it can be
// compared with the `proof_bytes` column the reference published in crates/channel/benchmarkdata.csv:2-9.
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include "fri.hpp"
using namespace oracle;
int main(int argc, char** argv) {
    int k_lo = argc > 1 ? atoi(argv[1]) : 11, k_hi = argc > 2 ? atoi(argv[2]) : 12;
    uint64_t rng_seed = 1337;
    for (int k = 11; k <= k_hi; ++k) {
        rng_seed = rng_seed * 1103515245ULL + 12345ULL;      // end_to_end.rs:248
        if (k < k_lo) continue;
        size_t n0 = (size_t)1 << k;
        StdRng rng = StdRng::seed_from_u64(rng_seed);
        std::vector<Fr> cols[4];
        for (int c = 0; c < 4; ++c) { cols[c].resize(n0); for (size_t i = 0; i < n0; ++i) cols[c][i] = fr_rand<Fr>(rng); }
        DeepFriParams prm; prm.schedule = {16, 16, 8}; prm.r = 32; prm.seed_z = 0xDEEFBAADULL;
        auto t0 = std::chrono::steady_clock::now();
        DeepFriProof p = deep_fri_prove(cols[0], cols[1], cols[2], cols[3], n0, prm);
        double ps = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        t0 = std::chrono::steady_clock::now();
        bool ok = deep_fri_verify(prm, p);
        double vs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("paper,%d,%zu,%zu,prove_s=%.3f,verify_s=%.3f,verify=%d\n", k, deep_fri_proof_size_bytes(p), encode_proof(p).size(), ps, vs, ok ? 1 : 0);
        fflush(stdout);
    }
    return 0;
}

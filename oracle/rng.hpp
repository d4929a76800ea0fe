// oracle/rng.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// `rand::rngs::StdRng` of rand 0.8.5 == ChaCha12 (rand_chacha 0.3.1, rand_core 0.6.4; Cargo.lock:794-816),
// restated from the published ChaCha definition (RFC 7539 block function, 12 rounds, 64-bit block
// counter in words 12..13, stream id 0 in words 14..15).  Reference call sites:
//   StdRng::from_seed(bytes) + gen::<u64>()   fri.rs:66,71,185-186,492-493,515-520
//   StdRng::seed_from_u64(s) + F::rand        channel/benches/end_to_end.rs:249-253; merkle/src/lib.rs:915-917
#pragma once
#include <cstdint>
#include <cstring>
#include "fr.hpp"

namespace oracle {

struct StdRng {
    uint32_t key[8];
    uint64_t counter;
    uint32_t buf[16];
    int idx;  // next unread word in buf; 16 => empty

    static inline uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    static inline void qr(uint32_t* s, int a, int b, int c, int d) {
        s[a] += s[b]; s[d] ^= s[a]; s[d] = rotl(s[d], 16);
        s[c] += s[d]; s[b] ^= s[c]; s[b] = rotl(s[b], 12);
        s[a] += s[b]; s[d] ^= s[a]; s[d] = rotl(s[d], 8);
        s[c] += s[d]; s[b] ^= s[c]; s[b] = rotl(s[b], 7);
    }
    void refill() {
        uint32_t in[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
        for (int i = 0; i < 8; ++i) in[4 + i] = key[i];
        in[12] = (uint32_t)counter; in[13] = (uint32_t)(counter >> 32); in[14] = 0; in[15] = 0;
        uint32_t s[16]; memcpy(s, in, 64);
        for (int r = 0; r < 6; ++r) {  // 6 double rounds = ChaCha12
            qr(s, 0, 4, 8, 12); qr(s, 1, 5, 9, 13); qr(s, 2, 6, 10, 14); qr(s, 3, 7, 11, 15);
            qr(s, 0, 5, 10, 15); qr(s, 1, 6, 11, 12); qr(s, 2, 7, 8, 13); qr(s, 3, 4, 9, 14);
        }
        for (int i = 0; i < 16; ++i) buf[i] = s[i] + in[i];
        counter++; idx = 0;
    }
    // SeedableRng::from_seed([u8;32]): key words little-endian.
    static StdRng from_seed(const uint8_t seed[32]) {
        StdRng r;
        for (int i = 0; i < 8; ++i)
            r.key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
        r.counter = 0; r.idx = 16;
        return r;
    }
    // rand_core 0.6.4 SeedableRng::seed_from_u64: PCG32 expander, 8 LE words.
    static StdRng seed_from_u64(uint64_t state) {
        uint8_t seed[32];
        for (int i = 0; i < 8; ++i) {
            state = state * 6364136223846793005ULL + 11634580027462260723ULL;
            uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
            seed[4 * i] = (uint8_t)x; seed[4 * i + 1] = (uint8_t)(x >> 8); seed[4 * i + 2] = (uint8_t)(x >> 16); seed[4 * i + 3] = (uint8_t)(x >> 24);
        }
        return from_seed(seed);
    }
    uint32_t next_u32() { if (idx >= 16) refill(); return buf[idx++]; }
    // BlockRng::next_u64 with an even index: lo word then hi word.  (Only u64 draws occur on this
    // path, so the index stays even and the odd-index branches of rand_core never trigger.)
    uint64_t next_u64() { uint64_t lo = next_u32(); uint64_t hi = next_u32(); return lo | (hi << 32); }
};

// ark-ff 0.5 `impl Distribution<Fp<P,N>> for Standard` (UniformRand::rand): fill the 4 limbs with
// next_u64 (limb 0 first), clear the top (256-255)=1 bit of limb 3, accept iff < r; the accepted
// limbs ARE the Montgomery representation (no conversion).  SURVEY.md Appendix A.
template <class F>
static inline F fr_rand(StdRng& rng) {
    for (;;) {
        F t;
        for (int i = 0; i < 4; ++i) t.l[i] = rng.next_u64();
        t.l[3] &= (~0ULL) >> 1;
        if (!F::geq_mod(t.l)) return t;
    }
}

}  // namespace oracle

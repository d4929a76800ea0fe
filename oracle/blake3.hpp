// oracle/blake3.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// From-scratch BLAKE3 (hash mode, 32-byte output), restating the published BLAKE3 spec; stands in
// for the un-vendored `blake3 1.8.2` crate (Cargo.lock:220) at the reference call site
// crates/utils/src/lib.rs:16-22 (`Hasher::new(); update(tag); update(data); finalize()`).
// Pinned by the public KATs checked in tests/test_oracle_primitives.py (empty input, 1-byte input).
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>
#include <vector>

namespace oracle {
namespace blake3 {

static const uint32_t IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A,
                               0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static const int MSG_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static inline void g(uint32_t* s, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    s[a] = s[a] + s[b] + mx; s[d] = rotr(s[d] ^ s[a], 16);
    s[c] = s[c] + s[d];      s[b] = rotr(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + my; s[d] = rotr(s[d] ^ s[a], 8);
    s[c] = s[c] + s[d];      s[b] = rotr(s[b] ^ s[c], 7);
}
static inline void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter,
                            uint32_t block_len, uint32_t flags, uint32_t out[16]) {
    uint32_t s[16], m[16];
    for (int i = 0; i < 8; ++i) s[i] = cv[i];
    for (int i = 0; i < 4; ++i) s[8 + i] = IV[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = block_len; s[15] = flags;
    memcpy(m, block, 64);
    for (int r = 0; r < 7; ++r) {
        g(s, 0, 4, 8, 12, m[0], m[1]);   g(s, 1, 5, 9, 13, m[2], m[3]);
        g(s, 2, 6, 10, 14, m[4], m[5]);  g(s, 3, 7, 11, 15, m[6], m[7]);
        g(s, 0, 5, 10, 15, m[8], m[9]);  g(s, 1, 6, 11, 12, m[10], m[11]);
        g(s, 2, 7, 8, 13, m[12], m[13]); g(s, 3, 4, 9, 14, m[14], m[15]);
        if (r < 6) { uint32_t p[16]; for (int i = 0; i < 16; ++i) p[i] = m[MSG_PERM[i]]; memcpy(m, p, 64); }
    }
    for (int i = 0; i < 8; ++i) { out[i] = s[i] ^ s[i + 8]; out[i + 8] = s[i + 8] ^ cv[i]; }
}
static inline void words_from_le(const uint8_t* b, size_t n, uint32_t w[16]) {
    uint8_t buf[64]; memset(buf, 0, 64); memcpy(buf, b, n);
    for (int i = 0; i < 16; ++i)
        w[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) | ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
}

// "Output" node: the pending final compression of a chunk or parent, so ROOT can be applied late.
struct Output { uint32_t cv[8]; uint32_t block[16]; uint64_t counter; uint32_t block_len; uint32_t flags; };
static inline void output_cv(const Output& o, uint32_t cv[8]) {
    uint32_t out[16]; compress(o.cv, o.block, o.counter, o.block_len, o.flags, out); memcpy(cv, out, 32);
}
static inline Output chunk_output(const uint8_t* data, size_t len, uint64_t chunk_counter) {
    uint32_t cv[8]; memcpy(cv, IV, 32);
    size_t nblocks = len == 0 ? 1 : (len + 63) / 64;
    Output o;
    for (size_t b = 0; b < nblocks; ++b) {
        size_t off = b * 64, bl = (len - off) < 64 ? (len - off) : 64;
        if (len == 0) bl = 0;
        uint32_t w[16]; words_from_le(data + off, bl, w);
        uint32_t flags = (b == 0 ? CHUNK_START : 0) | (b + 1 == nblocks ? CHUNK_END : 0);
        if (b + 1 == nblocks) {
            memcpy(o.cv, cv, 32); memcpy(o.block, w, 64); o.counter = chunk_counter; o.block_len = (uint32_t)bl; o.flags = flags;
        } else {
            uint32_t out[16]; compress(cv, w, chunk_counter, 64, flags, out); memcpy(cv, out, 32);
        }
    }
    return o;
}
static inline Output parent_output(const uint32_t l[8], const uint32_t r[8]) {
    Output o; memcpy(o.cv, IV, 32); memcpy(o.block, l, 32); memcpy(o.block + 8, r, 32);
    o.counter = 0; o.block_len = 64; o.flags = PARENT; return o;
}

// One-shot hash of `len` bytes → 32 bytes.
static inline void hash(const uint8_t* data, size_t len, uint8_t out32[32]) {
    const size_t CHUNK = 1024;
    size_t nchunks = len == 0 ? 1 : (len + CHUNK - 1) / CHUNK;
    std::vector<std::vector<uint32_t>> stack;  // CV stack (8 words each)
    Output last;
    for (size_t c = 0; c < nchunks; ++c) {
        size_t off = c * CHUNK, cl = (len - off) < CHUNK ? (len - off) : CHUNK;
        Output o = chunk_output(data + off, cl, c);
        if (c + 1 == nchunks) { last = o; break; }
        uint32_t cv[8]; output_cv(o, cv);
        // merge completed subtrees: one merge per trailing zero bit of the new total chunk count
        uint64_t total = c + 1;
        std::vector<uint32_t> cur(cv, cv + 8);
        while ((total & 1) == 0) {
            std::vector<uint32_t> left = stack.back(); stack.pop_back();
            Output p = parent_output(left.data(), cur.data());
            uint32_t pcv[8]; output_cv(p, pcv); cur.assign(pcv, pcv + 8);
            total >>= 1;
        }
        stack.push_back(cur);
    }
    // fold the stack right-to-left into the root
    Output o = last;
    while (!stack.empty()) {
        uint32_t cv[8]; output_cv(o, cv);
        std::vector<uint32_t> left = stack.back(); stack.pop_back();
        o = parent_output(left.data(), cv);
    }
    uint32_t outw[16]; compress(o.cv, o.block, 0 /* root output block 0 */, o.block_len, o.flags | ROOT, outw);
    // NB: for a root that is a single chunk the counter is the chunk counter (0) — identical.
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) out32[4 * i + j] = (uint8_t)(outw[i] >> (8 * j));
}

}  // namespace blake3
}  // namespace oracle

// stark_mlwe_amd/csrc/host_util.hpp — host-side scalar logic of the product (no device code).
//
// What a Rust host would get from `blake3`, `rand::StdRng` and arkworks when it drives the C-ABI:
//   * BLAKE3 (hash mode)            -> Poseidon constant derivation (utils/src/lib.rs:16-22)
//   * ChaCha12 block function        -> `StdRng::from_seed` + `gen::<u64>()` (fri.rs:66-71,185-186)
//   * from_le_bytes_mod_order, tags  -> transcript framing constants (transcript/src/lib.rs:13-41)
//   * constant tables for the kernels: LU factors of the MDS matrix and the sparse factorisation of
//     the partial rounds (algebraically identical to the dense rounds of poseidon/src/lib.rs:31-68).
// Written independently of oracle/ (which is test infrastructure and is never linked here).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "fr.hpp"
#include "fr29.hpp"

namespace stark {
namespace host {

// ------------------------------------------------------------------------------------------------
// BLAKE3, hash mode, arbitrary length, 32-byte digest (spec: chunks of 1024 B, blocks of 64 B,
// binary tree of chaining values).  Compact recursive formulation.
// ------------------------------------------------------------------------------------------------
struct Blake3 {
    static constexpr uint32_t kIV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
    enum : uint32_t { F_CHUNK_START = 1, F_CHUNK_END = 2, F_PARENT = 4, F_ROOT = 8 };
    static uint32_t ror(uint32_t x, unsigned n) { return (x >> n) | (x << (32 - n)); }
    static void mix(uint32_t* v, int a, int b, int c, int d, uint32_t x, uint32_t y) {
        v[a] += v[b] + x; v[d] = ror(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = ror(v[b] ^ v[c], 12);
        v[a] += v[b] + y; v[d] = ror(v[d] ^ v[a], 8);  v[c] += v[d]; v[b] = ror(v[b] ^ v[c], 7);
    }
    // full 16-word compression output
    static void compress(const uint32_t h[8], const uint8_t block[64], uint32_t blen, uint64_t ctr, uint32_t flags, uint32_t out[16]) {
        static const uint8_t sched[7][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8},
            {3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1}, {10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6},
            {12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4}, {9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7},
            {11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13}};
        uint32_t m[16], v[16];
        for (int i = 0; i < 16; ++i) m[i] = (uint32_t)block[4 * i] | (uint32_t)block[4 * i + 1] << 8 | (uint32_t)block[4 * i + 2] << 16 | (uint32_t)block[4 * i + 3] << 24;
        for (int i = 0; i < 8; ++i) v[i] = h[i];
        for (int i = 0; i < 4; ++i) v[8 + i] = kIV[i];
        v[12] = (uint32_t)ctr; v[13] = (uint32_t)(ctr >> 32); v[14] = blen; v[15] = flags;
        for (int r = 0; r < 7; ++r) {
            const uint8_t* s = sched[r];
            mix(v, 0, 4, 8, 12, m[s[0]], m[s[1]]);   mix(v, 1, 5, 9, 13, m[s[2]], m[s[3]]);
            mix(v, 2, 6, 10, 14, m[s[4]], m[s[5]]);  mix(v, 3, 7, 11, 15, m[s[6]], m[s[7]]);
            mix(v, 0, 5, 10, 15, m[s[8]], m[s[9]]);  mix(v, 1, 6, 11, 12, m[s[10]], m[s[11]]);
            mix(v, 2, 7, 8, 13, m[s[12]], m[s[13]]); mix(v, 3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; ++i) { out[i] = v[i] ^ v[i + 8]; out[i + 8] = v[i + 8] ^ h[i]; }
    }
    struct Node { uint32_t h[8]; uint8_t block[64]; uint32_t blen; uint64_t ctr; uint32_t flags; };
    static void cv_of(const Node& n, uint32_t extra, uint32_t cv[8]) { uint32_t o[16]; compress(n.h, n.block, n.blen, n.ctr, n.flags | extra, o); memcpy(cv, o, 32); }
    static Node chunk_node(const uint8_t* p, size_t len, uint64_t index) {
        Node n; memcpy(n.h, kIV, 32); n.ctr = index;
        size_t off = 0; bool first = true;
        while (len - off > 64) {
            uint32_t o[16]; compress(n.h, p + off, 64, index, first ? F_CHUNK_START : 0, o); memcpy(n.h, o, 32);
            off += 64; first = false;
        }
        memset(n.block, 0, 64); memcpy(n.block, p + off, len - off); n.blen = (uint32_t)(len - off);
        n.flags = (first ? F_CHUNK_START : 0) | F_CHUNK_END;
        return n;
    }
    // node covering p[0..len) whose first chunk has index `index` (left subtree = largest power of two of chunks < total)
    static Node subtree(const uint8_t* p, size_t len, uint64_t index) {
        if (len <= 1024) return chunk_node(p, len, index);
        size_t chunks = (len + 1023) / 1024, left = 1; while (left * 2 < chunks) left *= 2;
        Node l = subtree(p, left * 1024, index), r = subtree(p + left * 1024, len - left * 1024, index + left);
        Node n; memcpy(n.h, kIV, 32); n.ctr = 0; n.blen = 64; n.flags = F_PARENT;
        uint32_t a[8], b[8]; cv_of(l, 0, a); cv_of(r, 0, b);
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) { n.block[4 * i + j] = (uint8_t)(a[i] >> (8 * j)); n.block[32 + 4 * i + j] = (uint8_t)(b[i] >> (8 * j)); }
        return n;
    }
    static void hash(const uint8_t* p, size_t len, uint8_t out[32]) {
        Node root = subtree(p, len, 0);
        if (root.flags & F_PARENT) root.ctr = 0;
        uint32_t cv[8]; cv_of(root, F_ROOT, cv);
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(cv[i] >> (8 * j));
    }
};

// ------------------------------------------------------------------------------------------------
// ChaCha12 keystream as consumed by rand 0.8.5 `StdRng` (64-bit block counter, zero stream id).
// ------------------------------------------------------------------------------------------------
struct ChaCha12Rng {
    uint32_t st[16]; uint32_t out[16]; unsigned have = 0;
    explicit ChaCha12Rng(const uint8_t seed[32]) {
        st[0] = 0x61707865u; st[1] = 0x3320646eu; st[2] = 0x79622d32u; st[3] = 0x6b206574u;
        for (int i = 0; i < 8; ++i) st[4 + i] = (uint32_t)seed[4 * i] | (uint32_t)seed[4 * i + 1] << 8 | (uint32_t)seed[4 * i + 2] << 16 | (uint32_t)seed[4 * i + 3] << 24;
        st[12] = st[13] = st[14] = st[15] = 0;
    }
    static uint32_t rol(uint32_t x, unsigned n) { return (x << n) | (x >> (32 - n)); }
    void block() {
        uint32_t x[16]; memcpy(x, st, 64);
        auto Q = [&](int a, int b, int c, int d) {
            x[a] += x[b]; x[d] = rol(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rol(x[b] ^ x[c], 12);
            x[a] += x[b]; x[d] = rol(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rol(x[b] ^ x[c], 7); };
        for (int i = 0; i < 6; ++i) { Q(0, 4, 8, 12); Q(1, 5, 9, 13); Q(2, 6, 10, 14); Q(3, 7, 11, 15); Q(0, 5, 10, 15); Q(1, 6, 11, 12); Q(2, 7, 8, 13); Q(3, 4, 9, 14); }
        for (int i = 0; i < 16; ++i) out[i] = x[i] + st[i];
        if (++st[12] == 0) ++st[13];
        have = 16;
    }
    uint32_t next_u32() { if (!have) block(); return out[16 - have--]; }
    uint64_t next_u64() { uint64_t lo = next_u32(), hi = next_u32(); return lo | hi << 32; }   // gen::<u64>()
};

// ------------------------------------------------------------------------------------------------
// Host field helpers on fr_t (Pallas Fr is the prover field).
// ------------------------------------------------------------------------------------------------
typedef PallasFr PF;
inline fr_t h_one() { return fr_one<PF>(); }
inline fr_t h_zero() { return fr_zero<PF>(); }
inline fr_t h_mul(const fr_t& a, const fr_t& b) { return fr_mul<PF>(a, b); }
inline fr_t h_add(const fr_t& a, const fr_t& b) { return fr_add<PF>(a, b); }
inline fr_t h_sub(const fr_t& a, const fr_t& b) { return fr_sub<PF>(a, b); }
inline fr_t h_u64(uint64_t x) { return fr_from_u64<PF>(x); }
// int_le(bytes[0..n)) mod r, n <= 64, as a Montgomery element (F::from_le_bytes_mod_order).
inline fr_t h_from_le_bytes_mod_order(const uint8_t* b, size_t n) {
    // Horner over 32-bit words, most significant first: acc = acc * 2^32 + w.
    fr_t acc = h_zero(); const fr_t two32 = h_u64(1ull << 32);
    size_t nw = (n + 3) / 4;
    for (size_t k = nw; k-- > 0;) {
        uint32_t w = 0; for (size_t j = 0; j < 4; ++j) if (4 * k + j < n) w |= (uint32_t)b[4 * k + j] << (8 * j);
        acc = h_add(h_mul(acc, two32), h_u64(w));
    }
    return acc;
}
// serialize_uncompressed / serialize_compressed of Fp: 32-byte LE canonical.
inline void h_to_bytes_le(const fr_t& a, uint8_t out[32]) {
    fr_t c = fr_to_canonical<PF>(a);
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(c.v[i] >> (8 * j));
}
// utils/src/lib.rs:16-22  fr_from_hash(tag, data).
inline fr_t h_fr_from_hash(const char* tag, const std::vector<uint8_t>& data) {
    std::vector<uint8_t> buf(tag, tag + strlen(tag)); buf.insert(buf.end(), data.begin(), data.end());
    uint8_t d[32]; Blake3::hash(buf.data(), buf.size(), d);
    uint8_t wide[64]; memset(wide, 0, 64); memcpy(wide, d, 32);            // utils/src/lib.rs:7-13
    return h_from_le_bytes_mod_order(wide, 64);
}
// transcript/src/lib.rs:13-29  domain_tag_to_field  /  :32-41 bytes_to_field_words.
inline fr_t h_tag(const char* s) {
    size_t n = strlen(s);
    if (n <= 32) { uint8_t le[32] = {0}; memcpy(le, s, n); return h_from_le_bytes_mod_order(le, 32); }
    fr_t acc = h_zero();
    for (size_t off = 0; off < n; off += 32) { uint8_t le[32] = {0}; size_t c = n - off < 32 ? n - off : 32; memcpy(le, s + off, c); acc = h_add(acc, h_from_le_bytes_mod_order(le, 32)); }
    return acc;
}
inline std::vector<fr_t> h_words(const char* s) {
    std::vector<fr_t> out; size_t n = strlen(s);
    for (size_t off = 0; off < n; off += 31) { uint8_t le[32] = {0}; size_t c = n - off < 31 ? n - off : 31; memcpy(le, s + off, c); out.push_back(h_from_le_bytes_mod_order(le, 32)); }
    return out;
}

// ------------------------------------------------------------------------------------------------
// Poseidon constants in the reference's form (poseidon/src/lib.rs:104-114, 176-216, 318-356).
// ------------------------------------------------------------------------------------------------
struct PoseidonConsts {
    int t = 0, rf = 0, rp = 0;
    std::vector<fr_t> mds;         // t*t row-major  mds[i*t+j]
    std::vector<fr_t> rc_full;     // rf*t
    std::vector<fr_t> rc_partial;  // rp
};
inline void put64(std::vector<uint8_t>& v, uint64_t x) { for (int j = 0; j < 8; ++j) v.push_back((uint8_t)(x >> (8 * j))); }
inline PoseidonConsts derive_consts(const std::string& seed, int t, int rf, int rp) {
    PoseidonConsts c; c.t = t; c.rf = rf; c.rp = rp;
    c.mds.resize((size_t)t * t); c.rc_full.resize((size_t)rf * t); c.rc_partial.resize(rp);
    for (int i = 0; i < t; ++i) for (int j = 0; j < t; ++j) { std::vector<uint8_t> d; put64(d, i); put64(d, j); d.insert(d.end(), seed.begin(), seed.end()); c.mds[(size_t)i * t + j] = h_fr_from_hash("POSEIDON-MDS", d); }
    for (int r = 0; r < rf; ++r) for (int i = 0; i < t; ++i) { std::vector<uint8_t> d; put64(d, r); put64(d, i); d.insert(d.end(), seed.begin(), seed.end()); c.rc_full[(size_t)r * t + i] = h_fr_from_hash("POSEIDON-RC-FULL", d); }
    for (int r = 0; r < rp; ++r) { std::vector<uint8_t> d; put64(d, r); d.insert(d.end(), seed.begin(), seed.end()); c.rc_partial[r] = h_fr_from_hash("POSEIDON-RC-PART", d); }
    return c;
}
inline int width_for_arity(size_t arity) { return arity <= 8 ? 9 : arity <= 16 ? 17 : arity <= 32 ? 33 : arity <= 64 ? 65 : arity <= 128 ? 129 : -1; }   // poseidon/src/lib.rs:155-166
inline int rp_for_width(int t) { switch (t) { case 9: return 60; case 17: return 64; case 33: return 68; case 65: return 76; case 129: return 84; default: return -1; } }  // :121-128
inline PoseidonConsts consts_for_width(int t) {          // poseidon_params_for_width + seed_for_t (:120-146,168-174)
    std::string seed = "POSEIDON-PALLAS-T"; for (int j = 0; j < 8; ++j) seed.push_back((char)(((uint64_t)t >> (8 * j)) & 0xff));
    return derive_consts(seed, t, 8, rp_for_width(t));
}
inline PoseidonConsts consts_transcript() { return derive_consts("POSEIDON-T17-X5-TRANSCRIPT", 17, 8, 64); }   // transcript/src/lib.rs:44-46

// ------------------------------------------------------------------------------------------------
// Kernel-form constants: in-place LU factors and the sparse partial-round factorisation.
//
// Partial rounds  s <- M * S_r(s)  (S_r touches lane 0 only) are rewritten as
//     s <- A_R S_R  A_{R-1} S_{R-1} ... A_1 S_1  (B_1 s)
// with M_R = M, M_k = A_k B_k, B_k = diag(1, Mhat_k), A_k = [[m00, v Mhat_k^-1], [w, I]],
// M_{k-1} = B_k M.  B_1 is merged into the MDS of the last first-half full round (mds_pre = B_1 M).
// B_k commutes with S_k, so the composition is unchanged: exact field arithmetic => identical bits.
// A dense matrix is applied in place as L*(U*x): U top-down, then unit-lower L bottom-up.
// ------------------------------------------------------------------------------------------------
struct KernelConsts {
    int t = 0, rf = 0, rp = 0;
    std::vector<fr_t> rc_full, rc_partial;
    std::vector<fr_t> lu, lu_pre;       // t*t each: U on/above the diagonal, strict lower = L (unit diagonal implied)
    std::vector<fr_t> row0;             // t: mds[0][*] (final round when only lane 0 is squeezed)
    std::vector<fr_t> sparse;           // rp*(2t-1): [a, u_1..u_{t-1}, w_1..w_{t-1}] in application order
    std::vector<fr_t> mds;              // t*t reference form (cooperative kernel: full rounds)
    std::vector<fr_t> mds_pre;          // t*t dense B_1*M (cooperative kernel: last first-half full round)
    std::vector<fr_t> gamma;            // (rp/4)*6: blocks of 4 partial rounds, gamma[q(q-1)/2+p] = sum_j u_{q,j} w_{p,j} (p < q)
    // the multiplier tables of every dot product again, as nine 29-bit limbs of c*2^261 mod r per entry (fr29.hpp)
    std::vector<uint32_t> lu29, lu_pre29, row0_29, sparse29, gamma29;
    std::vector<uint32_t> mds29, mds_pre29;   // dense forms for the one-wave kernel (every entry meets an S-box output: all scaled)
    // t = 17 only: the dense matrices as int8 MFMA operand fragments (signed radix-256 digits, Toeplitz windows), see mfma_frags
    std::vector<int8_t> mds_frag, mds_pre_frag;
    // t = 17 only: the partial rounds UNROLLED over all rp rounds for the five-wave latency kernel (poseidon_chain.hpp):
    //     X_{q+1} = c_{q+1} + sum_j u_{q,j} s_j^(0) + a_q y_q + sum_{p<q} Gamma_{q,p} y_p,   y_q = X_q^5,   Gamma_{q,p} = sum_j u_{q,j} w_{p,j}
    // Every entry multiplies a y that reaches it through THREE Montgomery steps by 2^261 on operands carrying 2^256, so it is stored as
    // nine 29-bit limbs of c * 2^25 * 2^256 mod r (= c * (2^261)^5 / (2^256)^4): the product lands back in the 2^256 domain.
    //   chain_a [rp][64]      lane 16 + c: limb c of a_q; lane 32 + c: limb c of Gamma_{q+1,q} (0 for the last round); other lanes 0
    //   chain_g [rp][9][64]   limb i of Gamma_{q,p} for lane q >= p + 2, else 0   (p = first index)
    //   chain_w [rp][t-1][9]  w_{p,j}
    std::vector<uint32_t> chain_a, chain_g, chain_w;
    bool ok = false;
};
// scaled(i) == true: the entry multiplies an S-box output, which fr_pow5_r29 delivers as x^5 / 2^20 (fr29.hpp)
template <class Pred> inline std::vector<uint32_t> to_radix29(const std::vector<fr_t>& v, Pred scaled) {
    const fr_t k = fr_from_u64<PF>(1ull << FR29_SBOX_SHIFT);
    std::vector<uint32_t> o(v.size() * 9);
    for (size_t i = 0; i < v.size(); ++i) fr29_const_from<PF>(scaled(i) ? fr_mul<PF>(v[i], k) : v[i], &o[9 * i]);
    return o;
}
// A-operand fragments of v_mfma_i32_32x32x32_i8 for y = M * x with x an S-box output (poseidon_pair.hpp pair_apply_mds_mfma).
// Entry (i, e) is c = M[i][e] * 2^20 * 32 in Montgomery form (2^20: fr_pow5_r29's scale; 32: the Montgomery step by 2^261 instead of 2^256),
// written in SIGNED radix-256 digits d[0..31] in [-128, 127] (add 0x80 to every byte with carries, subtract 0x80 from every digit; c < r
// keeps the top byte below 0x80, so no 33rd digit).  The product's digit-column sums are S[c] = sum_{e,b} d_ie[c - b] * xd_e[b]: row c of
// a Toeplitz matrix.  Fragment ((i*2 + rt)*t + e) holds, for lane l (tile row r = l & 31, k half kh = l >> 5), the 16 bytes
// j -> d_ie[(32 rt + r) - (16 kh + j)] (zero outside 0..31): the left operand of the K-step of element e for the output digits 32 rt .. 32 rt + 31.
inline std::vector<int8_t> mfma_frags(const std::vector<fr_t>& mat, int t) {
    std::vector<int8_t> out((size_t)t * 2 * t * 64 * 16, 0);
    const fr_t scale = h_mul(fr_from_u64<PF>(1ull << FR29_SBOX_SHIFT), fr_from_u64<PF>(32));
    for (int i = 0; i < t; ++i) for (int e = 0; e < t; ++e) {
        const fr_t c = h_mul(mat[(size_t)i * t + e], scale);
        int8_t d[32]; int cy = 0;
        for (int b = 0; b < 32; ++b) { const int v = (int)((c.v[b >> 2] >> (8 * (b & 3))) & 0xff) + 0x80 + cy; cy = v >> 8; d[b] = (int8_t)((v & 0xff) - 0x80); }
        for (int rt = 0; rt < 2; ++rt) for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) {
            const int idx = (32 * rt + (l & 31)) - (16 * (l >> 5) + j);
            out[(((size_t)(i * 2 + rt) * t + e) * 64 + l) * 16 + j] = (idx >= 0 && idx < 32) ? d[idx] : (int8_t)0;
        }
    }
    return out;
}
// Gauss-Jordan inverse of an n x n matrix (row-major); returns false when singular.
inline bool mat_inverse(std::vector<fr_t> a, int n, std::vector<fr_t>& inv) {
    inv.assign((size_t)n * n, h_zero()); for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = h_one();
    for (int c = 0; c < n; ++c) {
        int p = -1; for (int r = c; r < n; ++r) if (!fr_is_zero(a[(size_t)r * n + c])) { p = r; break; }
        if (p < 0) return false;
        if (p != c) for (int k = 0; k < n; ++k) { std::swap(a[(size_t)p * n + k], a[(size_t)c * n + k]); std::swap(inv[(size_t)p * n + k], inv[(size_t)c * n + k]); }
        fr_t pi = fr_inv<PF>(a[(size_t)c * n + c]);
        for (int k = 0; k < n; ++k) { a[(size_t)c * n + k] = h_mul(a[(size_t)c * n + k], pi); inv[(size_t)c * n + k] = h_mul(inv[(size_t)c * n + k], pi); }
        for (int r = 0; r < n; ++r) if (r != c) {
            fr_t f = a[(size_t)r * n + c]; if (fr_is_zero(f)) continue;
            for (int k = 0; k < n; ++k) { a[(size_t)r * n + k] = h_sub(a[(size_t)r * n + k], h_mul(f, a[(size_t)c * n + k])); inv[(size_t)r * n + k] = h_sub(inv[(size_t)r * n + k], h_mul(f, inv[(size_t)c * n + k])); }
        }
    }
    return true;
}
// Doolittle LU without pivoting, packed in place; false on a zero pivot.
inline bool lu_pack(const std::vector<fr_t>& m, int n, std::vector<fr_t>& lu) {
    lu = m;
    for (int k = 0; k < n; ++k) {
        if (fr_is_zero(lu[(size_t)k * n + k])) return false;
        fr_t pi = fr_inv<PF>(lu[(size_t)k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            fr_t f = h_mul(lu[(size_t)i * n + k], pi); lu[(size_t)i * n + k] = f;
            for (int j = k + 1; j < n; ++j) lu[(size_t)i * n + j] = h_sub(lu[(size_t)i * n + j], h_mul(f, lu[(size_t)k * n + j]));
        }
    }
    return true;
}
inline KernelConsts make_kernel_consts(const PoseidonConsts& c) {
    KernelConsts k; k.t = c.t; k.rf = c.rf; k.rp = c.rp; k.rc_full = c.rc_full; k.rc_partial = c.rc_partial; k.mds = c.mds;
    const int t = c.t, n = t - 1;
    k.row0.assign(c.mds.begin(), c.mds.begin() + t);
    std::vector<fr_t> cur = c.mds;                    // M_R
    k.sparse.assign((size_t)c.rp * (2 * t - 1), h_zero());
    for (int r = c.rp - 1; r >= 0; --r) {
        std::vector<fr_t> hat((size_t)n * n), hinv;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) hat[(size_t)i * n + j] = cur[(size_t)(i + 1) * t + (j + 1)];
        if (!mat_inverse(hat, n, hinv)) return k;
        fr_t* sp = &k.sparse[(size_t)r * (2 * t - 1)];
        sp[0] = cur[0];
        for (int j = 0; j < n; ++j) { fr_t acc = h_zero(); for (int q = 0; q < n; ++q) acc = h_add(acc, h_mul(cur[(size_t)0 * t + (q + 1)], hinv[(size_t)q * n + j])); sp[1 + j] = acc; }   // u = v * hat^-1
        for (int i = 0; i < n; ++i) sp[t + i] = cur[(size_t)(i + 1) * t + 0];                                                                                                        // w
        // next = B * M : row 0 of M unchanged; rows 1.. = hat * M[1:, :]
        std::vector<fr_t> nxt((size_t)t * t);
        for (int j = 0; j < t; ++j) nxt[j] = c.mds[j];
        for (int i = 0; i < n; ++i) for (int j = 0; j < t; ++j) { fr_t acc = h_zero(); for (int q = 0; q < n; ++q) acc = h_add(acc, h_mul(hat[(size_t)i * n + q], c.mds[(size_t)(q + 1) * t + j])); nxt[(size_t)(i + 1) * t + j] = acc; }
        cur.swap(nxt);
    }
    // cur == B_1 * M == mds_pre
    // Blocks of 4 partial rounds: within a block the lanes 1..t-1 are read at their block-start value and
    // the S-box outputs x_p enter later rounds through gamma_{q,p} = sum_j u_{q,j} * w_{p,j}  (p < q).
    if (c.rp % 4) return k;
    k.gamma.assign((size_t)(c.rp / 4) * 6, h_zero());
    for (int b = 0; b < c.rp / 4; ++b)
        for (int q = 1; q < 4; ++q)
            for (int p2 = 0; p2 < q; ++p2) {
                const fr_t* uq = &k.sparse[(size_t)(4 * b + q) * (2 * t - 1)];
                const fr_t* wp = &k.sparse[(size_t)(4 * b + p2) * (2 * t - 1)];
                fr_t acc = h_zero();
                for (int j = 1; j < t; ++j) acc = h_add(acc, h_mul(uq[j], wp[t - 1 + j]));
                k.gamma[(size_t)b * 6 + q * (q - 1) / 2 + p2] = acc;
            }
    if (!lu_pack(c.mds, t, k.lu)) return k;
    if (!lu_pack(cur, t, k.lu_pre)) return k;
    k.mds_pre = cur;
    // radix-2^29 tables.  Scaled by 2^20: what meets an S-box output — U of every L*U (entries on/above the diagonal;
    // L acts on U's result), row 0 of M, a_k and w_k of the sparse rounds (u_k meets the lanes, not an S-box output), gamma.
    const auto upper = [t](size_t i) { return (int)(i % t) >= (int)(i / t); };
    const auto all = [](size_t) { return true; };
    const auto a_and_w = [t](size_t i) { const int c = (int)(i % (2 * t - 1)); return c == 0 || c >= t; };
    k.lu29 = to_radix29(k.lu, upper); k.lu_pre29 = to_radix29(k.lu_pre, upper); k.row0_29 = to_radix29(k.row0, all);
    k.sparse29 = to_radix29(k.sparse, a_and_w); k.gamma29 = to_radix29(k.gamma, all);
    k.mds29 = to_radix29(k.mds, all); k.mds_pre29 = to_radix29(k.mds_pre, all);
    if (t == 17) { k.mds_frag = mfma_frags(k.mds, t); k.mds_pre_frag = mfma_frags(k.mds_pre, t); }      // the wave-pair kernels' full rounds (poseidon_pair.hpp)
    if (t == 17 && c.rp == 64) {                                                                         // the five-wave latency kernel (poseidon_chain.hpp)
        const int rp = c.rp, w = 2 * t - 1;
        const fr_t scale = fr_from_u64<PF>(1ull << 25);
        auto limbs = [&](const fr_t& v, uint32_t* out) { const fr29_t u = fr29_unpack(h_mul(v, scale)); for (int i = 0; i < 9; ++i) out[i] = u.l[i]; };
        k.chain_a.assign((size_t)rp * 64, 0u); k.chain_g.assign((size_t)rp * 9 * 64, 0u); k.chain_w.assign((size_t)rp * (t - 1) * 9, 0u);
        for (int q = 0; q < rp; ++q) {
            uint32_t l[9]; limbs(k.sparse[(size_t)q * w], l);
            for (int i = 0; i < 9; ++i) k.chain_a[(size_t)q * 64 + 16 + i] = l[i];
            for (int j = 1; j < t; ++j) limbs(k.sparse[(size_t)q * w + t - 1 + j], &k.chain_w[((size_t)q * (t - 1) + (j - 1)) * 9]);
        }
        for (int q = 1; q < rp; ++q)
            for (int p2 = 0; p2 < q; ++p2) {
                const fr_t* uq = &k.sparse[(size_t)q * w]; const fr_t* wp = &k.sparse[(size_t)p2 * w];
                fr_t acc = h_zero();
                for (int j = 1; j < t; ++j) acc = h_add(acc, h_mul(uq[j], wp[t - 1 + j]));
                uint32_t l[9]; limbs(acc, l);
                if (q == p2 + 1) { for (int i = 0; i < 9; ++i) k.chain_a[(size_t)p2 * 64 + 32 + i] = l[i]; }             // Gamma_{p+1,p}: wave A's third row in round p
                else for (int i = 0; i < 9; ++i) k.chain_g[((size_t)p2 * 9 + i) * 64 + q] = l[i];                          // q >= p + 2: wave B, lane q
            }
    }
    k.ok = true;
    return k;
}

// Reference-form permutation on the host (dense MDS every round; poseidon/src/lib.rs:31-68,219-258).
// Used only by the host-check library to validate the kernel-form constants on CPU.
inline void permute_dense(fr_t* s, const PoseidonConsts& c) {
    const int t = c.t, half = c.rf / 2; std::vector<fr_t> o(t);
    auto mds = [&]() { for (int i = 0; i < t; ++i) { fr_t acc = h_zero(); for (int j = 0; j < t; ++j) acc = h_add(acc, h_mul(c.mds[(size_t)i * t + j], s[j])); o[i] = acc; } for (int i = 0; i < t; ++i) s[i] = o[i]; };
    for (int r = 0; r < half; ++r) { for (int i = 0; i < t; ++i) s[i] = fr_pow5<PF>(h_add(s[i], c.rc_full[(size_t)r * t + i])); mds(); }
    for (int r = 0; r < c.rp; ++r) { s[0] = fr_pow5<PF>(h_add(s[0], c.rc_partial[r])); mds(); }
    for (int r = half; r < c.rf; ++r) { for (int i = 0; i < t; ++i) s[i] = fr_pow5<PF>(h_add(s[i], c.rc_full[(size_t)r * t + i])); mds(); }
}

}  // namespace host
}  // namespace stark

// stark_mlwe_amd/csrc/hostcheck.cpp — DIAGNOSTIC library (libstark_mlwe_hostcheck.so), CPU-only.
//
// Instantiates, on the host, the SAME inline code the kernels are built from (fr.hpp arithmetic,
// host_util.hpp constant derivation, poseidon_dev.hpp permutation / sponge bodies with a plain-array
// state) so that `pytest -m "not gpu"` can check the product's host logic and kernel bodies against
// the oracle without a GPU.  No product entry point loads or calls this library; it is not a
// fallback: libstark_mlwe_hip.so fails with STARK_ERR_HIP when no device is present.
#include <cstring>
#include <vector>
#include "fr.hpp"
#include "host_util.hpp"
#include "poseidon_dev.hpp"
#include "fri_plan.hpp"
#include "fri_verify.hpp"
#include "mfma_digits.hpp"
#include "poseidon_chain.hpp"

using namespace stark;

static fr_t ld4(const uint64_t* p) { fr_t x; for (int i = 0; i < 4; ++i) { x.v[2 * i] = (uint32_t)p[i]; x.v[2 * i + 1] = (uint32_t)(p[i] >> 32); } return x; }
static void st4(uint64_t* p, const fr_t& x) { for (int i = 0; i < 4; ++i) p[i] = (uint64_t)x.v[2 * i] | ((uint64_t)x.v[2 * i + 1] << 32); }

struct HcParams { host::PoseidonConsts ref; host::KernelConsts kc; PoseidonDev dev; };
static void bind(HcParams* P) {
    P->kc = host::make_kernel_consts(P->ref);
    P->dev.t = P->kc.t; P->dev.rf = P->kc.rf; P->dev.rp = P->kc.rp; P->dev.rc_full = P->kc.rc_full.data(); P->dev.rc_partial = P->kc.rc_partial.data();
    P->dev.lu = P->kc.lu.data(); P->dev.lu_pre = P->kc.lu_pre.data(); P->dev.row0 = P->kc.row0.data(); P->dev.sparse = P->kc.sparse.data(); P->dev.mds = P->kc.mds.data(); P->dev.mds_pre = P->kc.mds_pre.data(); P->dev.gamma = P->kc.gamma.data();
    P->dev.lu29 = P->kc.lu29.data(); P->dev.lu_pre29 = P->kc.lu_pre29.data(); P->dev.row0_29 = P->kc.row0_29.data(); P->dev.sparse29 = P->kc.sparse29.data(); P->dev.gamma29 = P->kc.gamma29.data(); P->dev.mds29 = P->kc.mds29.data(); P->dev.mds_pre29 = P->kc.mds_pre29.data();
    P->dev.mds_frag = nullptr; P->dev.mds_pre_frag = nullptr;
    P->dev.chain_a = P->kc.chain_a.empty() ? nullptr : P->kc.chain_a.data(); P->dev.chain_g = P->kc.chain_g.empty() ? nullptr : P->kc.chain_g.data(); P->dev.chain_w = P->kc.chain_w.empty() ? nullptr : P->kc.chain_w.data();
}

extern "C" {

// field: 0 Pallas, 1 BLS12-381.  op: 0 add, 1 sub, 2 mul, 3 inv, 4 from_u64(a[0]), 5 to_canonical, 6 root_of_unity(a[0]), 7 pow_u64(a, b[0])
int hc_fr_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    fr_t x = ld4(a), y = b ? ld4(b) : x, z;
    if (field == 0) {
        switch (op) { case 0: z = fr_add<PallasFr>(x, y); break; case 1: z = fr_sub<PallasFr>(x, y); break; case 2: z = fr_mul<PallasFr>(x, y); break; case 3: z = fr_inv<PallasFr>(x); break;
            case 4: z = fr_from_u64<PallasFr>(a[0]); break; case 5: z = fr_to_canonical<PallasFr>(x); break; case 6: z = fr_root_of_unity<PallasFr>((unsigned)a[0]); break;
            case 7: z = fr_pow_u64<PallasFr>(x, b[0]); break; default: return -1; }
    } else {
        switch (op) { case 0: z = fr_add<Bls12381Fr>(x, y); break; case 1: z = fr_sub<Bls12381Fr>(x, y); break; case 2: z = fr_mul<Bls12381Fr>(x, y); break; case 3: z = fr_inv<Bls12381Fr>(x); break;
            case 4: z = fr_from_u64<Bls12381Fr>(a[0]); break; case 5: z = fr_to_canonical<Bls12381Fr>(x); break; case 6: z = fr_root_of_unity<Bls12381Fr>((unsigned)a[0]); break;
            case 7: z = fr_pow_u64<Bls12381Fr>(x, b[0]); break; default: return -1; }
    }
    st4(out, z); return 0;
}
// sum_i a_i*b_i through the product's dot-product accumulator (DotAcc: radix-2^29 constants, carry pass every 6 terms,
// chunks of 60) and, for comparison, through the radix-2^32 wide accumulator the cooperative kernels use (mode 1)
int hc_wide_dot(const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
    DotAcc d; d.init();
    for (size_t i = 0; i < n; ++i) { uint32_t c[9]; fr29_const_from<PF>(ld4(a + 4 * i), c); d.mac(c, ld4(b + 4 * i)); }
    st4(out, d.finish()); return 0;
}
int hc_wide_dot32(const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
    if (n > 24) return -1;
    fr_wide w; fr_wide_zero(w);
    for (size_t i = 0; i < n; ++i) fr_wide_mac_f<PF>(w, ld4(a + 4 * i), ld4(b + 4 * i));
    st4(out, fr_wide_reduce<PF>(w)); return 0;
}
}  // extern "C"
// One size-2^log_b sub-NTT exactly as a tile of ntt_dev.hpp computes it (nine-limb lazy decimation-in-time butterflies, the same carry-pass
// schedule, tables carrying the factor 32, the multiply-by-32 epilogue), sequentially on the host.  data: 2^log_b canonical Montgomery values,
// natural order in and out.  Also reports the largest limb / top limb any operand of a product reached (bounds check for the tests).
template <class F> static int hc_ntt29_impl(uint64_t* data, int log_b, int inverse, uint64_t* max_limb, uint64_t* max_top) {
    if (log_b < 1 || log_b > 16) return -1;
    const size_t B = (size_t)1 << log_b;
    fr_t w = fr_root_of_unity<F>((unsigned)log_b); if (inverse) w = fr_inv<F>(w);
    const fr_t k32 = fr_from_u64<F>(32);
    std::vector<fr29_t> tw(B / 2), x(B);
    { fr_t acc = k32; for (size_t i = 0; i < B / 2; ++i) { tw[i] = fr29_unpack(acc); acc = fr_mul<F>(acc, w); } }
    uint32_t D[9]; ntt29_offset<F>(D);
    auto brev = [&](size_t v) { size_t r = 0; for (int i = 0; i < log_b; ++i) r |= ((v >> i) & 1) << (log_b - 1 - i); return r; };
    for (size_t p = 0; p < B; ++p) x[brev(p)] = fr29_unpack(ld4(data + 4 * p));
    uint64_t ml = 0, mt = 0;
    const int head = log_b >= 3 ? 3 : log_b;
    {   // the register head of the kernels: groups of 2^head consecutive (bit-reversed) rows
        const fr29_t w1 = tw[log_b >= 3 ? B >> 3 : 0], w2 = tw[log_b >= 2 ? B >> 2 : 0], w3 = tw[log_b >= 3 ? 3 * (B >> 3) : 0];
        for (size_t g = 0; g < (B >> head); ++g) {
            fr29_t* xs = &x[g << head];
            if (head == 3) ntt29_head<F, 3>(*reinterpret_cast<fr29_t (*)[8]>(xs), w1, w2, w3, D);
            else if (head == 2) ntt29_head<F, 2>(*reinterpret_cast<fr29_t (*)[4]>(xs), w1, w2, w3, D);
            else ntt29_head<F, 1>(*reinterpret_cast<fr29_t (*)[2]>(xs), w1, w2, w3, D);
        }
    }
    for (int s = head + 1; s <= log_b; ++s) {
        const size_t half = (size_t)1 << (s - 1); const bool nrm = ntt29_norm_before(s);
        for (size_t bq = 0; bq < B / 2; ++bq) {
            const size_t j = bq & (half - 1), grp = bq >> (s - 1), i0 = (grp << s) + j, i1 = i0 + half;
            fr29_t a = x[i0], b = x[i1];
            { fr29_t bb = b, aa = a; if (nrm) { fr29_norm(bb); fr29_norm(aa); } for (int k = 0; k < 8; ++k) { ml = std::max<uint64_t>(ml, bb.l[k]); ml = std::max<uint64_t>(ml, aa.l[k]); } mt = std::max<uint64_t>(mt, std::max(bb.l[8], aa.l[8])); }
            ntt29_butterfly<F>(a, b, tw[j << (log_b - s)], D, nrm);
            x[i0] = a; x[i1] = b;
        }
    }
    const bool nrm_out = ntt29_norm_after(log_b);
    fr_t mult = k32; if (inverse) mult = fr_mul<F>(k32, fr_inv<F>(fr_from_u64<F>((uint64_t)B)));
    for (size_t k = 0; k < B; ++k) {
        fr29_t y = x[k];
        for (int i = 0; i < 8; ++i) ml = std::max<uint64_t>(ml, nrm_out ? 0 : y.l[i]);
        if (inverse) {   // an output with a table factor: the product is the reduction
            if (nrm_out) fr29_norm(y);
            mt = std::max<uint64_t>(mt, y.l[8]);
            const fr29_t r = fr29_mul_mont<F>(fr29_unpack(mult), y);
            st4(data + 4 * k, fr29_pack_reduce<F>(r.l));
        } else {         // no factor: reduce without a product
            fr29_partial_reduce<F>(y);
            st4(data + 4 * k, fr29_pack_reduce<F>(y.l));
        }
    }
    if (max_limb) *max_limb = ml; if (max_top) *max_top = mt;
    return 0;
}
extern "C" {
// fr29_partial_reduce on n lazy nine-limb values (9 x u32 each, any limbs below 2^32 with the value below 2^261): in place
int hc_partial_reduce(int field, uint32_t* limbs, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        fr29_t x; for (int k = 0; k < 9; ++k) x.l[k] = limbs[9 * i + k];
        if (field == 0) fr29_partial_reduce<PallasFr>(x); else fr29_partial_reduce<Bls12381Fr>(x);
        for (int k = 0; k < 9; ++k) limbs[9 * i + k] = x.l[k];
    }
    return 0;
}
int hc_ntt29(int field, uint64_t* data, int log_b, int inverse, uint64_t* max_limb, uint64_t* max_top) {
    return field == 0 ? hc_ntt29_impl<PallasFr>(data, log_b, inverse, max_limb, max_top) : hc_ntt29_impl<Bls12381Fr>(data, log_b, inverse, max_limb, max_top);
}
int hc_blake3(const uint8_t* p, size_t n, uint8_t* out32) { host::Blake3::hash(p, n, out32); return 0; }
int hc_chacha12_u64s(const uint8_t* seed32, size_t n, uint64_t* out) { host::ChaCha12Rng r(seed32); for (size_t i = 0; i < n; ++i) out[i] = r.next_u64(); return 0; }
int hc_from_le_bytes_mod_order(const uint8_t* b, size_t n, uint64_t* out) { st4(out, host::h_from_le_bytes_mod_order(b, n)); return 0; }
int hc_to_bytes_le(const uint64_t* a, uint8_t* out32) { host::h_to_bytes_le(ld4(a), out32); return 0; }

// kind 0: consts_for_width(t); 1: transcript; 2: t17 from seed string
void* hc_params_new(int kind, int t, const char* seed) {
    HcParams* P = new HcParams();
    P->ref = kind == 0 ? host::consts_for_width(t) : kind == 1 ? host::consts_transcript() : host::derive_consts(seed, 17, 8, 64);
    bind(P); return P;
}
int hc_params_ok(void* h) { return ((HcParams*)h)->kc.ok ? 1 : 0; }
void hc_params_free(void* h) { delete (HcParams*)h; }
int hc_params_export(void* h, uint64_t* mds, uint64_t* rc_full, uint64_t* rc_partial) {
    HcParams* P = (HcParams*)h;
    for (size_t i = 0; i < P->ref.mds.size(); ++i) st4(mds + 4 * i, P->ref.mds[i]);
    for (size_t i = 0; i < P->ref.rc_full.size(); ++i) st4(rc_full + 4 * i, P->ref.rc_full[i]);
    for (size_t i = 0; i < P->ref.rc_partial.size(); ++i) st4(rc_partial + 4 * i, P->ref.rc_partial[i]);
    return 0;
}
// One full round's linear layer  y = M * x  (x = S-box outputs) of t = 17 states, two ways on the host: through the in-place L*U rows the VALU
// kernels use (which = 0), and through an emulation of the matrix-core path (which = 1): signed recoding, the int8 FRAGMENT TABLES the kernels
// read (host_util.hpp mfma_frags: lane l of fragment (i, rt, e) holds A[row l & 31][k = 16 (l >> 5) + j]), D[row][col] = sum_k A[row][k] B[k][col]
// per tile, the accumulator rows as the two lanes of a sponge receive them, the fold and the Montgomery step of mfma_digits.hpp.  pre: the B_1 * M matrix.
int hc_full_round_linear(void* h, int which, int pre, uint64_t* states, size_t n) {
    HcParams* P = (HcParams*)h; const int t = P->dev.t;
    if (t != 17 || P->kc.mds_frag.empty()) return -1;
    std::vector<fr_t> st(t), out(t);
    const std::vector<int8_t>& F = pre ? P->kc.mds_pre_frag : P->kc.mds_frag;
    for (size_t s = 0; s < n; ++s) {
        for (int j = 0; j < t; ++j) st[j] = ld4(states + 4 * (s * t + j));
        if (which == 0) { ArrayState a{st.data()}; apply_lu(a, pre ? P->dev.lu_pre29 : P->dev.lu29, t); out = st; }
        else {
            std::vector<fr_t> xd(t); for (int e = 0; e < t; ++e) xd[e] = recode_signed(st[e]);
            for (int i = 0; i < t; ++i) {
                int64_t col[18]; for (int k = 0; k < 18; ++k) col[k] = 0;
                for (int rt = 0; rt < 2; ++rt) {
                    int32_t S[32];
                    for (int r = 0; r < 32; ++r) { int64_t acc = 0;
                        for (int e = 0; e < t; ++e) for (int kh = 0; kh < 2; ++kh) { const int8_t* a = &F[((((size_t)(i * 2 + rt) * t + e) * 64) + (r + 32 * kh)) * 16];
                            const int8_t* b = reinterpret_cast<const int8_t*>(xd[e].v) + 16 * kh; for (int j = 0; j < 16; ++j) acc += (int64_t)a[j] * b[j]; }
                        if (acc > 0x7fffffffll || acc < -0x80000000ll) return -2; S[r] = (int32_t)acc; }
                    int32_t lo[16], hi[16];
                    for (int reg = 0; reg < 16; ++reg) { const int row = (reg & 3) + 8 * (reg >> 2); lo[reg] = S[row]; hi[reg] = S[row + 4]; }
                    mfma_fold_rows(col, lo, hi, rt);
                }
                out[i] = mfma_finish_cols(col);
            }
        }
        for (int j = 0; j < t; ++j) st4(states + 4 * (s * t + j), out[j]);
    }
    return 0;
}
// kernel-form permutation (LU + sparse) of nstates AoS states on the host
int hc_permute_kernel_form(void* h, uint64_t* states, size_t n) {
    HcParams* P = (HcParams*)h; int t = P->dev.t; std::vector<fr_t> st(t);
    for (size_t i = 0; i < n; ++i) { for (int j = 0; j < t; ++j) st[j] = ld4(states + 4 * (i * t + j)); ArrayState s{st.data()}; permute_core(s, P->dev, false); for (int j = 0; j < t; ++j) st4(states + 4 * (i * t + j), st[j]); }
    return 0;
}
// the permutation with its partial rounds UNROLLED over all rp rounds from the chain tables (poseidon_chain.hpp chain_partial_model: the algebra and
// table scaling of the three-wave latency kernel); full rounds dense.  t = 17 only.
int hc_permute_chain_model(void* h, uint64_t* states, size_t n) {
    HcParams* P = (HcParams*)h; const int t = P->ref.t, half = P->ref.rf / 2; if (!P->dev.chain_a) return -1;
    std::vector<fr_t> st(t), o(t);
    auto dense = [&](const std::vector<fr_t>& M) { for (int i = 0; i < t; ++i) { fr_t acc = host::h_zero(); for (int j = 0; j < t; ++j) acc = host::h_add(acc, host::h_mul(M[(size_t)i * t + j], st[j])); o[i] = acc; } st = o; };
    for (size_t i = 0; i < n; ++i) {
        for (int j = 0; j < t; ++j) st[j] = ld4(states + 4 * (i * t + j));
        for (int r = 0; r < half; ++r) { for (int j = 0; j < t; ++j) st[j] = fr_pow5<PallasFr>(host::h_add(st[j], P->ref.rc_full[(size_t)r * t + j])); dense(r == half - 1 ? P->kc.mds_pre : P->kc.mds); }
        chain_partial_model(st.data(), P->dev);
        for (int r = half; r < P->ref.rf; ++r) { for (int j = 0; j < t; ++j) st[j] = fr_pow5<PallasFr>(host::h_add(st[j], P->ref.rc_full[(size_t)r * t + j])); dense(P->kc.mds); }
        for (int j = 0; j < t; ++j) st4(states + 4 * (i * t + j), st[j]);
    }
    return 0;
}
// the constants of the row-form Montgomery step (poseidon_chain.hpp row_consts_host): nine limbs of -r^-1 mod 2^261, five limbs of t = r - 2^254
int hc_row_consts(uint32_t* out14) { const RowConstsHost K = row_consts_host(); for (int i = 0; i < 9; ++i) out14[i] = K.ni[i]; for (int i = 0; i < 5; ++i) out14[9 + i] = K.t[i]; return 0; }
// the chain tables' shapes: chain_a [rp][64], chain_g [rp][9][64], chain_w [rp][t-1][9]; copies one table out (which = 0, 1, 2); returns its length in words
size_t hc_chain_table(void* h, int which, uint32_t* out, size_t cap) {
    HcParams* P = (HcParams*)h; const std::vector<uint32_t>& v = which == 0 ? P->kc.chain_a : which == 1 ? P->kc.chain_g : P->kc.chain_w;
    if (out) for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return v.size();
}
// reference-form (dense) permutation on the host, from the same constants
int hc_permute_dense(void* h, uint64_t* states, size_t n) {
    HcParams* P = (HcParams*)h; int t = P->ref.t; std::vector<fr_t> st(t);
    for (size_t i = 0; i < n; ++i) { for (int j = 0; j < t; ++j) st[j] = ld4(states + 4 * (i * t + j)); host::permute_dense(st.data(), P->ref); for (int j = 0; j < t; ++j) st4(states + 4 * (i * t + j), st[j]); }
    return 0;
}
// kernel bodies on the host ---------------------------------------------------------------------------
int hc_leaf_pair(void* tparams, const uint64_t* f, const uint64_t* f_next, size_t n, size_t m, uint64_t* hout) {
    HcParams* P = (HcParams*)tparams;
    const fr_t AB = host::h_tag("FSv1-ABSORB-BYTES"), CH = host::h_tag("FSv1-CHALLENGE");
    fr_t init[17]; for (auto& x : init) x = host::h_zero();
    init[0] = AB; init[1] = host::h_words("FRI/leaf/poseidon")[0]; init[2] = AB; init[3] = host::h_words("FRI/leaf")[0];
    init[6] = CH; init[7] = AB; init[8] = host::h_words("leaf")[0]; init[16] = host::h_tag("FSv1-TRANSCRIPT-INIT");
    fr_t st[17];
    for (size_t i = 0; i < n; ++i) { ArrayState s{st}; st4(hout + 4 * i, leaf_pair_body(s, P->dev, init, ld4(f + 4 * i), f_next ? ld4(f_next + 4 * (i / m)) : host::h_zero())); }
    return 0;
}
int hc_hash_ds_level(void* params, int mode, size_t arity, uint32_t level, uint64_t pos0, uint64_t label, const uint64_t* in0, const uint64_t* in1, size_t n_in, uint64_t* out) {
    HcParams* P = (HcParams*)params;
    DsJob J; J.arity_f = host::h_u64(arity); J.level_f = host::h_u64(level); J.label_f = host::h_u64(label); J.pos0 = pos0; J.arity = arity; J.n_in = n_in; J.mode = mode; J.cp_div = 1;
    J.n_out = mode == 1 ? n_in : (n_in + arity - 1) / arity;
    std::vector<fr_t> a(n_in), b(in1 ? n_in : 0), st(P->dev.t);
    for (size_t i = 0; i < n_in; ++i) { a[i] = ld4(in0 + 4 * i); if (in1) b[i] = ld4(in1 + 4 * i); }
    for (size_t k = 0; k < J.n_out; ++k) { ArrayState s{st.data()}; st4(out + 4 * k, hash_ds_body(s, P->dev, J, a.data(), in1 ? b.data() : nullptr, k)); }
    return 0;
}
int hc_tr_hash(void* tparams, const char* tag, const uint64_t* fields, size_t k, size_t n, uint64_t* out) {
    HcParams* P = (HcParams*)tparams;
    std::vector<fr_t> fr; const fr_t AB = host::h_tag("FSv1-ABSORB-BYTES"), CH = host::h_tag("FSv1-CHALLENGE");
    fr.push_back(AB); for (auto& w : host::h_words("FRI/FS")) fr.push_back(w);
    fr.push_back(AB); for (auto& w : host::h_words(tag)) fr.push_back(w);
    int np = (int)fr.size();
    fr.push_back(CH); fr.push_back(AB); for (auto& w : host::h_words("out")) fr.push_back(w);
    TrJob J; J.prefix = fr.data(); J.np = np; J.suffix = fr.data() + np; J.ns = (int)fr.size() - np; J.cap = host::h_tag("FSv1-TRANSCRIPT-INIT"); J.k = k; J.n = n;
    std::vector<fr_t> fl(n * k); for (size_t i = 0; i < n * k; ++i) fl[i] = ld4(fields + 4 * i);
    fr_t st[17];
    for (size_t i = 0; i < n; ++i) { ArrayState s{st}; st4(out + 4 * i, tr_hash_body(s, P->dev, J, fl.data(), i)); }
    return 0;
}
int hc_hash_stream(void* params, int mode, const uint64_t* a, size_t na, const uint64_t* b, size_t nb, const uint64_t* tag, size_t n, uint64_t* out) {
    HcParams* P = (HcParams*)params;
    std::vector<fr_t> av(n * na), bv(n * nb), st(P->dev.t);
    for (size_t i = 0; i < n * na; ++i) av[i] = ld4(a + 4 * i); for (size_t i = 0; i < n * nb; ++i) bv[i] = ld4(b + 4 * i);
    for (size_t k = 0; k < n; ++k) { ArrayState s{st.data()}; st4(out + 4 * k, hash_stream_body(s, P->dev, mode, av.data(), na, bv.data(), nb, tag ? ld4(tag) : host::h_zero(), k)); }
    return 0;
}

// ---- query plan / assemble (fri_plan.hpp) with the transcript hashes computed on the host -------------------
struct HcHasher : TrHasher {
    void* tp; explicit HcHasher(void* p) : tp(p) {}
    int32_t hash(const char* tag, const fr_t* fields, size_t k, size_t n, fr_t* out) override {
        std::vector<uint64_t> in(4 * k * n), o(4 * n);
        for (size_t i = 0; i < k * n; ++i) st4(in.data() + 4 * i, fields[i]);
        int rc = hc_tr_hash(tp, tag, in.data(), k, n, o.data()); if (rc) return rc;
        for (size_t i = 0; i < n; ++i) out[i] = ld4(o.data() + 4 * i);
        return 0;
    }
};
struct HcPlan { FriPlan plan; void* tp; };
void* hc_fri_plan_create(void* tparams, const uint64_t* roots, size_t n0, const size_t* schedule, size_t L, size_t r) {
    std::vector<fr_t> rt(L + 1); for (size_t l = 0; l <= L; ++l) rt[l] = ld4(roots + 4 * l);
    HcPlan* P = new HcPlan(); P->tp = tparams; P->plan.r = r; std::string err;
    if (!P->plan.shape.make(n0, schedule, L, rt.data(), err)) { delete P; return nullptr; }
    HcHasher H(tparams); if (fri_plan_make(P->plan, H)) { delete P; return nullptr; }
    return P;
}
size_t hc_fri_plan_num_requests(void* p) { return ((HcPlan*)p)->plan.req.size(); }
int hc_fri_plan_requests(void* p, uint32_t* kind, uint32_t* which, uint32_t* level, uint64_t* index) {
    auto& rq = ((HcPlan*)p)->plan.req;
    for (size_t i = 0; i < rq.size(); ++i) { kind[i] = rq[i].kind; which[i] = rq[i].which; level[i] = rq[i].level; index[i] = rq[i].index; }
    return 0;
}
// returns the encoded length (0 on error); writes at most cap bytes
size_t hc_fri_plan_assemble(void* p, const uint64_t* values, size_t n_values, uint8_t* buf, size_t cap, size_t* est) {
    HcPlan* P = (HcPlan*)p; if (n_values != P->plan.req.size()) return 0;
    std::vector<fr_t> v(n_values); for (size_t i = 0; i < n_values; ++i) v[i] = ld4(values + 4 * i);
    ReplaySource src(v.data(), v.size()); HcHasher H(P->tp); std::vector<uint8_t> b; size_t e = 0;
    if (assemble_proof(P->plan.shape, P->plan.r, H, src, b, e) || src.pos != v.size()) return 0;
    if (est) *est = e;
    if (buf && cap >= b.size()) memcpy(buf, b.data(), b.size());
    return b.size();
}
void hc_fri_plan_free(void* p) { delete (HcPlan*)p; }

// ---- verifier (fri_verify.hpp) with every hash computed by the host instantiation of the kernel bodies -------------
struct HcVerifyHasher : VerifyHasher {
    void* tp; explicit HcVerifyHasher(void* t) : tp(t) {}
    std::map<int, HcParams*> mp;
    ~HcVerifyHasher() { for (auto& kv : mp) delete kv.second; }
    HcParams* params(size_t arity) { int t = host::width_for_arity(arity); auto it = mp.find(t); if (it != mp.end()) return it->second; HcParams* P = new HcParams(); P->ref = host::consts_for_width(t); bind(P); mp[t] = P; return P; }
    int32_t leaf_pairs(const fr_t* f, const fr_t* s, size_t n, fr_t* out) override {
        std::vector<uint64_t> a(4 * n), b(4 * n), o(4 * n);
        for (size_t i = 0; i < n; ++i) { st4(a.data() + 4 * i, f[i]); st4(b.data() + 4 * i, s[i]); }
        hc_leaf_pair(tp, a.data(), b.data(), n, 1, o.data());
        for (size_t i = 0; i < n; ++i) out[i] = ld4(o.data() + 4 * i);
        return 0;
    }
    int32_t ds_nodes(size_t arity, size_t chunk, uint32_t level, uint64_t label, const uint64_t* positions, const fr_t* children, size_t n, fr_t* out) override {
        HcParams* P = params(arity); std::vector<fr_t> st(P->dev.t);
        DsJob J; J.arity_f = host::h_u64(arity); J.level_f = host::h_u64(level); J.label_f = host::h_u64(label); J.pos0 = 0; J.arity = chunk; J.n_in = n * chunk; J.n_out = n; J.mode = 0; J.pos_list = positions;
        for (size_t k = 0; k < n; ++k) { ArrayState s{st.data()}; out[k] = hash_ds_body(s, P->dev, J, children, nullptr, k); }
        return 0;
    }
    int32_t ds_pair_leaves(size_t arity, uint64_t label, const uint64_t* positions, const fr_t* f, const fr_t* cp, size_t n, fr_t* out) override {
        HcParams* P = params(arity); std::vector<fr_t> st(P->dev.t);
        DsJob J; J.arity_f = host::h_u64(arity); J.level_f = host::h_u64(0xFFFFFFFFu); J.label_f = host::h_u64(label); J.pos0 = 0; J.arity = arity; J.n_in = n; J.n_out = n; J.mode = 1; J.pos_list = positions;
        for (size_t k = 0; k < n; ++k) { ArrayState s{st.data()}; out[k] = hash_ds_body(s, P->dev, J, f, cp, k); }
        return 0;
    }
};
// 1 accept, 0 reject, negative: internal error
int hc_deep_fri_verify(void* tparams, const uint8_t* bytes, size_t len, const size_t* schedule, size_t L, size_t r) {
    DeepFriProofHost P; if (!decode_proof(bytes, len, P)) return 0;
    HcVerifyHasher H(tparams); bool ok = false;
    int32_t rc = deep_fri_verify_host(H, P, schedule, L, r, ok); if (rc) return rc;
    return ok ? 1 : 0;
}
// MerkleProver::verify_single / verify_pairs (merkle/src/lib.rs:800-855) over the canonical MerkleProof encoding
int hc_merkle_verify(void* tparams, int pairs, size_t cfg_arity, uint64_t label, const uint64_t* root, const size_t* idx, size_t k, const uint64_t* vals, const uint64_t* cp, const uint8_t* proof, size_t len) {
    ByteReader R(proof, len); MerkleProofHost pr; if (!dec_mproof(R, pr) || R.left()) return 0;
    HcVerifyHasher H(tparams); bool ok = false;
    std::vector<size_t> ix(idx, idx + k); std::vector<fr_t> v(k), c(k);
    for (size_t i = 0; i < k; ++i) { v[i] = ld4(vals + 4 * i); if (pairs) c[i] = ld4(cp + 4 * i); }
    int32_t rc = pairs ? verify_pairs_ds_host(H, cfg_arity, ld4(root), ix, v, c, pr, label, ok) : verify_many_ds_host(H, cfg_arity, ld4(root), ix, v, pr, label, ok);
    if (rc) return rc;
    return ok ? 1 : 0;
}

}  // extern "C"

// stark_mlwe_amd/csrc/capi_ntt.hip — NTT / iNTT / LDE entry points and plans (crates/fft/src/lib.rs:6-32),
// the per-GPU building blocks of the multi-GPU six-step NTT, and the synthetic-input generator.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "ctx.hpp"
#include "fri_dev.hpp"

using namespace stark;

namespace stark {

struct DevTable { fr_t* lo = nullptr; fr_t* hi = nullptr; int lo_bits = 0; PowTable view() const { return PowTable{lo, hi, lo_bits}; } };
struct NttPlan {
    int field = 0, log_n = 0; bool inverse = false;
    int P = 1; int log_b[3] = {0, 0, 0};
    fr_t* stage_tw[3] = {nullptr, nullptr, nullptr};   // w_B^(+-e), e < B/2 per pass
    DevTable root;                                    // powers of w_N^(+-1)
    fr_t* scale = nullptr;                            // n^-1 (inverse plans)
    // coset cache (one coset value at a time)
    bool have_coset = false; fr_t coset; DevTable coset_tab;
    // direct tables (log_n <= 24): inter-pass twiddles of the strided passes, coset pre-scale
    fr_t* tw_direct[2] = {nullptr, nullptr}; fr_t* coset_direct = nullptr;
    // merged form for plans with a strided first pass: (g^S)^p by point + the first pass's twiddles times g^rest (one full-length table read less per element)
    fr_t* coset_small = nullptr; fr_t* tw_coset_direct = nullptr;
    // power tables of coset shifts used by the multi-GPU column phase (a handful: the 2^log_blowup cosets of an LDE)
    std::vector<std::pair<fr_t, DevTable>> shift_tabs;
    // plain (c0 = 1) power tables for the element-wise kernels of stark_ntt_rows_coset_dev: coset shifts, and w_N itself (key = one)
    std::vector<std::pair<fr_t, DevTable>> plain_tabs;
    ~NttPlan() {
        for (auto& st : shift_tabs) { if (st.second.lo) (void)hipFree(st.second.lo); if (st.second.hi) (void)hipFree(st.second.hi); }
        for (auto& st : plain_tabs) { if (st.second.lo) (void)hipFree(st.second.lo); if (st.second.hi) (void)hipFree(st.second.hi); }
        for (auto p : tw_direct) if (p) (void)hipFree(p);
        if (coset_direct) (void)hipFree(coset_direct);
        if (coset_small) (void)hipFree(coset_small); if (tw_coset_direct) (void)hipFree(tw_coset_direct);
        for (auto p : stage_tw) if (p) (void)hipFree(p);
        if (root.lo) (void)hipFree(root.lo); if (root.hi) (void)hipFree(root.hi); if (scale) (void)hipFree(scale);
        if (coset_tab.lo) (void)hipFree(coset_tab.lo); if (coset_tab.hi) (void)hipFree(coset_tab.hi);
    }
};

}  // namespace stark

static const size_t kMaxLds = 160 * 1024;
// direct (one-product) twiddle / coset tables for transforms up to 2^ntt_direct_max_log points (option, default 24; 0 disables): they cost
// n*32 B of HBM per strided pass and plan, which the VALU-bound transform does not notice, and save a product per element and pass
static inline int ntt_direct_max(const stark_ctx* ctx) { return ctx->opt_ntt_direct_max_log; }

// The NTT kernels multiply by table entries with ONE Montgomery step by 2^261 on nine-limb values (ntt_dev.hpp): every table
// of a plan carries the factor 32 that makes that step a product in the 2^256 domain.
template <class F> static inline fr_t x32(const fr_t& v) { return fr_mul<F>(v, fr_from_u64<F>(32)); }
template <class F> static inline void pass_consts(NttPassArgs& A) {
    ntt29_offset<F>(A.dlimb);
}

template <class F>
static int32_t fill_table(stark_ctx* ctx, const fr_t& g, const fr_t& c0, int lo_bits, int hi_bits, DevTable& T) {
    if (T.lo) { (void)hipFree(T.lo); T.lo = nullptr; } if (T.hi) { (void)hipFree(T.hi); T.hi = nullptr; }
    STARK_HIP(ctx, hipMalloc((void**)&T.lo, ((size_t)1 << lo_bits) * sizeof(fr_t))); STARK_HIP(ctx, hipMalloc((void**)&T.hi, ((size_t)1 << hi_bits) * sizeof(fr_t)));
    T.lo_bits = lo_bits;
    uint64_t tot = (1ull << lo_bits) + (1ull << hi_bits);
    hipLaunchKernelGGL(k_fill_pow_table<F>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, T.lo, T.hi, lo_bits, hi_bits, g, c0);
    STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
}

template <class F>
static int32_t get_plan(stark_ctx* ctx, int log_n, bool inverse, NttPlan** out) {
    uint64_t key = ((uint64_t)F::ID << 40) | ((uint64_t)log_n << 8) | (inverse ? 1 : 0);
    auto it = ctx->plans.find(key);
    if (it != ctx->plans.end()) { *out = it->second; return STARK_OK; }
    NttPlan* p = new NttPlan(); p->field = F::ID; p->log_n = log_n; p->inverse = inverse;
    if (log_n <= 10) { p->P = 1; p->log_b[0] = log_n; }
    else if (log_n <= 20) { p->P = 2; p->log_b[0] = (log_n + 1) / 2; p->log_b[1] = log_n - p->log_b[0]; }
    else { p->P = 3; p->log_b[0] = (log_n + 2) / 3; p->log_b[1] = (log_n - p->log_b[0] + 1) / 2; p->log_b[2] = log_n - p->log_b[0] - p->log_b[1]; }
    auto bail = [&](int32_t rc) { delete p; return rc; };
    fr_t w = fr_root_of_unity<F>((unsigned)log_n); if (inverse) w = fr_inv<F>(w);
    int lo_bits = (log_n + 1) / 2, hi_bits = log_n - lo_bits;
    { int32_t rc = fill_table<F>(ctx, w, x32<F>(fr_one<F>()), lo_bits, hi_bits, p->root); if (rc) return bail(rc); }
    for (int i = 0; i < p->P; ++i) {
        int lb = p->log_b[i]; fr_t wb = fr_root_of_unity<F>((unsigned)lb); if (inverse) wb = fr_inv<F>(wb);
        DevTable T; int32_t rc = fill_table<F>(ctx, wb, x32<F>(fr_one<F>()), lb > 0 ? lb - 1 : 0, 0, T); if (rc) return bail(rc);
        p->stage_tw[i] = T.lo; (void)hipFree(T.hi);
    }
    if (ntt_direct_max(ctx) >= log_n) {       // direct twiddle tables: 2^log_m entries per strided pass
        int rem = log_n;
        for (int i = 0; i + 1 < p->P; ++i) {
            if (hipMalloc((void**)&p->tw_direct[i], ((size_t)1 << rem) * sizeof(fr_t)) != hipSuccess) { p->tw_direct[i] = nullptr; (void)hipGetLastError(); break; }   // no memory: keep the two-level lookup
            hipLaunchKernelGGL(k_fill_tw_direct<F>, dim3((unsigned)((((uint64_t)1 << rem) + 255) / 256)), dim3(256), 0, ctx->stream, p->root.view(), log_n, rem, p->log_b[i], p->tw_direct[i]);
            rem -= p->log_b[i];
        }
    }
    if (inverse) {
        fr_t ninv = x32<F>(fr_inv<F>(fr_from_u64<F>(1ull << log_n)));
        if (hipMalloc((void**)&p->scale, sizeof(fr_t)) != hipSuccess) return bail(ctx->fail(STARK_ERR_OOM, "ntt scale"));
        if (hipMemcpyAsync(p->scale, &ninv, sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return bail(ctx->fail(STARK_ERR_HIP, "ntt scale copy"));
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(ctx->fail(STARK_ERR_HIP, "sync"));
    }
    ctx->plans[key] = p; *out = p; return STARK_OK;
}

static inline size_t ntt_lds_bytes(int log_b, int log_c) { return ntt_tile_words((size_t)1 << (log_b + log_c), log_b > 0 ? (size_t)1 << (log_b - 1) : 1) * 4; }
// one workgroup per CU (tile > 80 KiB of LDS) => 512 threads so that every SIMD still holds 2 waves
static inline unsigned ntt_threads(size_t lds) { return lds > 80 * 1024 ? 512u : 256u; }
// tile elements E = B*C: 2^11 by default (64 KiB + twiddles => 2 workgroups per CU); option "ntt_log_tile" overrides for tuning
static inline int ntt_minw(const stark_ctx* ctx) { return ctx->opt_ntt_min_waves; }
// total_log: log2 of all elements the launch covers; small launches take smaller tiles so that the grid still fills the chip
static inline int pick_log_c(const stark_ctx* ctx, int log_b, int cap, int total_log = 30) {
    int le = ctx->opt_ntt_log_tile; if (!ctx->opt_ntt_log_tile_forced && total_log - le < 9) le = std::max(8, std::min(le, total_log - 8));
    int lc = std::max(2, le - log_b); lc = std::max(0, std::min(lc, cap));
    while (lc > 0 && ntt_lds_bytes(log_b, lc) > kMaxLds) --lc;      // 2^10-point sub-NTTs: two columns per tile (36 B per element in LDS)
    return lc;
}

template <class F, int MINW, bool STRIDED>
static void launch_pass(bool pre, unsigned th, size_t lds, hipStream_t st, const NttPassArgs& A, const fr_t* src, fr_t* dst) {
    const dim3 grid(A.ntiles), block(th);
    if (STRIDED) { if (pre) hipLaunchKernelGGL((k_ntt_strided<F, MINW, true>), grid, block, lds, st, A, src, dst); else hipLaunchKernelGGL((k_ntt_strided<F, MINW, false>), grid, block, lds, st, A, src, dst); }
    else { if (pre) hipLaunchKernelGGL((k_ntt_last<F, MINW, true>), grid, block, lds, st, A, src, dst); else hipLaunchKernelGGL((k_ntt_last<F, MINW, false>), grid, block, lds, st, A, src, dst); }
}
template <class F, bool STRIDED>
static int32_t launch_any(stark_ctx* ctx, NttPassArgs A, uint64_t total_elems, bool pre, const fr_t* src, fr_t* dst) {
    size_t lds = ntt_lds_bytes(A.log_b, A.log_c);
    if (lds > kMaxLds) return ctx->fail(STARK_ERR_UNSUPPORTED, "NTT tile exceeds LDS");
    A.ntiles = (uint32_t)(total_elems >> (A.log_b + A.log_c));
    if (ntt_minw(ctx) > 2 && lds <= 40 * 1024) launch_pass<F, 4, STRIDED>(pre, 256u, lds, ctx->stream, A, src, dst);
    else launch_pass<F, 2, STRIDED>(pre, ntt_threads(lds), lds, ctx->stream, A, src, dst);
    STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
}
template <class F> static int32_t launch_strided(stark_ctx* ctx, const NttPassArgs& A, uint64_t total_elems, const fr_t* src, fr_t* dst) { return launch_any<F, true>(ctx, A, total_elems, A.pre_direct || A.pre.lo, src, dst); }
template <class F> static int32_t launch_last(stark_ctx* ctx, const NttPassArgs& A, uint64_t total_elems, const fr_t* src, fr_t* dst) { return launch_any<F, false>(ctx, A, total_elems, A.pre.lo != nullptr, src, dst); }

template <class F> static int32_t plain_table(stark_ctx* ctx, NttPlan* big, const fr_t& base, int log_n, PowTable* out);   // plain (c0 = 1) power tables, defined below
// `batch` vectors of 2^log_n elements each, contiguous.  data is transformed in place (scratch from the context).
template <class F>
static int32_t ntt_run(stark_ctx* ctx, fr_t* data, int log_n, uint64_t batch, bool inverse, const fr_t* coset, const fr_t* scale_override_dev, int log_nonzero = -1) {
    if (log_n < 0 || log_n > 30) return ctx->fail(STARK_ERR_INVALID_ARG, "log_n out of range");
    if (batch == 0) return STARK_OK;
    if (log_n == 0) {   // size-1 transform: identity (n^-1 = 1, g^0 = 1)
        return STARK_OK;
    }
    NttPlan* p = nullptr; STARK_TRY(get_plan<F>(ctx, log_n, inverse, &p));
    PowTable none{nullptr, nullptr, 0};
    PowTable pre = none, post = none;
    if (coset) {
        if (!p->have_coset || !fr_eq(p->coset, *coset)) {
            int lo_bits = (log_n + 1) / 2, hi_bits = log_n - lo_bits;
            if (!inverse) STARK_TRY(fill_table<F>(ctx, *coset, x32<F>(fr_one<F>()), lo_bits, hi_bits, p->coset_tab));              // g^j
            else STARK_TRY(fill_table<F>(ctx, fr_inv<F>(*coset), x32<F>(fr_inv<F>(fr_from_u64<F>(1ull << log_n))), lo_bits, hi_bits, p->coset_tab));   // n^-1 g^-k
            p->coset = *coset; p->have_coset = true;
            if (p->coset_direct) { (void)hipFree(p->coset_direct); p->coset_direct = nullptr; }
            if (p->coset_small) { (void)hipFree(p->coset_small); p->coset_small = nullptr; }
            if (p->tw_coset_direct) { (void)hipFree(p->tw_coset_direct); p->tw_coset_direct = nullptr; }
            bool merged = false;
            if (!inverse && p->P >= 2 && p->tw_direct[0] && ctx->opt_ntt_merged_coset && ntt_direct_max(ctx) >= log_n) {
                // merged tables: the pre-scale's g^rest goes into the first pass's twiddle table, what is left is (g^S)^p by point index
                PowTable gplain; STARK_TRY(plain_table<F>(ctx, p, *coset, log_n, &gplain));
                const int lb0 = p->log_b[0], ls = log_n - lb0;
                if (hipMalloc((void**)&p->coset_small, ((size_t)1 << lb0) * sizeof(fr_t)) == hipSuccess && hipMalloc((void**)&p->tw_coset_direct, ((size_t)1 << log_n) * sizeof(fr_t)) == hipSuccess) {
                    hipLaunchKernelGGL(k_fill_coset_merged<F>, dim3((unsigned)((((uint64_t)1 << log_n) + 255) / 256)), dim3(256), 0, ctx->stream, p->coset_tab.view(), gplain, (const fr_t*)p->tw_direct[0], ls, lb0, p->coset_small, p->tw_coset_direct);
                    merged = hipGetLastError() == hipSuccess;
                }
                if (!merged) { (void)hipGetLastError(); if (p->coset_small) { (void)hipFree(p->coset_small); p->coset_small = nullptr; } if (p->tw_coset_direct) { (void)hipFree(p->tw_coset_direct); p->tw_coset_direct = nullptr; } }
            }
            if (!merged) {
                if (!inverse && ntt_direct_max(ctx) >= log_n && hipMalloc((void**)&p->coset_direct, ((size_t)1 << log_n) * sizeof(fr_t)) == hipSuccess)
                    hipLaunchKernelGGL(k_fill_pow_direct<F>, dim3((unsigned)((((uint64_t)1 << log_n) + 255) / 256)), dim3(256), 0, ctx->stream, p->coset_tab.view(), 1ull << log_n, p->coset_direct);
                else (void)hipGetLastError();
            }
        }
        if (!inverse) pre = p->coset_tab.view(); else post = p->coset_tab.view();
    }
    const fr_t* pre_direct = (coset && !inverse) ? p->coset_direct : nullptr;
    const bool merged = coset && !inverse && p->coset_small && p->tw_coset_direct;
    const uint64_t total = batch << log_n;
    fr_t* scratch = nullptr;
    if (p->P > 1) { void* s = nullptr; STARK_TRY(ctx_scratch(ctx, total * sizeof(fr_t), &s)); scratch = (fr_t*)s; }
    NttPassArgs A; memset(&A, 0, sizeof(A)); pass_consts<F>(A);
    A.log_n = log_n; A.root = p->root.view(); A.pre = none; A.post = none; A.scale = nullptr; A.rest0 = 0; A.log_vec = log_n;
    const fr_t* src = data;
    int rem = log_n;                       // log2 of the current sub-problem size
    for (int i = 0; i + 1 < p->P; ++i) {   // strided passes
        A.log_b = p->log_b[i]; A.log_m = rem; A.stride = 1ull << (rem - A.log_b);
        A.log_c = pick_log_c(ctx, A.log_b, rem - A.log_b, log_n);
        A.stage_tw = p->stage_tw[i]; A.pre = (i == 0) ? pre : none; A.pre_direct = (i == 0) ? pre_direct : nullptr; A.tw_direct = p->tw_direct[i];
        A.pre_small = nullptr;
        if (i == 0 && merged) { A.pre_small = p->coset_small; A.tw_direct = p->tw_coset_direct; }
        // zero-padded input (LDE): element j is non-zero only for j < 2^log_nonzero; in the first strided pass that is the points p < 2^log_nonzero / stride
        A.nz_points = (i == 0 && log_nonzero >= 0 && log_nonzero < log_n && (1ull << log_nonzero) >= A.stride) ? (uint32_t)((1ull << log_nonzero) / A.stride) : 0u;
        STARK_TRY(launch_strided<F>(ctx, A, total, src, scratch));
        src = scratch; rem -= A.log_b;
    }
    A.pre = (p->P == 1) ? pre : none; A.pre_direct = nullptr; A.pre_small = nullptr; A.tw_direct = nullptr; A.nz_points = 0;
    A.log_b = p->log_b[p->P - 1]; A.stage_tw = p->stage_tw[p->P - 1];
    A.log_b1 = p->P >= 2 ? p->log_b[0] : 0; A.log_b2 = p->P == 3 ? p->log_b[1] : 0;
    A.log_c = p->P == 1 ? 0 : pick_log_c(ctx, A.log_b, A.log_b1, log_n);
    A.post = post; A.scale = post.lo ? nullptr : (scale_override_dev ? scale_override_dev : (inverse ? p->scale : nullptr));
    STARK_TRY(launch_last<F>(ctx, A, total, src, data));
    return STARK_OK;
}

template <class F>
static int32_t lde_run(stark_ctx* ctx, const fr_t* evals, int log_n, int log_blowup, const fr_t* coset, fr_t* out) {
    const uint64_t n = 1ull << log_n, N = n << log_blowup;
    STARK_HIP(ctx, hipMemcpyAsync(out, evals, n * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
    STARK_TRY(ntt_run<F>(ctx, out, log_n, 1, true, nullptr, nullptr));                       // evaluations on H -> coefficients
    fr_t one = fr_one<F>(); bool unit = !coset || fr_eq(*coset, one);
    // The zero padding is never written when the big transform has a strided first pass whose stride divides n: that pass reads
    // only the n coefficient rows and takes the rest as zero (NttPassArgs::nz_points).  Otherwise (tiny transforms) pad for real.
    // (first-pass size taken from the plan itself, so that a retuned split can never leave out[n..N) unwritten AND unread-as-zero)
    const int big = log_n + log_blowup;
    NttPlan* bp = nullptr; STARK_TRY(get_plan<F>(ctx, big, false, &bp));
    const bool skip = N > n && bp->P > 1 && (big - bp->log_b[0]) <= log_n;
    if (N > n && !skip) { hipLaunchKernelGGL(k_zero_fill<F>, dim3((unsigned)((N - n + 255) / 256)), dim3(256), 0, ctx->stream, out + n, N - n); STARK_HIP(ctx, hipGetLastError()); }
    return ntt_run<F>(ctx, out, big, 1, false, unit ? nullptr : coset, nullptr, skip ? log_n : -1);   // coefficients -> coset evaluations on the larger domain
}

void stark::ntt_plans_free(stark_ctx* ctx) { for (auto& kv : ctx->plans) delete kv.second; ctx->plans.clear(); }

// Per-DEVICE kernel attributes (the default tile is 64 KiB + twiddles, above the 64 KiB a kernel may use without opting in):
// called from stark_ctx_create with the context's device current, so a process holding contexts on several GPUs sets them on each.
template <class F> static void set_attrs_for() {
    (void)hipFuncSetAttribute((const void*)k_ntt_strided<F, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)k_ntt_strided<F, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)k_ntt_last<F, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)k_ntt_last<F, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
}
void stark::ntt_set_attrs() { set_attrs_for<PallasFr>(); set_attrs_for<Bls12381Fr>(); }

// Multi-GPU phase A: column NTTs of size 2^log_rows over a row-major [2^log_rows][ncols] slab whose first
// column has global index col0, followed by the twiddle w_N^(col_global * k), N = 2^log_n.  In place.
template <class F>
static int32_t columns_run(stark_ctx* ctx, fr_t* slab, int log_rows, uint64_t ncols, uint64_t col0, int log_n, bool inverse, const fr_t* shift = nullptr) {
    if (log_rows < 1 || log_rows > 10) return ctx->fail(STARK_ERR_UNSUPPORTED, "column NTT size must be 2..1024");
    if (ncols == 0 || (ncols & (ncols - 1))) return ctx->fail(STARK_ERR_INVALID_ARG, "ncols must be a power of two");
    NttPlan* big = nullptr; STARK_TRY(get_plan<F>(ctx, log_n, inverse, &big));        // root table of w_N
    NttPlan* sm = nullptr; STARK_TRY(get_plan<F>(ctx, log_rows, inverse, &sm));       // stage twiddles of w_R (P == 1 plan)
    NttPassArgs A; memset(&A, 0, sizeof(A)); pass_consts<F>(A);
    int log_cols = 0; while ((1ull << log_cols) < ncols) ++log_cols;
    A.log_b = log_rows; A.log_c = pick_log_c(ctx, log_rows, log_cols); A.log_n = log_n; A.stride = ncols; A.log_m = log_n;
    A.stage_tw = sm->stage_tw[0]; A.root = big->root.view(); A.rest0 = col0;
    if (shift && !fr_eq(*shift, fr_one<F>())) {
        // coset evaluation: x[j] *= shift^j with j the GLOBAL natural index of the element (row * C + global column)
        if (inverse) return ctx->fail(STARK_ERR_UNSUPPORTED, "column phase: the coset pre-scale belongs to a forward transform");
        DevTable* T = nullptr;
        for (auto& st : big->shift_tabs) if (fr_eq(st.first, *shift)) T = &st.second;
        if (!T) {
            if (big->shift_tabs.size() >= 32) { STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); auto& old = big->shift_tabs.front(); (void)hipFree(old.second.lo); (void)hipFree(old.second.hi); big->shift_tabs.erase(big->shift_tabs.begin()); }
            big->shift_tabs.push_back({*shift, DevTable()});
            const int lo_bits = (log_n + 1) / 2, hi_bits = log_n - lo_bits;
            STARK_TRY(fill_table<F>(ctx, *shift, x32<F>(fr_one<F>()), lo_bits, hi_bits, big->shift_tabs.back().second));
            T = &big->shift_tabs.back().second;
        }
        A.pre = T->view(); A.pre_row_stride = 1ull << (log_n - log_rows);
    }
    return launch_strided<F>(ctx, A, (uint64_t)ncols << log_rows, slab, slab);
}
// Multi-GPU forward transform, first local phase, on the layout the inverse transform leaves behind (dist.py ShardedLde): rows k1 = row0 + i of the
// [R][C] view c[k1 + R k'] (R = 2^(log_n - log_cols) rows in the whole vector, this rank holds nrows of them, each contiguous over k').
//   dst[i][m] = w_n^(k1 m) * sum_k' src[i][k'] shift^(k' R + k1) w_C^(k' m)
// i.e. coset pre-scale at the natural index, size-C transforms along the contiguous axis, inter-step twiddle.  The second phase is a plain size-R
// transform over k1 after the exchange.  src is left untouched (the 2^log_blowup cosets of an LDE all start from the same coefficients).
template <class F>
static int32_t plain_table(stark_ctx* ctx, NttPlan* big, const fr_t& base, int log_n, PowTable* out) {
    for (auto& st : big->plain_tabs) if (fr_eq(st.first, base)) { *out = st.second.view(); return STARK_OK; }
    if (big->plain_tabs.size() >= 40) { STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); auto& old = big->plain_tabs.front(); (void)hipFree(old.second.lo); (void)hipFree(old.second.hi); big->plain_tabs.erase(big->plain_tabs.begin()); }
    big->plain_tabs.push_back({base, DevTable()});
    const int lo_bits = (log_n + 1) / 2, hi_bits = log_n - lo_bits;
    STARK_TRY(fill_table<F>(ctx, base, fr_one<F>(), lo_bits, hi_bits, big->plain_tabs.back().second));
    *out = big->plain_tabs.back().second.view(); return STARK_OK;
}
template <class F>
static int32_t rows_coset_run(stark_ctx* ctx, const fr_t* src, fr_t* dst, uint64_t nrows, int log_cols, uint64_t row0, int log_n, const fr_t& shift) {
    if (log_cols < 0 || log_cols > log_n || log_n > 30) return ctx->fail(STARK_ERR_INVALID_ARG, "rows_coset: sizes");
    const int log_rows = log_n - log_cols;
    if (!nrows || row0 + nrows > (1ull << log_rows)) return ctx->fail(STARK_ERR_INVALID_ARG, "rows_coset: row range");
    NttPlan* big = nullptr; STARK_TRY(get_plan<F>(ctx, log_n, false, &big));
    PowTable tsh, troot;
    STARK_TRY(plain_table<F>(ctx, big, shift, log_n, &tsh));
    // w_N is stored under a key no coset can equal by accident only if it differs from every shift in use: the root of unity itself is a legal shift,
    // and then the two tables are the same table — which is correct
    STARK_TRY(plain_table<F>(ctx, big, fr_root_of_unity<F>((unsigned)log_n), log_n, &troot));
    STARK_TRY(plain_table<F>(ctx, big, shift, log_n, &tsh));          // in case inserting w_N evicted it (the cache is a small FIFO)
    const uint64_t tot = nrows << log_cols; const unsigned grid = (unsigned)((tot + 255) / 256);
    hipLaunchKernelGGL(k_rows_coset_pre<F>, dim3(grid), dim3(256), 0, ctx->stream, src, dst, tsh, nrows, log_cols, row0, log_rows);
    STARK_HIP(ctx, hipGetLastError());
    STARK_TRY(ntt_run<F>(ctx, dst, log_cols, nrows, false, nullptr, nullptr));
    hipLaunchKernelGGL(k_rows_twiddle<F>, dim3(grid), dim3(256), 0, ctx->stream, dst, troot, nrows, log_cols, row0, log_n);
    STARK_HIP(ctx, hipGetLastError());
    return STARK_OK;
}

// ---- one column of a trace block-sharded over the ranks of the context's communicator: the LDE as ONE call (dist.py ShardedLde in C++) ----------
// A host without Python composes nothing: rank q passes its natural-order block [q n/W, (q+1) n/W) of the 2^log_n evaluations and receives its block of the
// 2^(log_n + log_blowup) evaluations on shift * <w_N>.  Four all-to-alls (stark_comm_all_to_all_dev; a device copy when there is one rank and no
// communicator), whatever the blow-up: natural rows -> column blocks; the inverse transform's transpose; ONE exchange for the first-phase outputs of all
// cosets; ONE to natural blocks.  Between them only local phases (columns_run, ntt_run rows, rows_coset_run) and the pack kernel.
static int32_t pack3(stark_ctx* ctx, const fr_t* src, fr_t* dst, uint64_t d0, uint64_t d1, uint64_t d2, int p0, int p1, int p2) {
    const uint64_t d[3] = {d0, d1, d2}, st[3] = {d1 * d2, d2, 1}; const int pm[3] = {p0, p1, p2};
    const uint64_t tot = d0 * d1 * d2; if (!tot) return STARK_OK;
    hipLaunchKernelGGL(k_permute3, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, src, dst, d[pm[0]], d[pm[1]], d[pm[2]], st[pm[0]], st[pm[1]], st[pm[2]]);
    STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
}
static int32_t exchange(stark_ctx* ctx, int W, const fr_t* send, fr_t* recv, size_t elems_per_peer) {
    if (W == 1 && !ctx->comm) { STARK_HIP(ctx, hipMemcpyAsync(recv, send, elems_per_peer * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream)); return STARK_OK; }
    return stark_comm_all_to_all_dev(ctx, send, recv, elems_per_peer * sizeof(fr_t));
}
// One rank's buffers and the five local phases between the four exchanges.  The same code runs under the real communicator (lde_sharded_run) and under
// the in-process emulation of W ranks on one GPU (stark_diag_lde_sharded_emulated_dev: what checks the W > 1 index arithmetic without a second GPU).
struct ShardPlan { int W, log_n, lb, log_rows, log_cols; uint64_t R, Cc, b, nrl, ncl, nl; };
struct ShardRank { int rank; const fr_t* block; fr_t* out; DevBuf t0, t1, big0, big1; };
static int32_t shard_plan(stark_ctx* ctx, int W, int log_n, int lb, ShardPlan& P) {
    if (log_n < 2 || log_n + lb > 30) return ctx->fail(STARK_ERR_INVALID_ARG, "lde_sharded: sizes");
    P.W = W; P.log_n = log_n; P.lb = lb; P.log_rows = std::min(10, log_n / 2); P.log_cols = log_n - P.log_rows;
    P.R = 1ull << P.log_rows; P.Cc = 1ull << P.log_cols; P.b = 1ull << lb;
    if (W < 1 || (W & (W - 1)) || P.R % W || P.Cc % W) return ctx->fail(STARK_ERR_INVALID_ARG, "lde_sharded: the ranks must divide the 2^log_rows x 2^log_cols view");
    P.nrl = P.R / W; P.ncl = P.Cc / W; P.nl = P.nrl * P.Cc;               // local rows, local columns, local elements of one size-n vector
    return STARK_OK;
}
static int32_t shard_alloc(stark_ctx* ctx, const ShardPlan& P, ShardRank& K) {
    STARK_HIP(ctx, K.t0.alloc(ctx, P.nl * sizeof(fr_t))); STARK_HIP(ctx, K.t1.alloc(ctx, P.nl * sizeof(fr_t)));
    STARK_HIP(ctx, K.big0.alloc(ctx, P.b * P.nl * sizeof(fr_t))); STARK_HIP(ctx, K.big1.alloc(ctx, P.b * P.nl * sizeof(fr_t)));
    return STARK_OK;
}
// phase k runs after exchange k (phase 0: before the first).  Exchange k sends `send_of(k)` and receives into `recv_of(k)`, `per_peer(k)` elements per peer.
static const fr_t* shard_send(const ShardRank& K, int x) { return x == 0 ? K.t0.fr() : x == 1 ? K.t1.fr() : K.big1.fr(); }
static fr_t* shard_recv(const ShardRank& K, int x) { return x == 0 ? K.t1.fr() : x == 1 ? K.t0.fr() : K.big0.fr(); }
static uint64_t shard_per_peer(const ShardPlan& P, int x) { return x < 2 ? P.nrl * P.ncl : P.b * P.nrl * P.ncl; }
template <class F>
static int32_t shard_phase(stark_ctx* ctx, const ShardPlan& P, ShardRank& K, int phase, const fr_t& shift) {
    const uint64_t W = P.W, nrl = P.nrl, ncl = P.ncl, nl = P.nl, b = P.b, R = P.R;
    switch (phase) {
    case 0:   // natural row block [nrl][W][ncl] -> [W][nrl][ncl]; exchange 0 makes it the column block [R][ncl]
        return pack3(ctx, K.block, K.t0.fr(), nrl, W, ncl, 1, 0, 2);
    case 1:   // inverse six-step transform: column phase + twiddle; exchange 1 is its transpose
        return columns_run<F>(ctx, K.t1.fr(), P.log_rows, ncl, (uint64_t)K.rank * ncl, P.log_n, true);
    case 2: { // [W][nrl][ncl] -> [nrl][C] (rows k1 of c[k1 + R k']); row phase with n^-1; first phase of every coset transform on that slab; pack for exchange 2
        STARK_TRY(pack3(ctx, K.t0.fr(), K.t1.fr(), W, nrl, ncl, 1, 0, 2));
        { DevBuf sc; const fr_t ninv = x32<F>(fr_inv<F>(fr_from_u64<F>(1ull << P.log_n)));
          STARK_HIP(ctx, sc.alloc(ctx, sizeof(fr_t))); STARK_HIP(ctx, hipMemcpyAsync(sc.p, &ninv, sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
          STARK_TRY((ntt_run<F>(ctx, K.t1.fr(), P.log_cols, nrl, true, nullptr, sc.fr())));
          STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); }                                       // the scale word is freed on leaving this scope
        const fr_t wN = fr_root_of_unity<F>((unsigned)(P.log_n + P.lb)); fr_t sh = shift;
        for (uint64_t s_ = 0; s_ < b; ++s_) { STARK_TRY((rows_coset_run<F>(ctx, K.t1.fr(), K.big0.fr() + s_ * nl, nrl, P.log_cols, (uint64_t)K.rank * nrl, P.log_n, sh))); sh = fr_mul<F>(sh, wN); }
        return pack3(ctx, K.big0.fr(), K.big1.fr(), b * nrl, W, ncl, 1, 0, 2); }                     // [b nrl][W][ncl] -> [W][b nrl][ncl]
    case 3:   // [W][b][nrl ncl] -> [b][R][ncl] -> [b ncl][R]; size-R transforms; [b][ncl][R] -> [R][ncl][b] for exchange 3
        STARK_TRY(pack3(ctx, K.big0.fr(), K.big1.fr(), W, b, nrl * ncl, 1, 0, 2));
        STARK_TRY(pack3(ctx, K.big1.fr(), K.big0.fr(), b, R, ncl, 0, 2, 1));
        STARK_TRY((ntt_run<F>(ctx, K.big0.fr(), P.log_rows, b * ncl, false, nullptr, nullptr)));
        return pack3(ctx, K.big0.fr(), K.big1.fr(), b, ncl, R, 2, 1, 0);
    default:  // [W][nrl][ncl b] -> [nrl][W][ncl b]: natural order of the interleaved result out[(K2 C + K1) b + s]
        return pack3(ctx, K.big0.fr(), K.out, W, nrl, ncl * b, 1, 0, 2);
    }
}
template <class F>
static int32_t lde_sharded_run(stark_ctx* ctx, const fr_t* block, int log_n, int lb, const fr_t& shift, fr_t* out) {
    const int W = ctx->comm ? stark_comm_size(ctx) : 1, rank = ctx->comm ? stark_comm_rank(ctx) : 0;
    ShardPlan P; STARK_TRY(shard_plan(ctx, W, log_n, lb, P));
    ShardRank K; K.rank = rank; K.block = block; K.out = out; STARK_TRY(shard_alloc(ctx, P, K));
    for (int phase = 0; phase < 5; ++phase) {
        STARK_TRY((shard_phase<F>(ctx, P, K, phase, shift)));
        if (phase < 4) STARK_TRY(exchange(ctx, W, shard_send(K, phase), shard_recv(K, phase), shard_per_peer(P, phase)));
    }
    return STARK_OK;
}
// W virtual ranks on ONE GPU: the phases of every rank in turn, each exchange as device copies (chunk q of rank p's send buffer -> chunk p of rank q's
// receive buffer).  Exactly the code and index arithmetic of the real W-rank run, minus RCCL.
template <class F>
static int32_t lde_sharded_emulated(stark_ctx* ctx, int W, const fr_t* evals, int log_n, int lb, const fr_t& shift, fr_t* out) {
    ShardPlan P; STARK_TRY(shard_plan(ctx, W, log_n, lb, P));
    std::vector<ShardRank> K(W);
    for (int r = 0; r < W; ++r) { K[r].rank = r; K[r].block = evals + (size_t)r * P.nl; K[r].out = out + ((size_t)r * P.nl << lb); STARK_TRY(shard_alloc(ctx, P, K[r])); }
    for (int phase = 0; phase < 5; ++phase) {
        for (int r = 0; r < W; ++r) STARK_TRY((shard_phase<F>(ctx, P, K[r], phase, shift)));
        if (phase < 4) {
            const uint64_t per = shard_per_peer(P, phase);
            for (int p = 0; p < W; ++p) for (int q = 0; q < W; ++q)
                STARK_HIP(ctx, hipMemcpyAsync(shard_recv(K[q], phase) + (size_t)p * per, shard_send(K[p], phase) + (size_t)q * per, per * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    return STARK_OK;
}

extern "C" {

int32_t stark_lde_sharded_dev(stark_ctx_t* ctx, int32_t field_id, const uint64_t* block, size_t log_n, size_t log_blowup, const uint64_t* shift4, uint64_t* out) {
    if (!ctx || !block || !out || !shift4 || block == out) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    const fr_t sh = load_fr(shift4);
    if (field_id == STARK_FIELD_PALLAS_FR) return lde_sharded_run<PallasFr>(ctx, as_fr(block), (int)log_n, (int)log_blowup, sh, as_fr(out));
    if (field_id == STARK_FIELD_BLS12_381_FR) return lde_sharded_run<Bls12381Fr>(ctx, as_fr(block), (int)log_n, (int)log_blowup, sh, as_fr(out));
    return ctx->fail(STARK_ERR_INVALID_ARG, "unknown field id");
}
int32_t stark_diag_lde_sharded_emulated_dev(stark_ctx_t* ctx, int32_t field_id, int32_t nranks, const uint64_t* evals, size_t log_n, size_t log_blowup, const uint64_t* shift4, uint64_t* out) {
    if (!ctx || !evals || !out || !shift4 || evals == out || nranks < 1) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    const fr_t sh = load_fr(shift4);
    if (field_id == STARK_FIELD_PALLAS_FR) return lde_sharded_emulated<PallasFr>(ctx, nranks, as_fr(evals), (int)log_n, (int)log_blowup, sh, as_fr(out));
    if (field_id == STARK_FIELD_BLS12_381_FR) return lde_sharded_emulated<Bls12381Fr>(ctx, nranks, as_fr(evals), (int)log_n, (int)log_blowup, sh, as_fr(out));
    return ctx->fail(STARK_ERR_INVALID_ARG, "unknown field id");
}
int32_t stark_ntt_rows_coset_dev(stark_ctx_t* ctx, int32_t field_id, const uint64_t* src, uint64_t* dst, size_t nrows, size_t log_cols, size_t row0, size_t log_n, const uint64_t* shift4) {
    if (!ctx || !src || !dst || !shift4 || src == dst) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    const fr_t sh = load_fr(shift4);
    if (field_id == STARK_FIELD_PALLAS_FR) return rows_coset_run<PallasFr>(ctx, as_fr(src), as_fr(dst), nrows, (int)log_cols, row0, (int)log_n, sh);
    if (field_id == STARK_FIELD_BLS12_381_FR) return rows_coset_run<Bls12381Fr>(ctx, as_fr(src), as_fr(dst), nrows, (int)log_cols, row0, (int)log_n, sh);
    return ctx->fail(STARK_ERR_INVALID_ARG, "unknown field id");
}
int32_t stark_ntt_dev(stark_ctx_t* ctx, int32_t field_id, uint64_t* data, size_t log_n, int32_t inverse, const uint64_t* coset4) {
    if (!ctx || !data) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    fr_t cs; if (coset4) cs = load_fr(coset4);
    if (field_id == STARK_FIELD_PALLAS_FR) return ntt_run<PallasFr>(ctx, as_fr(data), (int)log_n, 1, inverse != 0, coset4 ? &cs : nullptr, nullptr);
    if (field_id == STARK_FIELD_BLS12_381_FR) return ntt_run<Bls12381Fr>(ctx, as_fr(data), (int)log_n, 1, inverse != 0, coset4 ? &cs : nullptr, nullptr);
    return ctx->fail(STARK_ERR_INVALID_ARG, "unknown field id");
}
int32_t stark_ntt(stark_ctx_t* ctx, int32_t field_id, uint64_t* data, size_t log_n, int32_t inverse, const uint64_t* coset4) {
    if (!ctx || !data || log_n > 30) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    size_t bytes = ((size_t)1 << log_n) * sizeof(fr_t); DevBuf d; STARK_HIP(ctx, d.alloc(ctx, bytes));
    STARK_HIP(ctx, hipMemcpyAsync(d.p, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    STARK_TRY(stark_ntt_dev(ctx, field_id, (uint64_t*)d.p, log_n, inverse, coset4));
    STARK_HIP(ctx, hipMemcpyAsync(data, d.p, bytes, hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}
int32_t stark_lde_dev(stark_ctx_t* ctx, int32_t field_id, const uint64_t* evals, size_t log_n, size_t log_blowup, const uint64_t* coset4, uint64_t* out) {
    if (!ctx || !evals || !out || log_n + log_blowup > 30) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    fr_t cs; if (coset4) cs = load_fr(coset4);
    if (field_id == STARK_FIELD_PALLAS_FR) return lde_run<PallasFr>(ctx, as_fr(evals), (int)log_n, (int)log_blowup, coset4 ? &cs : nullptr, as_fr(out));
    if (field_id == STARK_FIELD_BLS12_381_FR) return lde_run<Bls12381Fr>(ctx, as_fr(evals), (int)log_n, (int)log_blowup, coset4 ? &cs : nullptr, as_fr(out));
    return ctx->fail(STARK_ERR_INVALID_ARG, "unknown field id");
}
int32_t stark_lde(stark_ctx_t* ctx, int32_t field_id, const uint64_t* evals, size_t log_n, size_t log_blowup, const uint64_t* coset4, uint64_t* out) {
    if (!ctx || !evals || !out || log_n + log_blowup > 30) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    size_t n = (size_t)1 << log_n, N = n << log_blowup; DevBuf di, dout; STARK_HIP(ctx, di.alloc(ctx, n * sizeof(fr_t))); STARK_HIP(ctx, dout.alloc(ctx, N * sizeof(fr_t)));
    STARK_HIP(ctx, hipMemcpyAsync(di.p, evals, n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    STARK_TRY(stark_lde_dev(ctx, field_id, (const uint64_t*)di.p, log_n, log_blowup, coset4, (uint64_t*)dout.p));
    STARK_HIP(ctx, hipMemcpyAsync(out, dout.p, N * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}

int32_t stark_ntt_columns_dev(stark_ctx_t* ctx, int32_t field_id, uint64_t* slab, size_t log_rows, size_t ncols, size_t col0, size_t log_n, int32_t inverse) {
    if (!ctx || !slab) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (field_id == STARK_FIELD_PALLAS_FR) return columns_run<PallasFr>(ctx, as_fr(slab), (int)log_rows, ncols, col0, (int)log_n, inverse != 0);
    if (field_id == STARK_FIELD_BLS12_381_FR) return columns_run<Bls12381Fr>(ctx, as_fr(slab), (int)log_rows, ncols, col0, (int)log_n, inverse != 0);
    return ctx->fail(STARK_ERR_INVALID_ARG, "unknown field id");
}
int32_t stark_ntt_columns_coset_dev(stark_ctx_t* ctx, int32_t field_id, uint64_t* slab, size_t log_rows, size_t ncols, size_t col0, size_t log_n, const uint64_t* shift4) {
    if (!ctx || !slab || !shift4) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    const fr_t sh = load_fr(shift4);
    if (field_id == STARK_FIELD_PALLAS_FR) return columns_run<PallasFr>(ctx, as_fr(slab), (int)log_rows, ncols, col0, (int)log_n, false, &sh);
    if (field_id == STARK_FIELD_BLS12_381_FR) return columns_run<Bls12381Fr>(ctx, as_fr(slab), (int)log_rows, ncols, col0, (int)log_n, false, &sh);
    return ctx->fail(STARK_ERR_INVALID_ARG, "unknown field id");
}
int32_t stark_permute3_dev(stark_ctx_t* ctx, const uint64_t* src, uint64_t* dst, size_t d0, size_t d1, size_t d2, int32_t p0, int32_t p1, int32_t p2) {
    if (!ctx || !src || !dst || src == dst) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    const int pm[3] = {p0, p1, p2}; int seen = 0; for (int i = 0; i < 3; ++i) { if (pm[i] < 0 || pm[i] > 2) return ctx->fail(STARK_ERR_INVALID_ARG, "permutation"); seen |= 1 << pm[i]; }
    if (seen != 7) return ctx->fail(STARK_ERR_INVALID_ARG, "permutation");
    const uint64_t d[3] = {d0, d1, d2}, st[3] = {(uint64_t)d1 * d2, (uint64_t)d2, 1};
    const uint64_t tot = (uint64_t)d0 * d1 * d2; if (!tot) return STARK_OK;
    hipLaunchKernelGGL(k_permute3, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, as_fr(src), as_fr(dst), d[pm[0]], d[pm[1]], d[pm[2]], st[pm[0]], st[pm[1]], st[pm[2]]);
    STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
}
int32_t stark_interleave_dev(stark_ctx_t* ctx, const uint64_t* src, uint64_t* dst, size_t n, size_t stride, size_t offset) {
    if (!ctx || !src || !dst || !stride || offset >= stride) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (!n) return STARK_OK;
    hipLaunchKernelGGL(k_interleave, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, as_fr(src), as_fr(dst), (uint64_t)n, (uint64_t)stride, (uint64_t)offset);
    STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
}
// Multi-GPU phase B: `nrows` contiguous NTTs of size 2^log_cols.  scale4 (optional) multiplies every output
// (the caller passes N^-1 of the FULL transform for an inverse; the per-row n^-1 is not applied).
int32_t stark_ntt_rows_dev(stark_ctx_t* ctx, int32_t field_id, uint64_t* slab, size_t nrows, size_t log_cols, int32_t inverse, const uint64_t* scale4) {
    if (!ctx || !slab) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    DevBuf sc; fr_t one_f;
    if (field_id != STARK_FIELD_PALLAS_FR && field_id != STARK_FIELD_BLS12_381_FR) return ctx->fail(STARK_ERR_INVALID_ARG, "unknown field id");
    const bool pallas = field_id == STARK_FIELD_PALLAS_FR;
    if (scale4) { fr_t s = load_fr(scale4); s = pallas ? x32<PallasFr>(s) : x32<Bls12381Fr>(s);      // into the kernels' table domain
        STARK_HIP(ctx, sc.alloc(ctx, sizeof(fr_t))); STARK_HIP(ctx, hipMemcpyAsync(sc.p, &s, sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); }
    else if (inverse) {   // suppress the plan's per-row n^-1: multiply by one
        one_f = pallas ? x32<PallasFr>(fr_one<PallasFr>()) : x32<Bls12381Fr>(fr_one<Bls12381Fr>());
        STARK_HIP(ctx, sc.alloc(ctx, sizeof(fr_t))); STARK_HIP(ctx, hipMemcpyAsync(sc.p, &one_f, sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream)); STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    int32_t rc;
    if (field_id == STARK_FIELD_PALLAS_FR) rc = ntt_run<PallasFr>(ctx, as_fr(slab), (int)log_cols, nrows, inverse != 0, nullptr, sc.p ? sc.fr() : nullptr);
    else if (field_id == STARK_FIELD_BLS12_381_FR) rc = ntt_run<Bls12381Fr>(ctx, as_fr(slab), (int)log_cols, nrows, inverse != 0, nullptr, sc.p ? sc.fr() : nullptr);
    else return ctx->fail(STARK_ERR_INVALID_ARG, "unknown field id");
    if (rc) return rc;
    if (sc.p) STARK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // sc freed on return
    return STARK_OK;
}

// ---- crates/field: Domain::new / compute_powers (field/src/lib.rs:43-53, 125-133) ------------------------------------------------
// get_root_of_unity(2^log_n) of the field (host-only: no context, no device)
int32_t stark_root_of_unity(int32_t field_id, size_t log_n, uint64_t* out4) {
    if (!out4 || log_n > 32) return STARK_ERR_INVALID_ARG;                                   // two-adicity 32 in both fields
    if (field_id == STARK_FIELD_PALLAS_FR) store_fr(out4, fr_root_of_unity<PallasFr>((unsigned)log_n));
    else if (field_id == STARK_FIELD_BLS12_381_FR) store_fr(out4, fr_root_of_unity<Bls12381Fr>((unsigned)log_n));
    else return STARK_ERR_INVALID_ARG;
    return STARK_OK;
}
}  // extern "C"
template <class F> static int32_t powers_run(stark_ctx* ctx, const fr_t& base, size_t n, fr_t* out_dev) {
    if (!n) return STARK_OK;
    int bits = 1; while (((size_t)1 << bits) < n) ++bits;
    const int lo_bits = (bits + 1) / 2, hi_bits = bits - lo_bits + 1;
    DevTable T; STARK_TRY(fill_table<F>(ctx, base, fr_one<F>(), lo_bits, hi_bits, T));
    hipLaunchKernelGGL(k_fill_pow_direct<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, T.view(), (uint64_t)n, out_dev);
    hipError_t e = hipGetLastError(); if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);     // the two-level table is a temporary of this call
    (void)hipFree(T.lo); (void)hipFree(T.hi);
    if (e != hipSuccess) return ctx->fail(STARK_ERR_HIP, "compute_powers");
    return STARK_OK;
}
extern "C" {
// compute_powers(base, n) = [1, base, ..., base^(n-1)] (also Domain::precompute_elements with base = omega); *_dev writes device memory
int32_t stark_compute_powers_dev(stark_ctx_t* ctx, int32_t field_id, const uint64_t* base4, size_t n, uint64_t* out) {
    if (!ctx || !base4 || (!out && n)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (field_id == STARK_FIELD_PALLAS_FR) return powers_run<PallasFr>(ctx, load_fr(base4), n, as_fr(out));
    if (field_id == STARK_FIELD_BLS12_381_FR) return powers_run<Bls12381Fr>(ctx, load_fr(base4), n, as_fr(out));
    return ctx->fail(STARK_ERR_INVALID_ARG, "unknown field id");
}
int32_t stark_compute_powers(stark_ctx_t* ctx, int32_t field_id, const uint64_t* base4, size_t n, uint64_t* out) {
    if (!ctx || !base4 || (!out && n)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    DevBuf d; STARK_HIP(ctx, d.alloc(ctx, n * sizeof(fr_t)));
    STARK_TRY(stark_compute_powers_dev(ctx, field_id, base4, n, (uint64_t*)d.p));
    if (n) STARK_HIP(ctx, hipMemcpyAsync(out, d.p, n * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
    STARK_HIP(ctx, hipStreamSynchronize(ctx->stream)); return STARK_OK;
}

int32_t stark_synth_column_dev(stark_ctx_t* ctx, uint64_t seed, uint64_t col, size_t i0, size_t n, uint64_t* out) {
    if (!ctx || (!out && n)) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (!n) return STARK_OK;
    hipLaunchKernelGGL(k_synth, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, seed, col, (uint64_t)i0, (uint64_t)n, as_fr(out));
    STARK_HIP(ctx, hipGetLastError()); return STARK_OK;
}

}  // extern "C"

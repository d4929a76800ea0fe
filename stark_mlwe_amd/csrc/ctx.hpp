// stark_mlwe_amd/csrc/ctx.hpp — internal objects behind the opaque C-ABI handles (product code).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <string>
#include <vector>
#include "../../include/stark_mlwe.h"
#include "fr.hpp"
#include "host_util.hpp"
#include "poseidon_params.hpp"
#include "ntt_dev.hpp"
#include "fri_plan.hpp"

namespace stark {

struct NttPlan;

}  // namespace stark

// Handles (trees, FRI states, plans, transcripts, caller-made parameter sets) keep their context alive: stark_ctx_destroy with live handles
// only marks the context; the LAST handle freed tears it down.  `CtxRef` is the FIRST member of every handle type, so it is destroyed last — after
// the handle's own destructor has returned its blocks to the context's pool.  (A Rust host drops its thread-local context at thread exit while
// Tree / ProverState values owned by the caller may outlive it — ADVICE r2.)
namespace stark { void ctx_ref(stark_ctx* c); void ctx_unref(stark_ctx* c); }
struct CtxRef {
    stark_ctx* c = nullptr;
    CtxRef() = default; CtxRef(const CtxRef&) = delete; CtxRef& operator=(const CtxRef&) = delete;
    void bind(stark_ctx* x) { if (x) stark::ctx_ref(x); if (c) stark::ctx_unref(c); c = x; }
    ~CtxRef() { if (c) stark::ctx_unref(c); }
};

struct stark_params {
    CtxRef ref_;                         // bound only for parameter sets handed to the caller (the context's own cached sets do not pin it)
    stark_ctx* ctx = nullptr;
    stark::host::PoseidonConsts ref;     // reference-form constants (as uploaded / derived)
    stark::host::KernelConsts kc;        // kernel-form constants
    stark::fr_t* blob = nullptr;         // one device allocation holding all tables
    stark::PoseidonDev dev{};            // device pointers into `blob`
};

struct stark_comm;
struct stark_ctx {
    int device = 0;
    int num_cus = 256;                               // compute units of the device (persistent NTT grids)
    stark_comm* comm = nullptr;                      // RCCL communicator spanning the ranks (capi_comm.hip); nullptr until stark_comm_init
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t side_stream = nullptr;               // lazily created: small independent jobs that run underneath a big one (fri_build)
    std::string err;
    int live_handles = 0; bool destroy_pending = false;   // see CtxRef
    std::vector<stark_ctx*> aux;                     // worker contexts on the same device (private streams) for the tails of a batch prove; created lazily, torn down with this context
    hipEvent_t ev0 = nullptr, ev1 = nullptr;         // stark_timer_start / stop
    hipEvent_t ev_fork = nullptr;                    // orders the side stream after the main one
    // lazily created constants
    stark_params* tparams = nullptr;                 // transcript params (t=17, "POSEIDON-T17-X5-TRANSCRIPT")
    std::map<int, stark_params*> merkle_params;      // poseidon_params_for_width(t)
    stark::fr_t* leaf_init = nullptr;                // 17-lane template of hash_leaf_pair (device)
    std::map<std::string, stark::fr_t*> tr_frames;   // per-tag transcript prefix/suffix frames (device): [prefix.., suffix..]
    std::map<std::string, std::pair<int, int>> tr_frame_dims;
    // scratch
    void* scratch = nullptr; size_t scratch_bytes = 0;
    // NTT plans
    std::map<uint64_t, stark::NttPlan*> plans;
    // Caching device allocator for the library's own temporaries, layers and tree levels: a block released here is reused by a
    // later request of the same (rounded) size WITHOUT hipFree / hipMalloc, so the hot path neither synchronises the device nor
    // pays allocation latency.  Safe because every use of such a block is ordered on this context's stream (the side stream
    // joins the main one before anything it touched is released).  stark_ctx_trim() returns the cached blocks to the driver.
    std::map<size_t, std::vector<void*>> pool_free;
    std::map<void*, size_t> pool_live;
    size_t pool_cached_bytes = 0;
    // host-side caches of values that depend only on their key
    struct ZKey { uint64_t seed; size_t level, size; bool operator<(const ZKey& o) const { return seed != o.seed ? seed < o.seed : (level != o.level ? level < o.level : size < o.size); } };
    std::map<ZKey, stark::fr_t> z_cache;                                // fri_sample_z_ell(seed_z, level, size)  (fri.rs:59-82)
    struct OmegaTab { int bits; stark::fr_t omega; stark::fr_t* lo; stark::fr_t* hi; int lo_bits; };
    std::vector<OmegaTab> omega_tabs;                                  // two-level power tables of a domain generator (DomainH, deep_ali/src/lib.rs:109-125)
    void* pinned = nullptr; size_t pinned_bytes = 0;                   // small pinned staging area for async uploads / downloads

    // tuning / diagnostic options (stark_ctx_set_option): explicit API state, never the environment
    int opt_ntt_direct_max_log = 24;     // direct (one-product) twiddle / coset tables for transforms up to 2^this (0 disables)
    bool opt_ntt_merged_coset = true;    // coset transforms: the pre-scale folded into the first pass's twiddle table (one table read instead of two); 0 = separate tables (comparison)
    int opt_ntt_log_tile = 11;           // log2 of the elements of an NTT tile (8..12); -1 would mean "auto" (the default also shrinks for small launches)
    bool opt_ntt_log_tile_forced = false;
    int opt_ntt_min_waves = 2;           // occupancy hint of the NTT kernels (2 or 4 waves per SIMD)
    int opt_sponge_debug = 0;            // timing experiments on the five-wave sponge (bit 0: A does not wait, 1: B idle, 2: C idle, 3: no full rounds, 4: no partial rounds); digests are WRONG when set
    bool opt_sponge_one_wave = false;    // long serial sponges on ONE wave (poseidon_coop.hpp, round 2) instead of three (poseidon_chain.hpp) (diagnostic / comparison)
    bool opt_poseidon_lane_only = false; // one-lane-per-sponge kernels instead of the wave-pair / one-wave forms (diagnostic)

    int32_t fail(int32_t code, const std::string& msg) { err = msg; return code; }
};

namespace stark {
int32_t ctx_alloc(stark_ctx* ctx, size_t bytes, void** out);   // pooled device memory (see stark_ctx::pool_free)
void ctx_release(stark_ctx* ctx, void* p);
}
struct stark_tree {
    CtxRef ref_;
    stark_ctx* ctx = nullptr; stark_params* p = nullptr;
    size_t arity = 0; uint64_t label = 0;
    std::vector<stark::fr_t*> levels; std::vector<size_t> lens; std::vector<char> owned;
    ~stark_tree() { for (size_t i = 0; i < levels.size(); ++i) if (owned[i] && levels[i]) stark::ctx_release(ctx, levels[i]); }
};


#define STARK_HIP(ctx, call)                                                                                     \
    do {                                                                                                         \
        hipError_t e__ = (call);                                                                                 \
        if (e__ != hipSuccess) return (ctx)->fail(e__ == hipErrorOutOfMemory ? STARK_ERR_OOM : STARK_ERR_HIP,     \
                                                  std::string(#call) + ": " + hipGetErrorString(e__));           \
    } while (0)
#define STARK_TRY(expr) do { int32_t rc__ = (expr); if (rc__ != STARK_OK) return rc__; } while (0)

namespace stark {

inline const fr_t* as_fr(const uint64_t* p) { return reinterpret_cast<const fr_t*>(p); }
inline fr_t* as_fr(uint64_t* p) { return reinterpret_cast<fr_t*>(p); }
inline fr_t load_fr(const uint64_t* p) { fr_t x; for (int i = 0; i < 4; ++i) { x.v[2 * i] = (uint32_t)p[i]; x.v[2 * i + 1] = (uint32_t)(p[i] >> 32); } return x; }
inline void store_fr(uint64_t* p, const fr_t& x) { for (int i = 0; i < 4; ++i) p[i] = (uint64_t)x.v[2 * i] | ((uint64_t)x.v[2 * i + 1] << 32); }

// RAII device buffer used inside entry points (returned to the context's pool on every return path).
struct DevBuf {
    stark_ctx* ctx = nullptr; void* p = nullptr;
    DevBuf() = default; DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) ctx_release(ctx, p); }
    hipError_t alloc(stark_ctx* c, size_t bytes) { ctx = c; return ctx_alloc(c, bytes, &p) == STARK_OK ? hipSuccess : hipErrorOutOfMemory; }
    fr_t* fr() const { return reinterpret_cast<fr_t*>(p); }
    void* release() { void* q = p; p = nullptr; return q; }
};

// open_union_of_paths over a device-resident tree (MerkleProofHost and the encoders: fri_plan.hpp)
int32_t merkle_open_host(stark_tree* t, const std::vector<size_t>& indices, MerkleProofHost& pr);

// shared internal entry points (defined in capi_core.hip / capi_fri.hip / capi_ntt.hip)
int32_t ctx_transcript_params(stark_ctx* ctx, stark_params** out);
int32_t ctx_merkle_params(stark_ctx* ctx, int t, stark_params** out);
int32_t ctx_scratch(stark_ctx* ctx, size_t bytes, void** out);
int32_t ctx_side_stream(stark_ctx* ctx, hipStream_t* out);
int32_t ctx_enter(stark_ctx* ctx);                                   // makes the context's device current (every entry point)
int32_t ctx_aux(stark_ctx* ctx, size_t k, stark_ctx** out);          // the k-th worker context of `ctx` (same device, private stream, the parent's options)
void comm_destroy(stark_ctx* ctx);
void ntt_set_attrs();                                                // per-device kernel attributes of the NTT kernels (capi_ntt.hip)
void ntt_plans_free(stark_ctx* ctx);
int32_t leaf_pair_hash_on(stark_ctx* ctx, hipStream_t st, const fr_t* f, const fr_t* f_next, size_t n, size_t m, fr_t* h);
int32_t merkle_build_on(stark_ctx* ctx, hipStream_t st, stark_params* p, size_t arity, uint64_t label, const fr_t* leaves, size_t n, int pairs, const fr_t* cp, size_t cp_div,
                        uint64_t first_pos, uint32_t level0, size_t stop_at_len, bool adopt, stark_tree** out);
int32_t hash_ds_scattered(stark_ctx* ctx, stark_params* p, int mode, size_t arity, size_t chunk, uint32_t level, uint64_t label, const uint64_t* positions_dev,
                          const fr_t* in0, const fr_t* in1, size_t n_hashes, fr_t* out);
int32_t tr_hash_dev(stark_ctx* ctx, const char* tag, const fr_t* fields_dev, size_t k, size_t n, fr_t* out_dev);
int32_t tr_hash_columns4_dev(stark_ctx* ctx, const char* const tags[4], const fr_t* const cols[4], size_t n0, fr_t* out4_dev);
int32_t tr_hash_columns_batch_dev(stark_ctx* ctx, const char* const tags[4], const fr_t* const* ptrs_dev, size_t batch, size_t n0, fr_t* out_dev);
int32_t tr_hash_host1(stark_ctx* ctx, const char* tag, const std::vector<fr_t>& fields, fr_t* out);   // one hash, host in/out

}  // namespace stark

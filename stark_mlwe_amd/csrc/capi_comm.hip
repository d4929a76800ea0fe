// stark_mlwe_amd/csrc/capi_comm.hip — the communicator behind the boundary: RCCL over xGMI, one process per GPU.
//
// SURVEY.md §8(b) row 1 puts the collectives of the path inside the library so that a host with no torch (the reference's Rust
// process) can drive several GPUs: rank 0 calls stark_comm_unique_id and hands the 128 bytes to its peers by whatever means the
// host has (a file, a socket, MPI, torch.distributed), every rank calls stark_comm_init on its context, and from then on
//   stark_comm_all_to_all_dev    the row/column transpose between the two local phases of the six-step NTT:
//                                one grouped ncclSend/ncclRecv pair per peer, i.e. one message per point-to-point xGMI link;
//   stark_comm_all_gather_dev    tree tops (a few digests), small layers, column digests;
//   stark_comm_all_reduce_u64_dev  the query-value table (every row is non-zero on exactly one rank, so SUM is a selection);
//   stark_comm_gather_dev        a trace column to the rank that runs its serial sponge;
// all enqueued on the CONTEXT's stream: ordered against the kernels that produce and consume the buffers, no host synchronisation.
// RCCL is bound at run time (dlopen): the library loads on a box without RCCL and fails with STARK_ERR_RCCL only when a
// communicator is asked for; when torch is in the process its bundled RCCL is reused.
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include "ctx.hpp"

using namespace stark;

namespace {
// the slice of rccl.h this file needs (declared here so that the build does not depend on the header's location)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;              // ncclSuccess == 0
enum { NCCL_INT8 = 0, NCCL_UINT8 = 1, NCCL_UINT64 = 5 };   // ncclDataType_t
enum { NCCL_SUM = 0 };                                      // ncclRedOp_t
// The ten entry points below were checked against rccl.h of ROCm 7.2 (NCCL_VERSION_CODE 22707: ncclUniqueId = 128 opaque bytes passed
// by value, ncclInt8 = 0, ncclUint8 = 1, ncclUint64 = 5, ncclSum = 0).  These have been stable through NCCL 2.x; a library that
// reports another major version is refused at load time rather than trusted (stark_comm_* then return STARK_ERR_RCCL).
constexpr int RCCL_MAJOR_CHECKED = 2;
struct Rccl {
    void* h = nullptr; bool tried = false; int version = 0;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr; ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    std::mutex mu;
    bool load() {
        std::lock_guard<std::mutex> lock(mu);          // contexts on different host threads may ask at the same time
        if (tried) return h != nullptr;
        tried = true;
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);            // a copy already in the process (torch's) wins
        const char* paths[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char* p : paths) if (!h) h = dlopen(p, RTLD_NOW | RTLD_LOCAL);
        if (!h) return false;
#define STARK_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(h, name)); if (!field) { h = nullptr; return false; }
        STARK_SYM(GetUniqueId, "ncclGetUniqueId") STARK_SYM(CommInitRank, "ncclCommInitRank") STARK_SYM(CommDestroy, "ncclCommDestroy")
        STARK_SYM(GetErrorString, "ncclGetErrorString") STARK_SYM(GroupStart, "ncclGroupStart") STARK_SYM(GroupEnd, "ncclGroupEnd")
        STARK_SYM(Send, "ncclSend") STARK_SYM(Recv, "ncclRecv") STARK_SYM(AllGather, "ncclAllGather") STARK_SYM(AllReduce, "ncclAllReduce")
        STARK_SYM(GetVersion, "ncclGetVersion")
#undef STARK_SYM
        if (GetVersion(&version) != 0 || version / 10000 != RCCL_MAJOR_CHECKED) { h = nullptr; return false; }   // NCCL_VERSION_CODE = major*10000 + minor*100 + patch
        return true;
    }
};
Rccl g_rccl;
}  // namespace

struct stark_comm { ncclComm_t comm = nullptr; int nranks = 1, rank = 0; };

#define STARK_NCCL(ctx, call)                                                                                    \
    do { ncclResult_t r__ = (call); if (r__ != 0) return (ctx)->fail(STARK_ERR_RCCL, std::string(#call) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r__) : "rccl error")); } while (0)

namespace stark { void comm_destroy(stark_ctx* ctx) { if (ctx->comm) { if (ctx->comm->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comm->comm); delete ctx->comm; ctx->comm = nullptr; } } }

extern "C" {

// Can this process bind RCCL?  Purely local (dlopen + ncclGetVersion): every rank calls it and the ranks agree on the answer BEFORE any of
// them enters the collective stark_comm_init — a rank that cannot load RCCL must not leave its peers blocked inside ncclCommInitRank.
int32_t stark_comm_available(int32_t* version_code) {
    const bool ok = g_rccl.load();
    if (version_code) *version_code = g_rccl.version;
    return ok ? STARK_OK : STARK_ERR_RCCL;
}
int32_t stark_comm_unique_id(uint8_t* id128) {
    if (!id128) return STARK_ERR_INVALID_ARG;
    if (!g_rccl.load()) return STARK_ERR_RCCL;
    ncclUniqueId id; if (g_rccl.GetUniqueId(&id) != 0) return STARK_ERR_RCCL;
    memcpy(id128, id.internal, 128); return STARK_OK;
}
int32_t stark_comm_init(stark_ctx_t* ctx, int32_t nranks, int32_t rank, const uint8_t* id128) {
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (!g_rccl.load()) return ctx->fail(STARK_ERR_RCCL, "librccl.so.1 not found (dlopen)");
    comm_destroy(ctx);
    ncclUniqueId id; memcpy(id.internal, id128, 128);
    stark_comm* c = new stark_comm(); c->nranks = nranks; c->rank = rank;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != 0) { delete c; return ctx->fail(STARK_ERR_RCCL, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); }
    ctx->comm = c; return STARK_OK;
}
int32_t stark_comm_destroy(stark_ctx_t* ctx) { if (!ctx) return STARK_ERR_INVALID_ARG; STARK_TRY(ctx_enter(ctx)); (void)hipStreamSynchronize(ctx->stream); comm_destroy(ctx); return STARK_OK; }
int32_t stark_comm_size(stark_ctx_t* ctx) { return (ctx && ctx->comm) ? ctx->comm->nranks : 0; }
int32_t stark_comm_rank(stark_ctx_t* ctx) { return (ctx && ctx->comm) ? ctx->comm->rank : -1; }

// send: nranks chunks of bytes_per_peer (chunk q goes to rank q); recv: chunk p = what rank p sent here.  send != recv.
int32_t stark_comm_all_to_all_dev(stark_ctx_t* ctx, const void* send, void* recv, size_t bytes_per_peer) {
    if (!ctx || !send || !recv || send == recv) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (!ctx->comm) return ctx->fail(STARK_ERR_RCCL, "no communicator: call stark_comm_init first");
    if (!bytes_per_peer) return STARK_OK;
    const stark_comm& c = *ctx->comm;
    STARK_NCCL(ctx, g_rccl.GroupStart());
    for (int p = 0; p < c.nranks; ++p) {        // one message per peer = one per xGMI link; RCCL schedules the group as a whole
        ncclResult_t r1 = g_rccl.Send((const char*)send + (size_t)p * bytes_per_peer, bytes_per_peer, NCCL_UINT8, p, c.comm, ctx->stream);
        ncclResult_t r2 = g_rccl.Recv((char*)recv + (size_t)p * bytes_per_peer, bytes_per_peer, NCCL_UINT8, p, c.comm, ctx->stream);
        if (r1 != 0 || r2 != 0) { (void)g_rccl.GroupEnd(); return ctx->fail(STARK_ERR_RCCL, "ncclSend/ncclRecv failed"); }
    }
    STARK_NCCL(ctx, g_rccl.GroupEnd());
    return STARK_OK;
}
// recv: nranks chunks of `bytes`, chunk p = rank p's send buffer.
int32_t stark_comm_all_gather_dev(stark_ctx_t* ctx, const void* send, void* recv, size_t bytes) {
    if (!ctx || !send || !recv) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (!ctx->comm) return ctx->fail(STARK_ERR_RCCL, "no communicator: call stark_comm_init first");
    if (!bytes) return STARK_OK;
    STARK_NCCL(ctx, g_rccl.AllGather(send, recv, bytes, NCCL_UINT8, ctx->comm->comm, ctx->stream));
    return STARK_OK;
}
// in-place capable (send == recv allowed) sum of `count` uint64 words
int32_t stark_comm_all_reduce_u64_dev(stark_ctx_t* ctx, const void* send, void* recv, size_t count) {
    if (!ctx || !send || !recv) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (!ctx->comm) return ctx->fail(STARK_ERR_RCCL, "no communicator: call stark_comm_init first");
    if (!count) return STARK_OK;
    STARK_NCCL(ctx, g_rccl.AllReduce(send, recv, count, NCCL_UINT64, NCCL_SUM, ctx->comm->comm, ctx->stream));
    return STARK_OK;
}
// every rank sends `bytes`; rank `root` receives nranks chunks in rank order (recv may be NULL elsewhere).
int32_t stark_comm_gather_dev(stark_ctx_t* ctx, const void* send, void* recv, size_t bytes, int32_t root) {
    if (!ctx || !send) return STARK_ERR_INVALID_ARG;
    STARK_TRY(ctx_enter(ctx));
    if (!ctx->comm) return ctx->fail(STARK_ERR_RCCL, "no communicator: call stark_comm_init first");
    const stark_comm& c = *ctx->comm;
    if (root < 0 || root >= c.nranks || (c.rank == root && !recv)) return ctx->fail(STARK_ERR_INVALID_ARG, "gather root");
    if (!bytes) return STARK_OK;
    STARK_NCCL(ctx, g_rccl.GroupStart());
    ncclResult_t bad = g_rccl.Send(send, bytes, NCCL_UINT8, root, c.comm, ctx->stream);
    if (c.rank == root) for (int p = 0; p < c.nranks && bad == 0; ++p) bad = g_rccl.Recv((char*)recv + (size_t)p * bytes, bytes, NCCL_UINT8, p, c.comm, ctx->stream);
    if (bad != 0) { (void)g_rccl.GroupEnd(); return ctx->fail(STARK_ERR_RCCL, "gather: ncclSend/ncclRecv failed"); }
    STARK_NCCL(ctx, g_rccl.GroupEnd());
    return STARK_OK;
}

}  // extern "C"

// The exchange step of the multi-GPU path without torch: one RCCL all-gather of the packed per-player records
// [cam(3) | R(216) | betas(10)] for hosts that drive libspsp.so directly (C / C++).  The Python mirror
// uses torch.distributed (backend "nccl" = RCCL) for the same collective; both are ONE ncclAllGather on equal-sized
// blocks -- the message is 29 KB per rank at 256 players, latency-bound, so no bucketing and no ring of small sends.
//
// RCCL is resolved at run time (dlopen): libspsp.so itself has no link-time dependency on it, and inside a PyTorch
// process the copy of librccl.so that torch already loaded is the one that gets used.
#include <dlfcn.h>

#include <cstring>

#include "common.hpp"

namespace {
using namespace spsp;

// the handful of RCCL entry points used here (signatures from <rccl/rccl.h>, ROCm 7.x)
typedef struct { char internal[128]; } UniqueId;
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, Comm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kNcclFloat = 7;          // ncclFloat32

struct Rccl {
    void* so = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllGatherFn all_gather = nullptr;
    GetErrorStringFn error_string = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            r.so = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);      // a copy this process already holds (PyTorch's)
            if (r.so) break;
        }
        for (const char* n : names) {
            if (r.so) break;
            r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (r.so) {
            r.get_unique_id = (GetUniqueIdFn)dlsym(r.so, "ncclGetUniqueId");
            r.comm_init_rank = (CommInitRankFn)dlsym(r.so, "ncclCommInitRank");
            r.comm_destroy = (CommDestroyFn)dlsym(r.so, "ncclCommDestroy");
            r.all_gather = (AllGatherFn)dlsym(r.so, "ncclAllGather");
            r.error_string = (GetErrorStringFn)dlsym(r.so, "ncclGetErrorString");
        }
    }
    const bool ok = r.so && r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_gather;
    return ok ? &r : nullptr;
}

#define SPSP_NCCL_CHECK(call)                                                                             \
    do {                                                                                                  \
        const int e_ = (call);                                                                            \
        if (e_ != 0) {                                                                                    \
            Rccl* rr_ = rccl();                                                                           \
            set_error("%s:%d %s -> RCCL error %d (%s)", __FILE__, __LINE__, #call, e_,                    \
                      (rr_ && rr_->error_string) ? rr_->error_string(e_) : "?");                          \
            return SPSP_ENCCL;                                                                            \
        }                                                                                                 \
    } while (0)

}  // namespace

struct spsp_comm {
    Comm comm = nullptr;
    int nranks = 0, rank = 0;
};

extern "C" {

int spsp_comm_unique_id(void* id128) {
    SPSP_REQUIRE(id128, "spsp_comm_unique_id: NULL buffer");
    Rccl* r = rccl();
    SPSP_REQUIRE(r, "spsp_comm_unique_id: librccl.so could not be loaded");
    UniqueId id;
    SPSP_NCCL_CHECK(r->get_unique_id(&id));
    std::memcpy(id128, &id, sizeof id);
    return SPSP_OK;
}

int spsp_comm_init_rank(spsp_ctx* ctx, const void* id128, int nranks, int rank, spsp_comm** out) {
    SPSP_REQUIRE(ctx && id128 && out, "spsp_comm_init_rank: NULL argument");
    SPSP_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "spsp_comm_init_rank: rank %d of %d", rank, nranks);
    Rccl* r = rccl();
    SPSP_REQUIRE(r, "spsp_comm_init_rank: librccl.so could not be loaded");
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));
    UniqueId id;
    std::memcpy(&id, id128, sizeof id);
    spsp_comm* c = new spsp_comm;
    c->nranks = nranks;
    c->rank = rank;
    const int e = r->comm_init_rank(&c->comm, nranks, id, rank);
    if (e != 0) {
        delete c;
        set_error("spsp_comm_init_rank: ncclCommInitRank -> RCCL error %d (%s)", e, r->error_string ? r->error_string(e) : "?");
        return SPSP_ENCCL;
    }
    *out = c;
    return SPSP_OK;
}

void spsp_comm_destroy(spsp_comm* c) {
    if (!c) return;
    Rccl* r = rccl();
    if (r && c->comm) (void)r->comm_destroy(c->comm);
    delete c;
}

int spsp_allgather_records(spsp_ctx* ctx, spsp_comm* c, const float* local, int n_per_rank, float* all, void* stream) {
    SPSP_REQUIRE(ctx && c && c->comm, "spsp_allgather_records: NULL communicator");
    SPSP_REQUIRE(n_per_rank >= 0, "spsp_allgather_records: negative count");
    if (n_per_rank == 0) return SPSP_OK;
    SPSP_REQUIRE(local && all, "spsp_allgather_records: NULL buffer");
    Rccl* r = rccl();
    SPSP_REQUIRE(r, "spsp_allgather_records: librccl.so could not be loaded");
    SPSP_NCCL_CHECK(r->all_gather(local, all, (size_t)n_per_rank * SPSP_RECORD_FLOATS, kNcclFloat, c->comm, as_stream(stream)));
    return SPSP_OK;
}

}  // extern "C"

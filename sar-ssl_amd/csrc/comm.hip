// Gradient-bucket all-reduce over RCCL behind the C ABI (include/sarssl_hip.h, "collectives").
//
// The reference trains data-parallel through torch.nn.DataParallel (code/learner.py:25-31, :102): replicate, scatter, gather, reduce on
// device 0, every step.  Here every rank owns one GPU and the flat f32 gradient buffer is summed bucket by bucket while backward still
// runs (sar_ssl_amd/dist.py).  The default host path issues those all-reduces through torch.distributed; these entry points are the same
// exchange for a host that does not carry torch: a communicator per (process, device), created from an id the caller moved between the
// ranks by its own means, and an in-place f32 sum of one contiguous bucket enqueued on the stream the caller names.
//
// RCCL is resolved at first use with dlopen / dlsym - the library has no link-time dependency on it (a process that never trains on
// more than one GPU never loads it, and inside a torch process the already-loaded librccl.so.1 is the one that is found).
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {
// (the few declarations of <rccl/rccl.h> this file needs, so that building the library does not need RCCL's headers either)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*CommCountFn)(const Comm, int*);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);
typedef int (*GetVersionFn)(int*);
constexpr int kFloat32 = 7, kSum = 0;              // ncclFloat32, ncclSum

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    CommCountFn comm_count = nullptr;
    AllReduceFn all_reduce = nullptr;
    GetErrorStringFn error_string = nullptr;
    GetVersionFn get_version = nullptr;
    bool ok = false;
};

Rccl* rccl() {
    static Rccl r;                                  // C++11: initialised once, thread-safe
    static const bool tried = [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the copy this process already maps (torch's), if any
            if (r.handle) break;
        }
        for (int i = 0; !r.handle && i < 3; ++i) r.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) return true;
        r.get_unique_id = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
        r.comm_init_rank = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
        r.comm_destroy = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
        r.comm_count = (CommCountFn)dlsym(r.handle, "ncclCommCount");
        r.all_reduce = (AllReduceFn)dlsym(r.handle, "ncclAllReduce");
        r.error_string = (GetErrorStringFn)dlsym(r.handle, "ncclGetErrorString");
        r.get_version = (GetVersionFn)dlsym(r.handle, "ncclGetVersion");
        r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce;
        return true;
    }();
    (void)tried;
    return &r;
}

int fail(const char* what, int rc) {
    Rccl* r = rccl();
    sarssl_set_error("%s: RCCL error %d (%s)", what, rc, (r->error_string ? r->error_string(rc) : "?"));
    return -3;
}
}  // namespace

// 1 when an RCCL library could be resolved in this process (no GPU needed for the answer).
extern "C" int sarssl_comm_available() { return rccl()->ok ? 1 : 0; }

// RCCL's version code (e.g. 22203), or -1.
extern "C" int sarssl_comm_rccl_version() {
    Rccl* r = rccl();
    int v = -1;
    if (!r->ok || !r->get_version || r->get_version(&v) != 0) return -1;
    return v;
}

// id128: 128 bytes the caller distributes to every rank (rank 0 generates it).
extern "C" int sarssl_comm_unique_id(void* id128) {
    Rccl* r = rccl();
    SARSSL_REQUIRE(id128 != nullptr, "sarssl_comm_unique_id");
    if (!r->ok) { sarssl_set_error("sarssl_comm_unique_id: no RCCL library (librccl.so.1) could be loaded"); return -3; }
    UniqueId id;
    const int rc = r->get_unique_id(&id);
    if (rc != 0) return fail("ncclGetUniqueId", rc);
    memcpy(id128, &id, sizeof(id));
    return 0;
}

// Communicator of `rank` among `nranks` on the CURRENT HIP device (collective call: every rank enters with the same id).  Null on failure.
extern "C" void* sarssl_comm_create(int nranks, int rank, const void* id128) {
    Rccl* r = rccl();
    if (!r->ok) { sarssl_set_error("sarssl_comm_create: no RCCL library (librccl.so.1) could be loaded"); return nullptr; }
    if (nranks < 1 || rank < 0 || rank >= nranks || !id128) { sarssl_set_error("sarssl_comm_create: bad arguments"); return nullptr; }
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    Comm c = nullptr;
    const int rc = r->comm_init_rank(&c, nranks, id, rank);
    if (rc != 0) { fail("ncclCommInitRank", rc); return nullptr; }
    return c;
}

extern "C" int sarssl_comm_destroy(void* comm) {
    Rccl* r = rccl();
    if (!comm) return 0;
    if (!r->ok) return -3;
    const int rc = r->comm_destroy((Comm)comm);
    return rc == 0 ? 0 : fail("ncclCommDestroy", rc);
}

extern "C" int sarssl_comm_size(void* comm) {
    Rccl* r = rccl();
    int n = -1;
    if (!comm || !r->ok || !r->comm_count || r->comm_count((Comm)comm, &n) != 0) return -1;
    return n;
}

// In-place sum over the communicator's ranks of one contiguous f32 gradient bucket, enqueued on `stream` (asynchronous; capturable).
// The 1/world scaling is not applied here: it is folded into the fused Adam kernel (sarssl_adam_step*: gscale).
extern "C" int sarssl_allreduce_bucket(void* comm, float* bucket, long count, void* stream) {
    Rccl* r = rccl();
    SARSSL_REQUIRE(comm != nullptr && bucket != nullptr && count > 0, "sarssl_allreduce_bucket");
    if (!r->ok) { sarssl_set_error("sarssl_allreduce_bucket: no RCCL library"); return -3; }
    const int rc = r->all_reduce(bucket, bucket, (size_t)count, kFloat32, kSum, (Comm)comm, (hipStream_t)stream);
    return rc == 0 ? 0 : fail("ncclAllReduce", rc);
}

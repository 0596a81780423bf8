// comm.hip -- RCCL behind the C ABI (oi_comm_*, oi_index_finalize_sharded, oi_search_sharded; api.hip has the entry points).
//
// librccl is loaded on first use with dlopen("librccl.so.1"): a host that never shards does not need RCCL installed, and
// a process that already mapped an RCCL (PyTorch ships one under the same soname) gets THAT copy back from the dynamic
// linker -- two RCCLs in one process would each bring their own bootstrap state.  No compile-time dependency either:
// the handful of prototypes below is the whole surface used (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclAllGather,
// ncclAllReduce, ncclCommDestroy, ncclGetErrorString).
#include <dlfcn.h>

#include "oi_internal.h"

namespace {

struct UniqueId { char internal[OI_COMM_ID_BYTES]; }; // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
static_assert(sizeof(UniqueId) == 128, "ncclUniqueId is 128 bytes");
// ncclDataType_t / ncclRedOp_t values used (rccl.h): ncclUint32 = 3, ncclUint64 = 5, ncclSum = 0
enum { kUint32 = 3, kUint64 = 5, kSum = 0 };

struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // OI_RCCL_LIB names the library to load instead (a deployment with RCCL outside the loader's path; the tests use it
        // to take the "no RCCL on this host" path).  dlerror() hands its message out ONCE and clears it: read it once.
        const char *override_name = getenv("OI_RCCL_LIB");
        std::string why;
        for (const char *name : {override_name, override_name ? nullptr : "librccl.so.1", override_name ? nullptr : "librccl.so"}) {
            if (!name) continue;
            r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
            const char *e = dlerror();
            if (why.empty()) why = e ? e : "dlopen failed";
        }
        if (!r.h) { r.why = why.empty() ? "dlopen failed" : why; return; }
        auto sym = [&](const char *n) -> void * {
            void *p = dlsym(r.h, n);
            if (!p && r.why.empty()) r.why = std::string("missing symbol ") + n;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &r;
}

int ready(Rccl **out) {
    Rccl *r = rccl();
    if (!r->h || !r->why.empty()) {
        oi_set_error("RCCL is not usable in this process: %s", r->why.c_str());
        return OI_ERR_UNSUPPORTED;
    }
    *out = r;
    return OI_OK;
}

#define OI_RCCL_CHECK(r, expr)                                                                     \
    do {                                                                                           \
        int oi_n_ = (expr);                                                                        \
        if (oi_n_ != 0) {                                                                          \
            oi_set_error("%s failed: %s (%s:%d)", #expr, (r)->GetErrorString(oi_n_), __FILE__, __LINE__); \
            return OI_ERR_COMM;                                                                    \
        }                                                                                          \
    } while (0)

} // namespace

int oi_rccl_unique_id(uint8_t *id_out) {
    Rccl *r;
    OI_CHECK(ready(&r));
    UniqueId id;
    OI_RCCL_CHECK(r, r->GetUniqueId(&id));
    memcpy(id_out, id.internal, OI_COMM_ID_BYTES);
    return OI_OK;
}

int oi_rccl_init(void **comm_out, uint32_t world, const uint8_t *id_bytes, uint32_t rank) {
    Rccl *r;
    OI_CHECK(ready(&r));
    UniqueId id;
    memcpy(id.internal, id_bytes, OI_COMM_ID_BYTES);
    OI_RCCL_CHECK(r, r->CommInitRank(comm_out, (int)world, id, (int)rank));
    return OI_OK;
}

void oi_rccl_destroy(void *comm) {
    Rccl *r = rccl();
    if (comm && r->CommDestroy) (void)r->CommDestroy(comm);
}

int oi_rccl_all_gather_u32(void *comm, const uint32_t *send, uint32_t *recv, size_t words, hipStream_t st) {
    Rccl *r;
    OI_CHECK(ready(&r));
    OI_RCCL_CHECK(r, r->AllGather(send, recv, words, kUint32, comm, st));
    return OI_OK;
}

int oi_rccl_all_reduce_sum(void *comm, void *buf, size_t count, bool u64, hipStream_t st) {
    Rccl *r;
    OI_CHECK(ready(&r));
    OI_RCCL_CHECK(r, r->AllReduce(buf, buf, count, u64 ? kUint64 : kUint32, kSum, comm, st));
    return OI_OK;
}

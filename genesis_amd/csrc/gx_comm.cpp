// The step's one collective behind the C ABI: a sum all-reduce of the flat fp32 gradient bucket over
// RCCL, for a host that is not PyTorch.  RCCL is resolved at run time (dlopen + dlsym) so that libgenesis_hip.so itself has no
// link-time dependency on it: a process that already holds an RCCL (PyTorch bundles one) shares that copy -- two RCCL runtimes
// in one process would each open their own xGMI / IPC state.
// Reference: train.py:153-155 (nn.DataParallel gathers the replicas' gradients on GPU 0); here one process per GPU and ONE
// in-place ncclAllReduce per step, enqueued on the caller's stream (so it can be captured into the step's HIP graph).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include <rccl/rccl.h>        // types and enums only; no symbol of it is linked

#include "gx_common.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    int (*dummy)() = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

// 0 on success; sets the error message otherwise
int rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return GX_OK;
    const char* env = getenv("GENESIS_RCCL_LIB");
    void* h = nullptr;
    if (env && *env) {
        h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        GX_CHECK_ARG(h, "gx_allreduce: cannot load GENESIS_RCCL_LIB=%s: %s", env, dlerror());
    }
    // a copy the process already mapped first (PyTorch's), then the system one
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !h; ++pass)
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) break;
        }
    GX_CHECK_ARG(h, "gx_allreduce: librccl.so not found (set GENESIS_RCCL_LIB): %s", dlerror());
#define GX_SYM(field, name)                                                                     \
    *(void**)(&g_rccl.field) = dlsym(h, name);                                                  \
    GX_CHECK_ARG(g_rccl.field, "gx_allreduce: %s missing from the RCCL library", name)
    GX_SYM(GetUniqueId, "ncclGetUniqueId");
    GX_SYM(CommInitRank, "ncclCommInitRank");
    GX_SYM(AllReduce, "ncclAllReduce");
    GX_SYM(CommDestroy, "ncclCommDestroy");
    GX_SYM(GetErrorString, "ncclGetErrorString");
#undef GX_SYM
    g_rccl.handle = h;
    return GX_OK;
}

struct GxComm { ncclComm_t comm; int rank, world; };

#define GX_CHECK_NCCL(call, what)                                                               \
    do {                                                                                        \
        ncclResult_t r__ = (call);                                                              \
        if (r__ != ncclSuccess) {                                                               \
            gx_set_error("%s: RCCL error %d: %s", what, (int)r__, g_rccl.GetErrorString(r__));  \
            return GX_ELAUNCH;                                                                  \
        }                                                                                       \
    } while (0)

}  // namespace

extern "C" {

size_t gx_allreduce_unique_id_bytes(void) { return sizeof(ncclUniqueId); }

int gx_allreduce_unique_id(void* id, size_t id_bytes) {
    GX_CHECK_ARG(id && id_bytes >= sizeof(ncclUniqueId), "gx_allreduce_unique_id: id buffer of >= %zu bytes", sizeof(ncclUniqueId));
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId u;
    GX_CHECK_NCCL(g_rccl.GetUniqueId(&u), "gx_allreduce_unique_id");
    memcpy(id, &u, sizeof(u));
    return GX_OK;
}

int gx_allreduce_init(const void* id, size_t id_bytes, int rank, int world, void** comm) {
    GX_CHECK_ARG(id && comm && id_bytes >= sizeof(ncclUniqueId), "gx_allreduce_init: id of >= %zu bytes, comm", sizeof(ncclUniqueId));
    GX_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "gx_allreduce_init: rank %d of %d", rank, world);
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    GxComm* c = new GxComm{nullptr, rank, world};
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);       // (the calling thread's current HIP device)
    if (r != ncclSuccess) {
        gx_set_error("gx_allreduce_init: RCCL error %d: %s", (int)r, g_rccl.GetErrorString(r));
        delete c;
        return GX_ELAUNCH;
    }
    *comm = c;
    return GX_OK;
}

int gx_allreduce_run(void* comm, float* buf, size_t count, gx_stream_t stream) {
    GX_CHECK_ARG(comm && buf, "gx_allreduce_run: null pointer");
    if (count == 0) return GX_OK;
    GxComm* c = (GxComm*)comm;
    GX_CHECK_NCCL(g_rccl.AllReduce(buf, buf, count, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream), "gx_allreduce_run");
    return GX_OK;
}

int gx_allreduce_destroy(void* comm) {
    if (!comm) return GX_OK;
    GxComm* c = (GxComm*)comm;
    ncclResult_t r = g_rccl.CommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) {
        gx_set_error("gx_allreduce_destroy: RCCL error %d: %s", (int)r, g_rccl.GetErrorString(r));
        return GX_ELAUNCH;
    }
    return GX_OK;
}

}  // extern "C"

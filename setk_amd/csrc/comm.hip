// comm.hip -- the work-queue barrier of a multi-GPU run, on RCCL, behind the C ABI.
//
// The reference shards this path over OS processes with no communication at all
// (scripts/run_adapt_beamformer.sh:69-92: split_scp.pl + run.pl JOB=1:nj).  One process per GPU
// needs exactly two things from its peers: "everybody has started / finished" and the sums of
// the "Processed N utterances" counters.  Both are one 8-byte-per-value all-reduce over xGMI
// (latency bound: ring or tree does not matter at this size).  Rounds 1 - 4 reached RCCL only
// through torch.distributed, which cost every rank of a multi-rank command line 1 - 2 s of
// `import torch`; this unit talks to librccl itself.  librccl is loaded on first use (dlopen),
// so a single-GPU run never maps it.
//
// Rendezvous: ncclGetUniqueId on rank 0, its 128 bytes carried to the other ranks by the caller
// (setk_amd/dist.py: a TCP socket on MASTER_ADDR:MASTER_PORT, the variables every launcher of
// torch.distributed.run style exports), then ncclCommInitRank on every rank.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/setk_hip.h"

namespace {
typedef int ncclResult_t;              // ncclSuccess == 0
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
constexpr int kNcclSum = 0, kNcclMax = 2, kNcclFloat64 = 8;  // rccl.h: ncclRedOp_t, ncclDataType_t

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
thread_local std::string g_err;

int fail(const std::string& what) {
    g_err = what;
    return SETK_ERR_HIP;
}

bool load_rccl() {
    if (g_rccl.so) return true;
    // SETK_RCCL_LIB names the library instead (a deployment with its own build; the tests point it
    // at a missing file to exercise the callers' TCP fallback)
    const char* forced = getenv("SETK_RCCL_LIB");
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* so = nullptr;
    std::string why;
    if (forced && *forced) {
        so = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!so) { const char* e = dlerror(); why = e ? e : "?"; }   // (dlerror() clears on read: once)
    } else {
        for (const char* n : names) {
            so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (so) break;
            const char* e = dlerror();
            why = e ? e : "?";
        }
    }
    if (!so) {
        g_err = "librccl not found: " + why;
        return false;
    }
    Rccl r;
    r.so = so;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(so, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(so, "ncclCommInitRank"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(so, "ncclAllReduce"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(so, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(so, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) {
        g_err = "librccl lacks an expected symbol";
        dlclose(so);
        return false;
    }
    g_rccl = r;
    return true;
}

std::string nccl_text(ncclResult_t rc) {
    char buf[64];
    snprintf(buf, sizeof buf, "rccl error %d", rc);
    std::string s = buf;
    if (g_rccl.GetErrorString) s += std::string(": ") + g_rccl.GetErrorString(rc);
    return s;
}
}  // namespace

struct setk_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double* d_buf = nullptr;  // kMaxValues doubles
};
constexpr int kMaxValues = 64;

extern "C" {

const char* setk_comm_last_error(void) { return g_err.c_str(); }

int setk_comm_unique_id(char out[SETK_COMM_ID_BYTES]) {
    if (!out) return SETK_ERR_INVALID;
    if (!load_rccl()) return SETK_ERR_UNSUPPORTED;
    ncclUniqueId id;
    const ncclResult_t rc = g_rccl.GetUniqueId(&id);
    if (rc) return fail("ncclGetUniqueId: " + nccl_text(rc));
    memcpy(out, id.internal, sizeof(id.internal));
    return SETK_OK;
}

int setk_comm_create(setk_comm_t* out, int device_ordinal, const char id[SETK_COMM_ID_BYTES], int rank,
                     int world) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return SETK_ERR_INVALID;
    *out = nullptr;
    if (!load_rccl()) return SETK_ERR_UNSUPPORTED;
    if (hipSetDevice(device_ordinal) != hipSuccess) return fail("hipSetDevice failed");
    setk_comm* c = new setk_comm();
    c->device = device_ordinal;
    c->rank = rank;
    c->world = world;
    ncclUniqueId uid;
    memcpy(uid.internal, id, sizeof(uid.internal));
    ncclResult_t rc = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (rc) {
        delete c;
        return fail("ncclCommInitRank: " + nccl_text(rc));
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_buf), kMaxValues * sizeof(double)) != hipSuccess) {
        setk_comm_destroy(c);
        return fail("stream / buffer for the collectives");
    }
    *out = c;
    return SETK_OK;
}

int setk_comm_allreduce_f64(setk_comm_t c, double* values, int n, int op) {
    if (!c || !values || n < 1 || n > kMaxValues || (op != SETK_COMM_SUM && op != SETK_COMM_MAX))
        return SETK_ERR_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return fail("hipSetDevice failed");
    if (hipMemcpyAsync(c->d_buf, values, n * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess)
        return fail("upload of the values");
    const ncclResult_t rc = g_rccl.AllReduce(c->d_buf, c->d_buf, (size_t)n, kNcclFloat64,
                                             op == SETK_COMM_MAX ? kNcclMax : kNcclSum, c->comm, c->stream);
    if (rc) return fail("ncclAllReduce: " + nccl_text(rc));
    if (hipMemcpyAsync(values, c->d_buf, n * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return fail("download of the result");
    return SETK_OK;
}

int setk_comm_barrier(setk_comm_t c) {
    double one = 1.0;
    return setk_comm_allreduce_f64(c, &one, 1, SETK_COMM_SUM);
}

int setk_comm_destroy(setk_comm_t c) {
    if (!c) return SETK_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SETK_OK;
}

}  // extern "C"

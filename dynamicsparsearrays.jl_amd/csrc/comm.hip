// csrc/comm.hip — the one data-path collective of the column-range sharded SpMV behind the C ABI (include/dsa.h: dsa_comm_*):
// an RCCL all-reduce (sum, Float64) of the partial y over the ranks, one process per GPU, over xGMI inside a node.
//
// The reference is single-process and has no counterpart.  RCCL is bound at RUN time (dlopen): libdsa_hip.so has no link-time
// dependency on it, a single-GPU user never loads it, and a host process that already carries an RCCL (PyTorch-ROCm ships its
// own librccl.so) keeps using that copy — two RCCL instances in one process would each want their own bootstrap threads and
// IPC handles.  Only the five entry points below are used; their prototypes are the public ones of <rccl/rccl.h>.
#include "../../include/dsa.h"
#include "dsa_dev.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>

namespace {

constexpr int RCCL_ID_BYTES = 128;                       // NCCL_UNIQUE_ID_BYTES
struct RcclUniqueId { char internal[RCCL_ID_BYTES]; };   // ncclUniqueId
typedef void* RcclComm;                                   // ncclComm_t
constexpr int RCCL_DOUBLE = 8, RCCL_SUM = 0;              // ncclFloat64, ncclSum

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("DSA_RCCL_LIB");
        // a copy that is already mapped into the process (e.g. PyTorch's) first, then the ROCm installation
        // DSA_RCCL_LIB names THE library to use (nothing else is tried: an override that silently falls back is not an override)
        const bool only_env = env && env[0];
        const char* names[] = {env, only_env ? nullptr : "librccl.so", only_env ? nullptr : "librccl.so.1",
                               only_env ? nullptr : "/opt/rocm/lib/librccl.so.1", only_env ? nullptr : "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            if (!n || !n[0]) continue;
            R.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (R.lib) break;
        }
        for (const char* n : names) {
            if (R.lib) break;
            if (!n || !n[0]) continue;
            R.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!R.lib) {
            const char* de = dlerror();          // (a second call returns NULL: the first one clears the error)
            R.why = std::string("librccl.so not found: ") + (de ? de : "");
            return;
        }
        R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(dlsym(R.lib, "ncclGetUniqueId"));
        R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(dlsym(R.lib, "ncclCommInitRank"));
        R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(dlsym(R.lib, "ncclCommDestroy"));
        R.AllReduce = reinterpret_cast<decltype(R.AllReduce)>(dlsym(R.lib, "ncclAllReduce"));
        R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(dlsym(R.lib, "ncclGetErrorString"));
        if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllReduce) { R.why = "librccl.so lacks the nccl* entry points"; R.lib = nullptr; }
    });
    return R;
}

thread_local std::string g_comm_err;
int32_t comm_fail(int32_t code, const std::string& msg) { g_comm_err = msg; dsa::set_last_error(msg.c_str()); return code; }
int32_t rccl_fail(const char* what, int rc) {
    Rccl& R = rccl();
    return comm_fail(DSA_ERCCL, std::string(what) + ": " + (R.GetErrorString ? R.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}

}  // namespace

struct dsa_comm { RcclComm comm = nullptr; int rank = 0, nranks = 1, device = 0; };

extern "C" {

int32_t dsa_comm_unique_id(uint8_t id[DSA_COMM_ID_BYTES]) {
    static_assert(DSA_COMM_ID_BYTES == RCCL_ID_BYTES, "the id crossing the ABI is RCCL's ncclUniqueId");
    Rccl& R = rccl();
    if (!R.lib) return comm_fail(DSA_ERCCL, R.why);
    RcclUniqueId u;
    const int rc = R.GetUniqueId(&u);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id, u.internal, RCCL_ID_BYTES);
    return DSA_OK;
}

int32_t dsa_comm_init(int32_t rank, int32_t nranks, const uint8_t id[DSA_COMM_ID_BYTES], dsa_comm_t** out) {
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return comm_fail(DSA_EARG, "rank / nranks out of range");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return comm_fail(DSA_EHIP, "no current device");
    auto* c = new dsa_comm();
    c->rank = rank; c->nranks = nranks; c->device = dev;
    if (nranks > 1 || id != nullptr) {                 // (a single rank WITH an id still gets a real RCCL communicator: smoke tests)
        if (id == nullptr) { delete c; return comm_fail(DSA_EARG, "more than one rank needs the unique id of rank 0"); }
        Rccl& R = rccl();
        if (!R.lib) { delete c; return comm_fail(DSA_ERCCL, R.why); }
        RcclUniqueId u;
        std::memcpy(u.internal, id, RCCL_ID_BYTES);
        const int rc = R.CommInitRank(&c->comm, nranks, u, rank);      // collective over the ranks: every process calls it
        if (rc != 0) { delete c; return rccl_fail("ncclCommInitRank", rc); }
    }
    *out = c;
    return DSA_OK;
}

int32_t dsa_comm_destroy(dsa_comm_t* c) {
    if (!c) return DSA_OK;
    int32_t st = DSA_OK;
    if (c->comm) {
        const int rc = rccl().CommDestroy(c->comm);
        if (rc != 0) st = rccl_fail("ncclCommDestroy", rc);
    }
    delete c;
    return st;
}

int32_t dsa_comm_info(dsa_comm_t* c, int32_t* rank, int32_t* nranks) {
    if (!c) return comm_fail(DSA_EARG, "null communicator");
    *rank = c->rank; *nranks = c->nranks;
    return DSA_OK;
}

// in place: y <- sum over the ranks of y (m doubles in HBM), asynchronous on `hip_stream`; a single rank has nothing to add
int32_t dsa_shard_allreduce_dev(dsa_comm_t* c, double* d_y, int64_t m, void* hip_stream) {
    if (!c || m < 0) return comm_fail(DSA_EARG, "null communicator or negative length");
    if (c->comm == nullptr || m == 0) return DSA_OK;
    const int rc = rccl().AllReduce(d_y, d_y, (size_t)m, RCCL_DOUBLE, RCCL_SUM, c->comm, static_cast<hipStream_t>(hip_stream));
    if (rc != 0) return rccl_fail("ncclAllReduce", rc);
    return DSA_OK;
}

}  // extern "C"

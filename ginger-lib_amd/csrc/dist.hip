// dist.hip -- include/ginger_hip_dist.h: all-gather of the per-GPU partial sums + fold.
// RCCL is bound at run time (dlopen) so that libginger_hip.so carries no link-time dependency on it.
#include <dlfcn.h>
#include <string.h>
#include <chrono>
#include <mutex>
#include <vector>
#include <rccl/rccl.h>
#include "runtime.h"
#include "../../include/ginger_hip_dist.h"

namespace {
using namespace gh_rt;

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct DistCtx {
    bool ready = false;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;        // RCCL transport
    gh_allgather_fn fn = nullptr;     // custom transport
    void* fn_ctx = nullptr;
    uint64_t *d_send = nullptr, *d_recv = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

Rccl R;
DistCtx D;
std::mutex d_mu;

int load_rccl() {
    if (R.lib) return GH_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { g_err = std::string("cannot load librccl: ") + dlerror(); return GH_E_DIST; }
#define SYM(field, name)                                                   \
    *reinterpret_cast<void**>(&R.field) = dlsym(h, name);                  \
    if (!R.field) { g_err = "librccl lacks " name; dlclose(h); return GH_E_DIST; }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(CommCount, "ncclCommCount")
    SYM(AllGather, "ncclAllGather")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    R.lib = h;
    return GH_OK;
}

#define NCCLCHK(call)                                                                      \
    do {                                                                                   \
        ncclResult_t r_ = (call);                                                          \
        if (r_ != ncclSuccess) { g_err = std::string(#call " failed: ") + R.GetErrorString(r_); return GH_E_DIST; } \
    } while (0)

int deg_of(gh_curve_t c) { return c == GH_MNT4753_G2 ? 2 : (c == GH_MNT6753_G2 ? 3 : 1); }
}  // namespace

extern "C" {

int gh_dist_unique_id(void* out_id128) {
    std::lock_guard<std::mutex> lk(d_mu);
    if (!out_id128) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    NCCLCHK(R.GetUniqueId(&id));
    static_assert(sizeof(id) == GH_DIST_UNIQUE_ID_BYTES, "unique id size");
    memcpy(out_id128, &id, sizeof id);
    return GH_OK;
}

int gh_dist_init_rccl(const void* id128, int rank, int world) {
    int rc = gh_init(nullptr, 0);          // binds the device (no-op if the host already called gh_init)
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(d_mu);
    if (D.ready) { g_err = "a communicator already exists"; return GH_E_BAD_ARG; }
    if (!id128 || world < 1 || rank < 0 || rank >= world) { g_err = "bad rank / world / id"; return GH_E_BAD_ARG; }
    if ((rc = load_rccl())) return rc;
    HIPCHK(hipSetDevice(g.device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    NCCLCHK(R.CommInitRank(&D.comm, world, id, rank));
    int cnt = 0;
    NCCLCHK(R.CommCount(D.comm, &cnt));
    D.rank = rank; D.world = cnt;
    HIPCHK(hipMalloc((void**)&D.d_send, 108 * 8));
    HIPCHK(hipMalloc((void**)&D.d_recv, (size_t)108 * 8 * cnt));
    HIPCHK(hipEventCreate(&D.ev0));
    HIPCHK(hipEventCreate(&D.ev1));
    D.ready = true;
    return GH_OK;
}

int gh_dist_init_custom(gh_allgather_fn fn, void* ctx, int rank, int world) {
    std::lock_guard<std::mutex> lk(d_mu);
    if (D.ready) { g_err = "a communicator already exists"; return GH_E_BAD_ARG; }
    if (!fn || world < 1 || rank < 0 || rank >= world) { g_err = "bad rank / world / callback"; return GH_E_BAD_ARG; }
    D.fn = fn; D.fn_ctx = ctx; D.rank = rank; D.world = world; D.comm = nullptr;
    D.ready = true;
    return GH_OK;
}

int gh_dist_info(int* rank, int* world) {
    std::lock_guard<std::mutex> lk(d_mu);
    if (!D.ready) { g_err = "no communicator (gh_dist_init_*)"; return GH_E_DIST; }
    if (rank) *rank = D.rank;
    if (world) *world = D.world;
    return GH_OK;
}

int gh_partials_allgather_fold(gh_curve_t curve, const uint64_t* partial_xyz, uint64_t* out_xyz, double* exchange_us) {
    std::lock_guard<std::mutex> lk(d_mu);
    if (!D.ready) { g_err = "no communicator (gh_dist_init_*)"; return GH_E_DIST; }
    if (!partial_xyz || !out_xyz || (int)curve < 0 || (int)curve > 3) { g_err = "bad argument"; return GH_E_BAD_ARG; }
    const size_t words = (size_t)36 * deg_of(curve);
    std::vector<uint64_t> all(words * (size_t)D.world);
    const auto t0 = std::chrono::steady_clock::now();
    if (D.comm) {
        hipStream_t st = g.stream;
        HIPCHK(hipMemcpyAsync(D.d_send, partial_xyz, words * 8, hipMemcpyHostToDevice, st));
        NCCLCHK(R.AllGather(D.d_send, D.d_recv, words, ncclUint64, D.comm, st));
        HIPCHK(hipMemcpyAsync(all.data(), D.d_recv, words * 8 * D.world, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    } else {
        if (D.fn(D.fn_ctx, partial_xyz, all.data(), words * 8) != 0) { g_err = "custom all-gather failed"; return GH_E_DIST; }
    }
    if (exchange_us) *exchange_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    std::vector<uint64_t> acc(all.begin(), all.begin() + words);
    for (int r = 1; r < D.world; r++) {
        int rc = gh_proj_add(curve, acc.data(), all.data() + words * (size_t)r);
        if (rc) return rc;
    }
    memcpy(out_xyz, acc.data(), words * 8);
    return GH_OK;
}

int gh_dist_shutdown(void) {
    std::lock_guard<std::mutex> lk(d_mu);
    if (!D.ready) return GH_OK;
    if (D.comm) {
        R.CommDestroy(D.comm);
        if (D.d_send) (void)hipFree(D.d_send);
        if (D.d_recv) (void)hipFree(D.d_recv);
        if (D.ev0) (void)hipEventDestroy(D.ev0);
        if (D.ev1) (void)hipEventDestroy(D.ev1);
    }
    D = DistCtx();
    return GH_OK;
}

}  // extern "C"

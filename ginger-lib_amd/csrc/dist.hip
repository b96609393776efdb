// dist.hip -- include/ginger_hip_dist.h: all-gather of the per-GPU partial sums + fold.
// RCCL is bound at run time (dlopen) so that libginger_hip.so carries no link-time dependency on it.
//
// Locking: every entry point takes the library's API lock first (api_mutex(): the device context, g_err, the library
// stream) and the communicator's lock second -- always in that order; gh_shutdown, which already holds the API lock,
// tears the communicator down through dist_teardown_locked().
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>
#include <rccl/rccl.h>
#include "runtime.h"
#include "../../include/ginger_hip_dist.h"

namespace {
using namespace gh_rt;

struct Rccl {
    void* lib = nullptr;
    std::string path;          // the file the symbols come from (dladdr)
    int version = 0;           // ncclGetVersion
    bool was_mapped = false;   // an RCCL was already in the process (e.g. torch's) and is the one in use
    std::string note;          // why a mapped copy was passed over (major version other than the build headers')
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

constexpr size_t MAX_WORDS = 108;      // 36 * deg u64 per partial sum (MNT6 G2)
constexpr size_t MAX_BATCH = 64;       // partial sums per exchange

struct DistCtx {
    bool ready = false;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;        // RCCL transport
    gh_allgather_fn fn = nullptr;     // custom transport
    void* fn_ctx = nullptr;
    uint64_t *d_send = nullptr, *d_recv = nullptr;
    hipStream_t stream = nullptr;     // the exchange's own stream: it never queues behind the bucket sort of the next MSM
};

Rccl R;
DistCtx D;
std::mutex d_mu;

// Which librccl: a process has ONE RCCL worth talking to.  If one is already mapped (a host that imported torch has torch's
// bundled copy), a second copy with its own state would be asking for trouble: use the mapped one and say so.  Otherwise
// the ROCm installation this library was built against, by path; the bare soname last.
int load_rccl() {
    if (R.lib) return GH_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    R.was_mapped = h != nullptr;
    R.note.clear();
    if (h) {
        // ... unless its ABI generation is not the one this library was compiled against (rccl.h NCCL_MAJOR): ncclUniqueId,
        // ncclComm_t and the call signatures are only promised within a major version.  Then the ROCm installation's copy is
        // loaded by path next to it, and gh_dist_transport says so.
        ncclResult_t (*gv)(int*) = nullptr;
        *reinterpret_cast<void**>(&gv) = dlsym(h, "ncclGetVersion");
        int v = 0;
        if (gv && gv(&v) == ncclSuccess && v / 10000 != NCCL_MAJOR) {
            char b[160];
            snprintf(b, sizeof b, "the RCCL already mapped by the host process is %d.%d.%d, built against %d.x: using the ROCm copy", v / 10000,
                     (v / 100) % 100, v % 100, NCCL_MAJOR);
            R.note = b;
            dlclose(h);
            h = nullptr;
            R.was_mapped = false;
        }
    }
    if (!h) {
        const char* env = getenv("GH_RCCL_PATH");
        const char* rocm = getenv("ROCM_PATH");
        std::string by_rocm = std::string(rocm && *rocm ? rocm : "/opt/rocm") + "/lib/librccl.so.1";
        const char* names[] = {env, by_rocm.c_str(), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
        for (const char* n : names) { if (n && *n) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; } }
    }
    if (!h) { const char* e = dlerror(); g_err = std::string("cannot load librccl: ") + (e ? e : "not found"); return GH_E_DIST; }
#define SYM(field, name)                                                   \
    *reinterpret_cast<void**>(&R.field) = dlsym(h, name);                  \
    if (!R.field) { g_err = "librccl lacks " name; dlclose(h); return GH_E_DIST; }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(CommCount, "ncclCommCount")
    SYM(AllGather, "ncclAllGather")
    SYM(GetVersion, "ncclGetVersion")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(R.AllGather), &info) && info.dli_fname) R.path = info.dli_fname;
    int v = 0;
    if (R.GetVersion(&v) == ncclSuccess) R.version = v;
    R.lib = h;
    return GH_OK;
}

#define NCCLCHK(call)                                                                      \
    do {                                                                                   \
        ncclResult_t r_ = (call);                                                          \
        if (r_ != ncclSuccess) { g_err = std::string(#call " failed: ") + R.GetErrorString(r_); return GH_E_DIST; } \
    } while (0)

int deg_of(gh_curve_t c) { return c == GH_MNT4753_G2 ? 2 : (c == GH_MNT6753_G2 ? 3 : 1); }
const MsmOps* ops_for(gh_curve_t c) {
    switch (c) {
        case GH_MNT4753_G1: return msm_ops_mnt4753_g1();
        case GH_MNT4753_G2: return msm_ops_mnt4753_g2();
        case GH_MNT6753_G1: return msm_ops_mnt6753_g1();
        case GH_MNT6753_G2: return msm_ops_mnt6753_g2();
        default: return nullptr;
    }
}

// both locks held
void teardown() {
    if (D.comm) R.CommDestroy(D.comm);
    if (D.d_send) (void)hipFree(D.d_send);
    if (D.d_recv) (void)hipFree(D.d_recv);
    if (D.stream) (void)hipStreamDestroy(D.stream);
    D = DistCtx();
}

// both locks held.  partials / outs: count x words u64
int exchange_fold(gh_curve_t curve, const uint64_t* partials, size_t count, uint64_t* outs, double* exchange_us) {
    const MsmOps* ops = ops_for(curve);
    if (!ops) { g_err = "unknown curve id"; return GH_E_BAD_ARG; }
    const size_t words = (size_t)36 * deg_of(curve), block = words * count;
    std::vector<uint64_t> all(block * (size_t)D.world);
    const auto t0 = std::chrono::steady_clock::now();
    if (D.comm) {
        HIPCHK(hipMemcpyAsync(D.d_send, partials, block * 8, hipMemcpyHostToDevice, D.stream));
        NCCLCHK(R.AllGather(D.d_send, D.d_recv, block, ncclUint64, D.comm, D.stream));
        HIPCHK(hipMemcpyAsync(all.data(), D.d_recv, block * 8 * D.world, hipMemcpyDeviceToHost, D.stream));
        HIPCHK(hipStreamSynchronize(D.stream));
    } else {
        if (D.fn(D.fn_ctx, partials, all.data(), block * 8) != 0) { g_err = "custom all-gather failed"; return GH_E_DIST; }
    }
    if (exchange_us) *exchange_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    // fold in rank order (add_assign, short_weierstrass_projective.rs:574-617), on the host
    for (size_t k = 0; k < count; k++) {
        std::vector<uint64_t> acc(all.begin() + (long)(k * words), all.begin() + (long)((k + 1) * words));
        for (int r = 1; r < D.world; r++) {
            const int rc = ops->proj_add(acc.data(), all.data() + block * (size_t)r + k * words);
            if (rc) return rc;
        }
        memcpy(outs + k * words, acc.data(), words * 8);
    }
    return GH_OK;
}
}  // namespace

namespace gh_rt {
// called by gh_shutdown with the API lock held: a later fold must not find a communicator bound to destroyed streams
void dist_teardown_locked() {
    std::lock_guard<std::mutex> lk(d_mu);
    if (D.ready || D.comm) teardown();
}
}  // namespace gh_rt

extern "C" {

int gh_dist_unique_id(void* out_id128) try {
    std::lock_guard<std::mutex> lk_api(api_mutex());
    std::lock_guard<std::mutex> lk(d_mu);
    if (!out_id128) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    NCCLCHK(R.GetUniqueId(&id));
    static_assert(sizeof(id) == GH_DIST_UNIQUE_ID_BYTES, "unique id size");
    memcpy(out_id128, &id, sizeof id);
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_dist_probe_rccl(void) try {
    std::lock_guard<std::mutex> lk_api(api_mutex());
    std::lock_guard<std::mutex> lk(d_mu);
    return load_rccl();
} catch (...) { return gh_rt::api_exception(); }

int gh_dist_init_rccl(const void* id128, int rank, int world) try {
    int rc = gh_init(nullptr, 0);          // binds the device (no-op if the host already called gh_init); takes the API lock itself
    if (rc) return rc;
    std::lock_guard<std::mutex> lk_api(api_mutex());
    std::lock_guard<std::mutex> lk(d_mu);
    if (D.ready) { g_err = "a communicator already exists"; return GH_E_BAD_ARG; }
    if (!id128 || world < 1 || rank < 0 || rank >= world) { g_err = "bad rank / world / id"; return GH_E_BAD_ARG; }
    if ((rc = load_rccl())) return rc;
    HIPCHK(hipSetDevice(g.device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    NCCLCHK(R.CommInitRank(&D.comm, world, id, rank));
    // from here on every failure path destroys what exists (the communicator must not outlive a failed init)
    auto fail = [&](int code) { teardown(); return code; };
    int cnt = 0;
    {
        ncclResult_t r = R.CommCount(D.comm, &cnt);
        if (r != ncclSuccess) { g_err = std::string("ncclCommCount failed: ") + R.GetErrorString(r); return fail(GH_E_DIST); }
    }
    D.rank = rank; D.world = cnt;
    hipError_t e = hipMalloc((void**)&D.d_send, MAX_WORDS * MAX_BATCH * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&D.d_recv, MAX_WORDS * MAX_BATCH * 8 * (size_t)cnt);
    if (e == hipSuccess) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        e = hipStreamCreateWithPriority(&D.stream, hipStreamNonBlocking, hi);
    }
    if (e != hipSuccess) { g_err = std::string("communicator buffers: ") + hipGetErrorString(e); return fail(e == hipErrorOutOfMemory ? GH_E_NOMEM : GH_E_HIP); }
    D.ready = true;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_dist_init_custom(gh_allgather_fn fn, void* ctx, int rank, int world) try {
    std::lock_guard<std::mutex> lk_api(api_mutex());
    std::lock_guard<std::mutex> lk(d_mu);
    if (D.ready) { g_err = "a communicator already exists"; return GH_E_BAD_ARG; }
    if (!fn || world < 1 || rank < 0 || rank >= world) { g_err = "bad rank / world / callback"; return GH_E_BAD_ARG; }
    D.fn = fn; D.fn_ctx = ctx; D.rank = rank; D.world = world; D.comm = nullptr;
    D.ready = true;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_dist_info(int* rank, int* world) try {
    std::lock_guard<std::mutex> lk_api(api_mutex());
    std::lock_guard<std::mutex> lk(d_mu);
    if (!D.ready) { g_err = "no communicator (gh_dist_init_*)"; return GH_E_DIST; }
    if (rank) *rank = D.rank;
    if (world) *world = D.world;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_dist_transport(int* rccl_ranks, int* rccl_version, char* path, size_t path_cap) try {
    std::lock_guard<std::mutex> lk_api(api_mutex());
    std::lock_guard<std::mutex> lk(d_mu);
    int ranks = 0;
    if (D.ready && D.comm) { int c = 0; if (R.CommCount(D.comm, &c) == ncclSuccess) ranks = c; }
    if (rccl_ranks) *rccl_ranks = ranks;                   // 0: no RCCL communicator (custom transport or none)
    if (rccl_version) *rccl_version = R.lib ? R.version : 0;
    if (path && path_cap)
        snprintf(path, path_cap, "%s%s%s%s%s", R.lib ? R.path.c_str() : "", R.lib && R.was_mapped ? " (already mapped by the host process, same major version as the build headers)" : "",
                 R.note.empty() ? "" : " (", R.note.c_str(), R.note.empty() ? "" : ")");
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_partials_allgather_fold(gh_curve_t curve, const uint64_t* partial_xyz, uint64_t* out_xyz, double* exchange_us) try {
    std::lock_guard<std::mutex> lk_api(api_mutex());
    std::lock_guard<std::mutex> lk(d_mu);
    if (!D.ready) { g_err = "no communicator (gh_dist_init_*)"; return GH_E_DIST; }
    if (!partial_xyz || !out_xyz || (int)curve < 0 || (int)curve > 3) { g_err = "bad argument"; return GH_E_BAD_ARG; }
    return exchange_fold(curve, partial_xyz, 1, out_xyz, exchange_us);
} catch (...) { return gh_rt::api_exception(); }

int gh_partials_allgather_fold_batch(gh_curve_t curve, const uint64_t* partials_xyz, size_t count, uint64_t* outs_xyz, double* exchange_us) try {
    std::lock_guard<std::mutex> lk_api(api_mutex());
    std::lock_guard<std::mutex> lk(d_mu);
    if (!D.ready) { g_err = "no communicator (gh_dist_init_*)"; return GH_E_DIST; }
    if (count == 0) { if (exchange_us) *exchange_us = 0; return GH_OK; }
    if (!partials_xyz || !outs_xyz || (int)curve < 0 || (int)curve > 3) { g_err = "bad argument"; return GH_E_BAD_ARG; }
    double total = 0;
    for (size_t k0 = 0; k0 < count; k0 += MAX_BATCH) {     // one exchange per MAX_BATCH partial sums
        const size_t cnt = count - k0 < MAX_BATCH ? count - k0 : MAX_BATCH;
        const size_t words = (size_t)36 * deg_of(curve);
        double us = 0;
        const int rc = exchange_fold(curve, partials_xyz + k0 * words, cnt, outs_xyz + k0 * words, &us);
        if (rc) return rc;
        total += us;
    }
    if (exchange_us) *exchange_us = total;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_dist_shutdown(void) try {
    std::lock_guard<std::mutex> lk_api(api_mutex());
    std::lock_guard<std::mutex> lk(d_mu);
    if (D.ready || D.comm) teardown();
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

}  // extern "C"

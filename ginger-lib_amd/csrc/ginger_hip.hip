// ginger_hip.hip -- the C ABI declared in include/ginger_hip.h plus the process-wide runtime
// (device context, workspace pool, prefix scan).  Per-curve MSM code lives in msm_<curve>.hip,
// the transforms in ntt.hip.  Build: __graft_entry__.py build() (hipcc --offload-arch=gfx950).
#include <atomic>
#include <condition_variable>
#include <exception>
#include <new>
#include <random>
#include <thread>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "runtime.h"
#include "scan_kernels.h"
#include "asm_kernels.h"

namespace gh_rt {

Ctx g;
std::string g_err;
static std::mutex g_mu;
std::mutex& api_mutex() { return g_mu; }
static char g_devname[256] = "";
static int g_device_req = -1;   // gh_init(devices): explicit device index; -1 = $LOCAL_RANK (or 0)

int ensure_init() {
    if (g.ready) return GH_OK;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        g_err = "no HIP device visible (this library has no CPU fallback)";
        return GH_E_NO_DEVICE;
    }
    int dev = 0;
    const char* lr = getenv("LOCAL_RANK");
    if (g_device_req >= 0) dev = g_device_req;
    else if (lr) dev = atoi(lr) % count;
    g.device = dev;
    HIPCHK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    g.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    snprintf(g_devname, sizeof g_devname, "%s, %d CUs, %s", prop.name, prop.multiProcessorCount, prop.gcnArchName);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_err = std::string("device is not gfx950: ") + prop.gcnArchName;
        return GH_E_NO_DEVICE;
    }
    HIPCHK(hipStreamCreate(&g.stream));
    {
        int least = 0, greatest = 0;   // numerically: least priority >= greatest priority
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        // GH_PRIO_ACC / GH_PRIO_RED = low | normal | high: measurement knobs; defaults: accumulation lowest (the
        // filler of the pipeline), reduction highest (short latency chains)
        auto prio = [&](const char* env, int dflt) {
            const char* v = getenv(env);
            if (!v) return dflt;
            if (!strcmp(v, "low")) return least;
            if (!strcmp(v, "high")) return greatest;
            return (least + greatest) / 2;
        };
        HIPCHK(hipStreamCreateWithPriority(&g.stream_acc, hipStreamDefault, prio("GH_PRIO_ACC", least)));
        HIPCHK(hipStreamCreateWithPriority(&g.stream_red, hipStreamDefault, prio("GH_PRIO_RED", greatest)));
        HIPCHK(hipStreamCreateWithPriority(&g.stream_acc2, hipStreamDefault, prio("GH_PRIO_ACC", least)));
    }
    for (auto& ev : g.tev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (auto& ev : g.ev) HIPCHK(hipEventCreate(&ev));
    for (auto& sl : g.pev) for (auto& ev : sl) HIPCHK(hipEventCreate(&ev));
    g.ready = true;
    return GH_OK;
}

int api_exception() noexcept {
    int rc = GH_E_HIP;
    const char* what = "unknown C++ exception";
    char buf[384];
    try {
        throw;
    } catch (const std::bad_alloc&) {
        rc = GH_E_NOMEM;
        what = "out of host memory (std::bad_alloc)";
    } catch (const std::exception& e) {
        snprintf(buf, sizeof buf, "C++ exception inside the library: %s", e.what());
        what = buf;
    } catch (...) {
    }
    try {
        std::lock_guard<std::mutex> lk(g_mu);     // the entry point's own guard is gone: the stack has been unwound
        g_err = what;
    } catch (...) {
    }
    return rc;
}

int scratch_guard(const void* kernel, size_t threads) {
    // frame size per kernel: one hipFuncGetAttributes per kernel and process
    static std::map<const void*, size_t> frames;
    size_t& reserved = g.scratch_reserved;           // what the runtime already holds for this context's queues (high-water mark)
    auto it = frames.find(kernel);
    if (it == frames.end()) {
        hipFuncAttributes a;
        memset(&a, 0, sizeof a);
        size_t f = 0;
        if (hipFuncGetAttributes(&a, kernel) == hipSuccess) f = a.localSizeBytes;
        else (void)hipGetLastError();
        it = frames.emplace(kernel, f).first;
    }
    const size_t frame = it->second;
    if (frame < 1024) return GH_OK;                  // a few MB at most: never the problem
    // the runtime sizes the reservation for the waves that can be resident, not for the grid
    const size_t resident = (size_t)g.num_cus * 32 * 64;
    const size_t lanes = threads < resident ? (threads + 63) / 64 * 64 : resident;
    const size_t need = frame * lanes;
    if (need <= reserved) return GH_OK;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return GH_OK; }
    const size_t margin = (size_t)256 << 20;
    if (free_b < need - reserved + margin) {
        char b[256];
        snprintf(b, sizeof b, "not enough device memory for the kernel's stack frames: %zu B per lane x %zu lanes = %zu MB of scratch, %zu MB free",
                 frame, lanes, need >> 20, free_b >> 20);
        g_err = b;
        return GH_E_NOMEM;
    }
    reserved = need;
    return GH_OK;
}

int pool_get(const char* name, size_t bytes, void** out) {
    DevBuf& b = g.pool[name];
    if (b.cap < bytes) {
        if (b.p) {
            // A pipelined batch re-uses a slot's buffers two jobs later, while the job before may still be reducing out of them
            // (msm_batch issues sort(k+1) before finish(k-1)): a LARGER job in that position replaces buffers that are in use.
            // Nothing may be freed under a running kernel -- wait for the device first (explicitly: not left to hipFree's implicit
            // synchronisation).  Growth is rare (the pool keeps 1/8 of slack); equal or shrinking jobs never come here.
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipFree(b.p));
            b.p = nullptr;
            b.cap = 0;
        }
        size_t cap = bytes + bytes / 8 + 256;
        HIPCHK(hipMalloc(&b.p, cap));
        b.cap = cap;
    }
    *out = b.p;
    return GH_OK;
}

size_t pool_cap(const char* name) {
    auto it = g.pool.find(name);
    return it == g.pool.end() ? 0 : it->second.cap;
}
void pool_release(const char* prefix) {
    const size_t len = strlen(prefix);
    for (auto it = g.pool.begin(); it != g.pool.end();) {
        if (it->first.compare(0, len, prefix) == 0) {
            if (it->second.p) (void)hipFree(it->second.p);
            it = g.pool.erase(it);
        } else {
            ++it;
        }
    }
}

// generic exclusive scan of n u32 on the library stream
int device_scan(const uint32_t* in, uint32_t* out, size_t n, const char* tmpname, hipStream_t stream) {
    using namespace gh;
    if (!stream) stream = g.stream;
    size_t per_block = (size_t)SCAN_BLOCK * SCAN_ITEMS;
    size_t nblocks = (n + per_block - 1) / per_block;
    uint32_t* sums;
    int rc = pool_get(tmpname, (nblocks + 1) * 4, (void**)&sums);
    if (rc) return rc;
    GH_LAUNCH(scan_partials_kernel, dim3((unsigned)nblocks), dim3(SCAN_BLOCK), 0, stream, in, sums, n);
    GH_LAUNCH(scan_block_sums_kernel, dim3(1), dim3(1024), 0, stream, sums, nblocks);
    GH_LAUNCH(scan_final_kernel, dim3((unsigned)nblocks), dim3(SCAN_BLOCK), 0, stream, in, sums, out, n);
    HIPCHK(hipGetLastError());
    return GH_OK;
}

// Window size.  Measured on MI355X (profiles/r01_window_sweep.txt): besides the usual trade of
// accumulate work (n * ceil(754/c) additions) against bucket-reduction work (2^(c-1) buckets per
// window), what matters is how full the TOP window is -- c = 13 (58 * 13 = 754), 18 (42 * 18 = 756),
// 19 and 21 leave no sparsely populated top window whose few buckets become over-long.
int auto_window(size_t n, int deg) {
    if (g.window_override > 0) return g.window_override;
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    if (deg > 1) {   // G2: the host fold and the reduction weigh more per window -> fewer, larger windows
        int c = lg - 4;
        return c < 4 ? 4 : (c > 20 ? 20 : c);
    }
    if (lg >= 23) return 19;
    if (lg >= 21) return 18;
    if (lg >= 19) return 16;
    if (lg >= 15) return 13;
    int c = lg - 3;
    return c < 4 ? 4 : c;
}

static const MsmOps* ops_of(gh_curve_t curve) {
    switch (curve) {
        case GH_MNT4753_G1: return msm_ops_mnt4753_g1();
        case GH_MNT4753_G2: return msm_ops_mnt4753_g2();
        case GH_MNT6753_G1: return msm_ops_mnt6753_g1();
        case GH_MNT6753_G2: return msm_ops_mnt6753_g2();
        default: g_err = "unknown curve id"; return nullptr;
    }
}

}  // namespace gh_rt

using namespace gh_rt;

// ==========================================================================================
extern "C" {

int gh_init(const int* devices, int n_devices) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g.ready) return GH_OK;
    if (devices && n_devices > 0) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { g_err = "no HIP device visible"; return GH_E_NO_DEVICE; }
        if (devices[0] < 0 || devices[0] >= count) { g_err = "device index out of range"; return GH_E_BAD_ARG; }
        g_device_req = devices[0];
    }
    return ensure_init();
} catch (...) { return gh_rt::api_exception(); }

int gh_shutdown(void) try {
    std::lock_guard<std::mutex> lk(g_mu);
    dist_teardown_locked();              // a communicator must not outlive the streams and the device binding it was made on
    if (!g.ready) return GH_OK;
    hipStreamSynchronize(g.stream);
    hipStreamSynchronize(g.stream_acc);
    hipStreamSynchronize(g.stream_acc2);
    hipStreamSynchronize(g.stream_red);
    for (auto& kv : g.pool) if (kv.second.p) hipFree(kv.second.p);
    g.pool.clear();
    for (auto& f : g.at_shutdown) f();   // function-local device / pinned allocations (msm_impl.h)
    g.at_shutdown.clear();
    for (int f = 0; f < 2; f++) {
        for (auto& kv : g.domains[f]) {
            Domain& d = kv.second;
            if (d.tw) hipFree(d.tw);
            if (d.coset) hipFree(d.coset);
            if (d.coset_inv) hipFree(d.coset_inv);
            if (d.scratch) hipFree(d.scratch);
            if (d.scratch2) hipFree(d.scratch2);
            if (d.d_size_inv) hipFree(d.d_size_inv);
        }
        g.domains[f].clear();
    }
    for (auto& ev : g.ev) hipEventDestroy(ev);
    for (auto& sl : g.pev) for (auto& ev : sl) hipEventDestroy(ev);
    for (auto& ev : g.tev) hipEventDestroy(ev);
    hipStreamDestroy(g.stream_acc);
    hipStreamDestroy(g.stream_acc2);
    hipStreamDestroy(g.stream_red);
    hipStreamDestroy(g.stream);
    g.scratch_reserved = 0;
    g.ready = false;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

const char* gh_last_error(void) try {
    static thread_local std::string tl;
    std::lock_guard<std::mutex> lk(g_mu);
    tl = g_err;
    return tl.c_str();
} catch (...) { (void)gh_rt::api_exception(); return "C++ exception inside the library"; }
const char* gh_device_name(void) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (ensure_init()) return "";
    return g_devname;
} catch (...) { (void)gh_rt::api_exception(); return "C++ exception inside the library"; }

int gh_msm(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases, const uint64_t* scalars,
           size_t n_scalars, uint64_t* out_xyz) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out_xyz || (n_bases && !bases) || (n_scalars && !scalars)) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(curve);
    if (!ops) return GH_E_BAD_ARG;
    int rc = ensure_init();
    if (rc) return rc;
    return ops->host(bases, infinity, n_bases, scalars, n_scalars, out_xyz);
} catch (...) { return gh_rt::api_exception(); }

int gh_bases_upload(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases, gh_bases_t* out_handle) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out_handle || (n_bases && !bases)) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(curve);
    if (!ops) return GH_E_BAD_ARG;
    int rc = ensure_init();
    if (rc) return rc;
    BasesBase* h = nullptr;
    rc = ops->upload(bases, infinity, n_bases, 0, &h);
    if (rc) return rc;
    *out_handle = reinterpret_cast<gh_bases_t>(h);
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

// GroupAffine::write (short_weierstrass_projective.rs:185-192): x || y || infinity byte, every base-field
// coefficient as 96 little-endian bytes of its CANONICAL integer (Fp768::write = into_repr().write, fp_768.rs:784-789).
int gh_bases_upload_wire(gh_curve_t curve, const uint8_t* bytes, size_t n_points, gh_bases_t* out_handle) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out_handle || (n_points && !bytes)) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(curve);
    if (!ops) return GH_E_BAD_ARG;
    int rc = ensure_init();
    if (rc) return rc;
    const int deg = curve == GH_MNT4753_G2 ? 2 : (curve == GH_MNT6753_G2 ? 3 : 1);
    static const uint64_t p4[12] = GH_P4_P_64, p6[12] = GH_P6_P_64;
    const uint64_t* mod = (curve == GH_MNT4753_G1 || curve == GH_MNT4753_G2) ? p4 : p6;
    const size_t rec = (size_t)192 * deg + 1, words = (size_t)24 * deg;
    std::vector<uint64_t> xy(n_points * words);
    std::vector<uint8_t> inf(n_points);
    for (size_t i = 0; i < n_points; i++) {
        const uint8_t* r = bytes + i * rec;
        memcpy(&xy[i * words], r, 192 * (size_t)deg);
        if (r[rec - 1] > 1) { g_err = "wire format: infinity flag is not 0 / 1 (bool::read fails)"; return GH_E_BAD_ARG; }
        inf[i] = r[rec - 1];
        for (size_t e = 0; e < 2 * (size_t)deg; e++) {   // FromBytes rejects values >= p (fp_768.rs:791-805)
            const uint64_t* v = &xy[i * words + 12 * e];
            bool lt = false;
            for (int k = 11; k >= 0; k--) { if (v[k] != mod[k]) { lt = v[k] < mod[k]; break; } }
            if (!lt) { g_err = "wire format: coordinate is not a canonical field element"; return GH_E_BAD_ARG; }
        }
    }
    BasesBase* h = nullptr;
    rc = ops->upload(xy.data(), inf.data(), n_points, 1, &h);
    if (rc) return rc;
    *out_handle = reinterpret_cast<gh_bases_t>(h);
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_bases_free(gh_bases_t handle) try {
    std::lock_guard<std::mutex> lk(g_mu);
    BasesBase* h = reinterpret_cast<BasesBase*>(handle);
    if (!h || h->magic != 0x6768424au) { g_err = "bad bases handle"; return GH_E_BAD_HANDLE; }
    if (h->d_points) hipFree(h->d_points);
    if (h->d_inf) hipFree(h->d_inf);
    if (h->d_table) hipFree(h->d_table);
    if (h->d_dup_starts) hipFree(h->d_dup_starts);
    if (h->d_dup_members) hipFree(h->d_dup_members);
    if (h->d_dup_chunks) hipFree(h->d_dup_chunks);
    h->magic = 0;
    delete h;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_bases_precompute_rows(gh_bases_t handle, int window_bits, int max_rows) try {
    std::lock_guard<std::mutex> lk(g_mu);
    BasesBase* h = reinterpret_cast<BasesBase*>(handle);
    if (!h || h->magic != 0x6768424au) { g_err = "bad bases handle"; return GH_E_BAD_HANDLE; }
    if (window_bits < 0 || window_bits == 1 || window_bits > 24) { g_err = "window must be 0 (auto) or in [2, 24]"; return GH_E_BAD_ARG; }
    if (max_rows < 0) { g_err = "max_rows must be 0 (no cap) or positive"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(h->curve);
    if (!ops) return GH_E_BAD_ARG;
    int rc = ensure_init();
    if (rc) return rc;
    return ops->precompute(h, window_bits, max_rows);
} catch (...) { return gh_rt::api_exception(); }
int gh_bases_precompute(gh_bases_t handle, int window_bits) { return gh_bases_precompute_rows(handle, window_bits, 0); }
int gh_bases_table_rows(gh_bases_t handle) try {
    BasesBase* h = reinterpret_cast<BasesBase*>(handle);
    return (h && h->magic == 0x6768424au && h->d_table) ? h->pre_W : 0;
} catch (...) { return gh_rt::api_exception(); }

int gh_bases_precomputed_window(gh_bases_t handle) try {
    BasesBase* h = reinterpret_cast<BasesBase*>(handle);
    return (h && h->magic == 0x6768424au && h->d_table) ? h->pre_c : 0;
} catch (...) { return gh_rt::api_exception(); }

size_t gh_bases_len(gh_bases_t handle) try {
    BasesBase* h = reinterpret_cast<BasesBase*>(handle);
    return (h && h->magic == 0x6768424au) ? h->n : 0;
} catch (...) { (void)gh_rt::api_exception(); return 0; }

int gh_msm_resident_dev(gh_bases_t handle, const void* d_scalars, size_t n_scalars, uint64_t* out_xyz) try {
    std::lock_guard<std::mutex> lk(g_mu);
    BasesBase* h = reinterpret_cast<BasesBase*>(handle);
    if (!h || h->magic != 0x6768424au) { g_err = "bad bases handle"; return GH_E_BAD_HANDLE; }
    if (!out_xyz || (n_scalars && !d_scalars)) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(h->curve);
    if (!ops) return GH_E_BAD_ARG;
    int rc = ensure_init();
    if (rc) return rc;
    return ops->run(h, d_scalars, n_scalars, out_xyz);
} catch (...) { return gh_rt::api_exception(); }

int gh_msm_resident_dev_batch(const gh_bases_t* handles, const void* const* d_scalars, const size_t* n_scalars, int count,
                              uint64_t* out_xyz) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (count < 0 || (count > 0 && (!handles || !d_scalars || !n_scalars || !out_xyz))) { g_err = "null argument"; return GH_E_BAD_ARG; }
    if (count == 0) return GH_OK;
    std::vector<BasesBase*> hs((size_t)count);
    for (int i = 0; i < count; i++) {
        BasesBase* h = reinterpret_cast<BasesBase*>(handles[i]);
        if (!h || h->magic != 0x6768424au) { g_err = "bad bases handle"; return GH_E_BAD_HANDLE; }
        if (h->curve != reinterpret_cast<BasesBase*>(handles[0])->curve) { g_err = "a batch must stay on one curve"; return GH_E_BAD_ARG; }
        if (n_scalars[i] && !d_scalars[i]) { g_err = "null argument"; return GH_E_BAD_ARG; }
        hs[(size_t)i] = h;
    }
    const MsmOps* ops = ops_of(hs[0]->curve);
    if (!ops) return GH_E_BAD_ARG;
    int rc = ensure_init();
    if (rc) return rc;
    rc = ops->batch(hs.data(), d_scalars, n_scalars, count, out_xyz);
    if (rc) {   // leave no stage of a failed pipeline in flight
        hipStreamSynchronize(g.stream); hipStreamSynchronize(g.stream_acc); hipStreamSynchronize(g.stream_red);
    }
    return rc;
} catch (...) { return gh_rt::api_exception(); }

int gh_msm_resident(gh_bases_t handle, const uint64_t* scalars, size_t n_scalars, uint64_t* out_xyz) try {
    // ONE critical section from staging the scalars to the result: the staging buffer is a shared pool slot
    std::lock_guard<std::mutex> lk(g_mu);
    BasesBase* h = reinterpret_cast<BasesBase*>(handle);
    if (!h || h->magic != 0x6768424au) { g_err = "bad bases handle"; return GH_E_BAD_HANDLE; }
    if (!out_xyz || (n_scalars && !scalars)) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(h->curve);
    if (!ops) return GH_E_BAD_ARG;
    int rc = ensure_init();
    if (rc) return rc;
    const size_t n = h->n < n_scalars ? h->n : n_scalars;
    void* d_s = nullptr;
    if (n > 0) {
        rc = pool_get("scalars", n * 96, &d_s);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(d_s, scalars, n * 96, hipMemcpyHostToDevice, g.stream));
    }
    return ops->run(h, d_s, n, out_xyz);
} catch (...) { return gh_rt::api_exception(); }

// ------------------------------------------------------------------------------------------
// Content-addressed resident keys.  VariableBaseMSM::multi_scalar_mul (variable_base.rs:85-90) is a pure function of its
// two slices, and the prover calls it with the same proving-key queries proof after proof.  gh_msm_cached keeps that
// contract and still moves only the scalars on a repeat: the bases are identified by a 128-bit hash over EVERY limb and
// infinity flag (never by their address: a buffer reused with other bases is another key), a hit is served from the
// resident copy, a miss is uploaded and remembered.  The cache is bounded (LRU by device bytes); the shift table is
// built from the `table_after`-th sighting on (default 2: what is seen twice is a proving key, what is seen once pays
// exactly what gh_msm pays).
namespace {
// A key's identity: FOUR 64-bit lanes over every limb and infinity flag.  Lanes a, b select the cache entry; lanes c, d are an
// independent pair (other multipliers, other seeds, other chunk order mixing) that a hit must ALSO match before the resident
// copy is trusted (round 4: a 128-bit unkeyed hash alone decided a hit, so a collision returned another key's sum with status
// 0).  All four lanes are keyed with per-process random seeds: colliding inputs cannot be prepared offline, and an accidental
// collision needs 256 bits to agree.  What remains assumed: 2^-256.
struct KeyHash { uint64_t a, b, c, d; };
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline void hash_words(const uint64_t* w, size_t n, uint64_t& h1, uint64_t& h2, uint64_t& h3, uint64_t& h4) {
    for (size_t i = 0; i < n; i++) {
        const uint64_t x = w[i];
        h1 = rotl64((h1 ^ x) * 0x9E3779B97F4A7C15ull, 29) + 0xD6E8FEB86659FD93ull;
        h2 = (rotl64(h2, 31) + x) * 0xC2B2AE3D27D4EB4Full ^ (h2 >> 33);
        h3 = rotl64(h3 + x * 0xFF51AFD7ED558CCDull, 27) * 0x94D049BB133111EBull ^ x;
        h4 = (h4 ^ rotl64(x, 17)) * 0xD1342543DE82EF95ull + (h4 >> 29);
    }
}
struct HashSeeds { uint64_t s[4]; };
const HashSeeds& hash_seeds() {
    static const HashSeeds hs = [] {
        HashSeeds v;
        try {
            std::random_device rd;
            for (auto& x : v.s) x = ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ 0x5851F42D4C957F2Dull;
        } catch (...) {   // no entropy source: address-space layout and the clock
            uint64_t t = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ (uint64_t)(uintptr_t)&v;
            for (auto& x : v.s) { t = t * 6364136223846793005ull + 1442695040888963407ull; x = t; }
        }
        return v;
    }();
    return hs;
}
int g_test_hooks = 0;       // gh_test_hooks(): bit 0 = lanes a, b of every key identity are constant (a forced collision)

// A small persistent pool for the hash (round 3 started up to 64 threads per call).  Workers are created once, on first use,
// inside a try block: if the process cannot have them (thread limit, cgroup pids) the caller hashes its chunks inline.
class HashPool {
public:
    static HashPool& get() { static HashPool* p = new HashPool(); return *p; }   // never destroyed: workers sleep until exit
    // runs fn(c) for c in [0, n) on the workers and the calling thread
    void run(size_t n, const std::function<void(size_t)>& fn) {
        if (n == 0) return;
        std::unique_lock<std::mutex> one(submit_);          // one job at a time
        if (n == 1 || workers_ == 0) { for (size_t c = 0; c < n; c++) fn(c); return; }
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn; n_ = n; next_.store(0); pending_ = workers_; gen_++;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }
private:
    HashPool() {
        unsigned nt = std::thread::hardware_concurrency();
        nt = nt == 0 ? 1 : (nt > 32 ? 32 : nt);      // 200 MB of bases per 2^20-pair call: 16 threads hashed them in 3 ms, the MSM behind it takes 24
        for (unsigned t = 1; t < nt; t++) {
            try {
                std::thread([this] { loop(); }).detach();
                workers_++;
            } catch (...) {
                break;
            }
        }
    }
    void work() {
        for (;;) {
            const size_t c = next_.fetch_add(1);
            if (c >= n_) return;
            (*fn_)(c);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
            }
            work();
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    std::mutex submit_, mu_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_ = 0;
    std::atomic<size_t> next_{0};
    unsigned workers_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
};

KeyHash content_hash(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n) {
    const int deg = curve == GH_MNT4753_G2 ? 2 : (curve == GH_MNT6753_G2 ? 3 : 1);
    const size_t words = n * (size_t)24 * deg;
    const HashSeeds& sd = hash_seeds();
    // chunks hashed in parallel, then the chunk hashes are hashed in order
    const size_t chunk = (size_t)1 << 16;
    const size_t n_chunks = (words + chunk - 1) / chunk;
    std::vector<uint64_t> ch(4 * n_chunks + 4);
    HashPool::get().run(n_chunks, [&](size_t c) {
        uint64_t h1 = sd.s[0] + c, h2 = sd.s[1] ^ c, h3 = sd.s[2] - c, h4 = sd.s[3] ^ (c * 0x9E3779B97F4A7C15ull);
        const size_t lo = c * chunk, len = words - lo < chunk ? words - lo : chunk;
        hash_words(bases + lo, len, h1, h2, h3, h4);
        ch[4 * c] = h1; ch[4 * c + 1] = h2; ch[4 * c + 2] = h3; ch[4 * c + 3] = h4;
    });
    uint64_t h1 = sd.s[1] ^ (uint64_t)curve, h2 = sd.s[0] + n, h3 = sd.s[3] + (uint64_t)curve * 0x100000001B3ull, h4 = sd.s[2] ^ n;
    hash_words(ch.data(), 4 * n_chunks, h1, h2, h3, h4);
    if (infinity) {   // flags as bit words; an all-zero flag array hashes like a missing one
        uint64_t any = 0;
        std::vector<uint64_t> fw((n + 63) / 64 + 1, 0);
        for (size_t i = 0; i < n; i++) if (infinity[i]) { fw[i >> 6] |= 1ull << (i & 63); any = 1; }
        if (any) hash_words(fw.data(), fw.size(), h1, h2, h3, h4);
    }
    if (g_test_hooks & 1) { h1 = 0x1111111111111111ull; h2 = 0x2222222222222222ull; }
    return KeyHash{h1, h2, h3, h4};
}
struct CachedKey {
    gh_curve_t curve;
    size_t n;
    KeyHash h;
    BasesBase* key;
    uint64_t stamp;       // LRU clock
    uint32_t sightings;
    size_t bytes;         // device bytes held (points + flags + table)
};
struct KeyCache {
    std::vector<CachedKey> e;
    uint64_t clock = 0;
    size_t max_bytes = 0;                     // 0 = not configured: half of what hipMemGetInfo reports free at the first call
    bool budget_set = false;
    int table_after = 2;                      // build the shift table at this sighting (0 = never)
    gh_key_cache_stats_t st{};
    bool registered = false;
    // the scalars of the call in progress: the cache's OWN buffer, not a pool buffer -- the copy into it runs while this thread
    // hashes, uploads or builds a shift table, and the table builder may drop every pool buffer to make room (pool_release)
    void* d_scal = nullptr;
    size_t scal_cap = 0;
};
KeyCache kc;
size_t key_bytes(const BasesBase* h) {
    const int deg = h->curve == GH_MNT4753_G2 ? 2 : (h->curve == GH_MNT6753_G2 ? 3 : 1);
    const size_t pt = (size_t)208 * deg;
    return h->n * pt + (h->d_inf ? h->n : 0) + (h->d_table ? (size_t)h->pre_W * h->n * pt : 0) + (size_t)4 * (h->n_dup_groups + 1 + h->n_dup_members);
}
void free_key(BasesBase* h) {
    if (!h) return;
    if (h->d_points) hipFree(h->d_points);
    if (h->d_inf) hipFree(h->d_inf);
    if (h->d_table) hipFree(h->d_table);
    if (h->d_dup_starts) hipFree(h->d_dup_starts);
    if (h->d_dup_members) hipFree(h->d_dup_members);
    if (h->d_dup_chunks) hipFree(h->d_dup_chunks);
    h->magic = 0;
    delete h;
}
void cache_drop_all() {
    for (auto& k : kc.e) free_key(k.key);
    kc.e.clear();
    kc.st.entries = 0; kc.st.bytes = 0;
}
// evict least recently used entries (never `keep`) until the cache fits its budget
void cache_fit(const BasesBase* keep) {
    for (;;) {
        size_t total = 0;
        for (auto& k : kc.e) total += k.bytes;
        kc.st.bytes = total; kc.st.entries = kc.e.size();
        if (total <= kc.max_bytes) return;
        size_t victim = kc.e.size();
        for (size_t i = 0; i < kc.e.size(); i++)
            if (kc.e[i].key != keep && (victim == kc.e.size() || kc.e[i].stamp < kc.e[victim].stamp)) victim = i;
        if (victim == kc.e.size()) return;           // only `keep` is left: a key larger than the budget stays for this call
        hipStreamSynchronize(g.stream); hipStreamSynchronize(g.stream_acc); hipStreamSynchronize(g.stream_red);
        free_key(kc.e[victim].key);
        kc.e.erase(kc.e.begin() + (long)victim);
        kc.st.evictions++;
    }
}
}  // namespace

uint64_t gh_bases_content_hash(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases, uint64_t* hi) try {
    if (n_bases && !bases) return 0;
    const KeyHash h = content_hash(curve, bases, infinity, n_bases);
    if (hi) *hi = h.b;
    return h.a;
} catch (...) { (void)gh_rt::api_exception(); return 0; }

int gh_bases_key_id(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases, uint64_t* out4) try {
    if (!out4 || (n_bases && !bases)) { std::lock_guard<std::mutex> lk(g_mu); g_err = "null argument"; return GH_E_BAD_ARG; }
    const KeyHash h = content_hash(curve, bases, infinity, n_bases);
    out4[0] = h.a; out4[1] = h.b; out4[2] = h.c; out4[3] = h.d;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_test_hooks(int flags) try {
    std::lock_guard<std::mutex> lk(g_mu);
    g_test_hooks = flags;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_key_cache_config(size_t max_bytes, int table_after) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (table_after < 0) { g_err = "table_after must be >= 0"; return GH_E_BAD_ARG; }
    kc.budget_set = max_bytes != GH_KEY_CACHE_AUTO;     // automatic: half of what is free at the next gh_msm_cached
    kc.max_bytes = kc.budget_set ? max_bytes : 0;
    kc.table_after = table_after;
    if (g.ready && kc.budget_set) cache_fit(nullptr);
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }
int gh_key_cache_clear(void) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g.ready) { hipStreamSynchronize(g.stream); hipStreamSynchronize(g.stream_acc); hipStreamSynchronize(g.stream_red); }
    cache_drop_all();
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }
int gh_key_cache_stats(gh_key_cache_stats_t* out) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out) { g_err = "null argument"; return GH_E_BAD_ARG; }
    *out = kc.st;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_msm_cached(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases, const uint64_t* scalars,
                  size_t n_scalars, uint64_t* out_xyz) try {
    if (!out_xyz || (n_bases && !bases) || (n_scalars && !scalars)) {
        std::lock_guard<std::mutex> lk(g_mu);
        g_err = "null argument";
        return GH_E_BAD_ARG;
    }
    // like msm_inner's zip (variable_base.rs:31), only the first min(n_bases, n_scalars) bases take part: they are the key
    const size_t n = n_bases < n_scalars ? n_bases : n_scalars;
    std::lock_guard<std::mutex> lk(g_mu);
    const MsmOps* ops = ops_of(curve);
    if (!ops) return GH_E_BAD_ARG;
    int rc = ensure_init();
    if (rc) return rc;
    // The scalars travel to the device WHILE the bases are hashed (round 3: 100 MB over PCIe and 200 MB through the hash are
    // 2.5 ms each at 2^20 pairs; one after the other they were a sixth of the call): a helper thread issues the copy on the
    // library stream, this thread hashes, and the MSM is queued behind the copy on the same stream.
    void* d_s = nullptr;
    hipError_t up_err = hipSuccess;
    std::thread uploader;
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{uploader};
    if (n > 0) {
        if (kc.scal_cap < n * 96) {
            if (kc.d_scal) { HIPCHK(hipFree(kc.d_scal)); kc.d_scal = nullptr; kc.scal_cap = 0; }
            HIPCHK(hipMalloc(&kc.d_scal, n * 96 + 256));
            kc.scal_cap = n * 96;
        }
        d_s = kc.d_scal;
        const int dev = g.device;
        hipStream_t st = g.stream;
        try {
            uploader = std::thread([=, &up_err] {
                hipError_t e = hipSetDevice(dev);
                if (e == hipSuccess) e = hipMemcpyAsync(d_s, scalars, n * 96, hipMemcpyHostToDevice, st);
                up_err = e;
            });
        } catch (...) {      // no thread to be had: the copy goes first, the hash after it
            HIPCHK(hipMemcpyAsync(d_s, scalars, n * 96, hipMemcpyHostToDevice, st));
        }
    }
    const KeyHash hh = content_hash(curve, bases, infinity, n);
    if (!kc.registered) {
        kc.registered = true;
        g.at_shutdown.push_back([] {
            cache_drop_all();
            if (kc.d_scal) (void)hipFree(kc.d_scal);
            kc.d_scal = nullptr; kc.scal_cap = 0;
            kc.registered = false;
        });
    }
    if (!kc.budget_set) {     // half of what is free now, unless gh_key_cache_config said otherwise
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        kc.max_bytes = free_b / 2;
        kc.budget_set = true;
    }
    CachedKey* hit = nullptr;
    for (auto& k : kc.e) {
        if (k.curve != curve || k.n != n || k.h.a != hh.a || k.h.b != hh.b) continue;
        if (k.h.c != hh.c || k.h.d != hh.d) { kc.st.collisions++; continue; }   // equal selection lanes, other content: not this key
        hit = &k;
        break;
    }
    if (!hit) {
        kc.st.misses++;
        BasesBase* h = nullptr;
        rc = ops->upload(bases, infinity, n, 0, &h);
        if (rc == GH_E_NOMEM && !kc.e.empty()) {      // the cache is only a cache: make room and try once more
            (void)hipGetLastError();
            hipStreamSynchronize(g.stream); hipStreamSynchronize(g.stream_acc); hipStreamSynchronize(g.stream_red);
            cache_drop_all();
            rc = ops->upload(bases, infinity, n, 0, &h);
        }
        if (rc) return rc;
        kc.e.push_back(CachedKey{curve, n, hh, h, 0, 0, key_bytes(h)});
        hit = &kc.e.back();
    } else {
        kc.st.hits++;
    }
    hit->stamp = ++kc.clock;
    hit->sightings++;
    BasesBase* key = hit->key;
    if (kc.table_after > 0 && hit->sightings == (uint32_t)kc.table_after && !key->d_table && n >= 4096) {
        // a quarter of the cache's budget per table, so that the four G1 queries of a proving key stay resident together: the
        // full table where it fits into that (2^20 bases: 7.8 of 16 GB), a partial one otherwise (2^24 bases: 4 rows, 14 GB)
        const size_t row_bytes = key_bytes(key);                      // no table yet: the bases themselves = one row
        size_t max_rows = row_bytes ? (kc.max_bytes / 4) / row_bytes : 0;
        if (max_rows > 4096) max_rows = 4096;
        const int prc = max_rows >= 2 ? ops->precompute(key, 0, (int)max_rows)   // optional: NOMEM / UNSUPPORTED leave the key on the per-window path
                                      : GH_E_NOMEM;
        if (prc == GH_OK) kc.st.tables_built++;
        else (void)hipGetLastError();
        hit->bytes = key_bytes(key);
    }
    cache_fit(key);                                   // may erase other entries: `hit` is not used below
    if (uploader.joinable()) uploader.join();
    HIPCHK(up_err);
    return ops->run(key, d_s, n, out_xyz);
} catch (...) { return gh_rt::api_exception(); }

int gh_msm_set_window(int c) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (c < 0 || c > 24 || c == 1) { g_err = "window must be 0 (auto) or in [2, 24]"; return GH_E_BAD_ARG; }
    g.window_override = c;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }
int gh_msm_set_affine(int on) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (on < 0 || on > 2) { g_err = "affine mode must be 0 (off), 1 (on) or 2 (automatic)"; return GH_E_BAD_ARG; }
    g.affine_mode = on;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }
int gh_msm_set_dedup(int on) try {
    std::lock_guard<std::mutex> lk(g_mu);
    g.dedup_mode = on ? 1 : 0;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }
int gh_msm_get_window(gh_curve_t curve, size_t n) try {
    std::lock_guard<std::mutex> lk(g_mu);
    return auto_window(n, curve == GH_MNT4753_G2 ? 2 : (curve == GH_MNT6753_G2 ? 3 : 1));
} catch (...) { return gh_rt::api_exception(); }
int gh_msm_batch_timing(int index, gh_msm_timing_t* out) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (index < 0 || (size_t)index >= g.batch_tm.size()) { g_err = "no such MSM in the last batch"; return GH_E_BAD_ARG; }
    if (out) *out = g.batch_tm[(size_t)index];
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }
int gh_msm_last_timing(gh_msm_timing_t* out) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (out) *out = g.last_msm;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_domain_supported(gh_field_t field, size_t num_coeffs, uint32_t* log_n) try {
    size_t size = 1;
    uint32_t lg = 0;
    while (size < num_coeffs) { size <<= 1; lg++; }
    if (log_n) *log_n = lg;
    int two_adicity = field == GH_MNT4753_FR ? GH_P6_TWO_ADICITY : GH_P4_TWO_ADICITY;
    return (int)lg < two_adicity ? 1 : 0;
} catch (...) { return gh_rt::api_exception(); }

int gh_fft_dev(gh_field_t field, void* d_data, uint32_t log_n, uint32_t flags) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!d_data) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    return fft_run(field, d_data, log_n, flags);
} catch (...) { return gh_rt::api_exception(); }

int gh_fft(gh_field_t field, const uint64_t* in, size_t n_in, uint64_t* out, uint32_t log_n, uint32_t flags) try {
    if (!out || (n_in && !in)) { std::lock_guard<std::mutex> lk(g_mu); g_err = "null argument"; return GH_E_BAD_ARG; }
    if (log_n >= 31) { std::lock_guard<std::mutex> lk(g_mu); g_err = "domain too large"; return GH_E_UNSUPPORTED; }
    const size_t N = (size_t)1 << log_n;
    // ONE critical section: "fft_io" is a shared pool slot
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    int two_adicity = field == GH_MNT4753_FR ? GH_P6_TWO_ADICITY : GH_P4_TWO_ADICITY;
    if ((int)log_n >= two_adicity) { g_err = "domain exceeds the field's 2-adicity"; return GH_E_UNSUPPORTED; }
    void* d = nullptr;
    rc = pool_get("fft_io", N * 96, &d);
    if (rc) return rc;
    size_t ncopy = n_in < N ? n_in : N;  // Vec::resize: truncate or zero-pad (domain.rs:121)
    if (ncopy) HIPCHK(hipMemcpyAsync(d, in, ncopy * 96, hipMemcpyHostToDevice, g.stream));
    if (ncopy < N) HIPCHK(hipMemsetAsync((char*)d + ncopy * 96, 0, (N - ncopy) * 96, g.stream));
    rc = fft_run(field, d, log_n, flags);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, d, N * 96, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_witness_map_dev(gh_field_t field, void* d_a, void* d_b, void* d_c, uint32_t log_n, const uint64_t* d1,
                       const uint64_t* d2, const uint64_t* d3, void* d_h) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!d_a || !d_b || !d_c || !d_h || !d1 || !d2 || !d3) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    return witness_map(field, d_a, d_b, d_c, log_n, d1, d2, d3, d_h);
} catch (...) { return gh_rt::api_exception(); }

int gh_witness_map(gh_field_t field, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint32_t log_n,
                   const uint64_t* d1, const uint64_t* d2, const uint64_t* d3, uint64_t* h) try {
    if (!a || !b || !c || !h || !d1 || !d2 || !d3) { g_err = "null argument"; return GH_E_BAD_ARG; }
    if (log_n >= 31) { g_err = "domain too large"; return GH_E_UNSUPPORTED; }
    const size_t N = (size_t)1 << log_n, bytes = N * 96;
    void *da = nullptr, *db = nullptr, *dc = nullptr, *dh = nullptr;
    int rc = gh_dev_alloc(&da, bytes);
    if (!rc) rc = gh_dev_alloc(&db, bytes);
    if (!rc) rc = gh_dev_alloc(&dc, bytes);
    if (!rc) rc = gh_dev_alloc(&dh, bytes + 96);
    if (!rc) rc = gh_dev_upload(da, a, bytes);
    if (!rc) rc = gh_dev_upload(db, b, bytes);
    if (!rc) rc = gh_dev_upload(dc, c, bytes);
    if (!rc) rc = gh_witness_map_dev(field, da, db, dc, log_n, d1, d2, d3, dh);
    if (!rc) rc = gh_dev_download(h, dh, bytes + 96);
    gh_dev_free(da); gh_dev_free(db); gh_dev_free(dc); gh_dev_free(dh);
    return rc;
} catch (...) { return gh_rt::api_exception(); }

int gh_sap_witness_map_dev(gh_field_t field, void* d_a, void* d_c, uint32_t log_n, const uint64_t* d1, const uint64_t* d2, void* d_h) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!d_a || !d_c || !d_h || !d1 || !d2) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    return sap_witness_map(field, d_a, d_c, log_n, d1, d2, d_h);
} catch (...) { return gh_rt::api_exception(); }

int gh_sap_witness_map(gh_field_t field, const uint64_t* a, const uint64_t* c, uint32_t log_n, const uint64_t* d1, const uint64_t* d2, uint64_t* h) try {
    if (!a || !c || !h || !d1 || !d2) { std::lock_guard<std::mutex> lk(g_mu); g_err = "null argument"; return GH_E_BAD_ARG; }
    if (log_n >= 31) { std::lock_guard<std::mutex> lk(g_mu); g_err = "domain too large"; return GH_E_UNSUPPORTED; }
    const size_t bytes = ((size_t)1 << log_n) * 96;
    void *da = nullptr, *dc = nullptr, *dh = nullptr;
    int rc = gh_dev_alloc(&da, bytes);
    if (!rc) rc = gh_dev_alloc(&dc, bytes);
    if (!rc) rc = gh_dev_alloc(&dh, bytes + 96);
    if (!rc) rc = gh_dev_upload(da, a, bytes);
    if (!rc) rc = gh_dev_upload(dc, c, bytes);
    if (!rc) rc = gh_sap_witness_map_dev(field, da, dc, log_n, d1, d2, dh);
    if (!rc) rc = gh_dev_download(h, dh, bytes + 96);
    gh_dev_free(da); gh_dev_free(dc); gh_dev_free(dh);
    return rc;
} catch (...) { return gh_rt::api_exception(); }

int gh_batch_inverse_dev(gh_field_t field, void* d_a, size_t n) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (n && !d_a) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    return batch_inverse(field, d_a, n);
} catch (...) { return gh_rt::api_exception(); }

int gh_batch_inverse(gh_field_t field, uint64_t* a, size_t n) try {
    if (n == 0) return GH_OK;
    if (!a) { std::lock_guard<std::mutex> lk(g_mu); g_err = "null argument"; return GH_E_BAD_ARG; }
    void* da = nullptr;
    int rc = gh_dev_alloc(&da, n * 96);
    if (!rc) rc = gh_dev_upload(da, a, n * 96);
    if (!rc) rc = gh_batch_inverse_dev(field, da, n);
    if (!rc) rc = gh_dev_download(a, da, n * 96);
    gh_dev_free(da);
    return rc;
} catch (...) { return gh_rt::api_exception(); }

int gh_lagrange_coefficients_dev(gh_field_t field, uint32_t log_n, const uint64_t* tau12, void* d_out) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!tau12 || !d_out) { g_err = "null argument"; return GH_E_BAD_ARG; }
    if (log_n >= 31) { g_err = "domain too large"; return GH_E_UNSUPPORTED; }
    int rc = ensure_init();
    if (rc) return rc;
    return lagrange_coefficients(field, log_n, tau12, d_out);
} catch (...) { return gh_rt::api_exception(); }

int gh_lagrange_coefficients(gh_field_t field, uint32_t log_n, const uint64_t* tau12, uint64_t* out) try {
    if (!tau12 || !out) { std::lock_guard<std::mutex> lk(g_mu); g_err = "null argument"; return GH_E_BAD_ARG; }
    if (log_n >= 31) { std::lock_guard<std::mutex> lk(g_mu); g_err = "domain too large"; return GH_E_UNSUPPORTED; }
    const size_t bytes = ((size_t)1 << log_n) * 96;
    void* d = nullptr;
    int rc = gh_dev_alloc(&d, bytes);
    if (!rc) rc = gh_lagrange_coefficients_dev(field, log_n, tau12, d);
    if (!rc) rc = gh_dev_download(out, d, bytes);
    gh_dev_free(d);
    return rc;
} catch (...) { return gh_rt::api_exception(); }

int gh_measure_fpmul_peak(double* products_per_s) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!products_per_s) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    return gh_asm::measure_fpmul_peak(products_per_s, g.stream);
} catch (...) { return gh_rt::api_exception(); }

int gh_kernel_resources(const char* which, uint32_t* scratch_bytes_per_lane, uint32_t* registers, uint32_t* lds_bytes) try {
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    return gh_asm::kernel_resources(which, scratch_bytes_per_lane, registers, lds_bytes);
} catch (...) { return gh_rt::api_exception(); }

int gh_fft_last_kernel_ms(float* ms) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (ms) *ms = g.last_fft_ms;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

static int vec_dispatch(gh_field_t field, int op, void* d_a, const void* d_b, const uint64_t* s, size_t n) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!d_a || (op != 2 && !d_b) || (op == 2 && !s)) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    return vec_op(field, op, d_a, d_b, s, n);
}
int gh_vec_mul_dev(gh_field_t field, void* d_a, const void* d_b, size_t n) { return vec_dispatch(field, 0, d_a, d_b, nullptr, n); }
int gh_vec_sub_dev(gh_field_t field, void* d_a, const void* d_b, size_t n) { return vec_dispatch(field, 1, d_a, d_b, nullptr, n); }
int gh_vec_scale_dev(gh_field_t field, void* d_a, const uint64_t* scalar12, size_t n) { return vec_dispatch(field, 2, d_a, nullptr, scalar12, n); }

int gh_vec_mul(gh_field_t field, uint64_t* a, const uint64_t* b, size_t n) try {
    if (n == 0) return GH_OK;
    if (!a || !b) { g_err = "null argument"; return GH_E_BAD_ARG; }
    void *da = nullptr, *db = nullptr;
    int rc;
    if ((rc = gh_dev_alloc(&da, n * 96))) return rc;
    if ((rc = gh_dev_alloc(&db, n * 96))) { gh_dev_free(da); return rc; }
    rc = gh_dev_upload(da, a, n * 96);
    if (!rc) rc = gh_dev_upload(db, b, n * 96);
    if (!rc) rc = gh_vec_mul_dev(field, da, db, n);
    if (!rc) rc = gh_dev_download(a, da, n * 96);
    gh_dev_free(da);
    gh_dev_free(db);
    return rc;
} catch (...) { return gh_rt::api_exception(); }
int gh_vec_scale(gh_field_t field, uint64_t* a, const uint64_t* scalar12, size_t n) try {
    if (n == 0) return GH_OK;
    if (!a || !scalar12) { g_err = "null argument"; return GH_E_BAD_ARG; }
    void* da = nullptr;
    int rc;
    if ((rc = gh_dev_alloc(&da, n * 96))) return rc;
    rc = gh_dev_upload(da, a, n * 96);
    if (!rc) rc = gh_vec_scale_dev(field, da, scalar12, n);
    if (!rc) rc = gh_dev_download(a, da, n * 96);
    gh_dev_free(da);
    return rc;
} catch (...) { return gh_rt::api_exception(); }

int gh_dev_alloc(void** d_ptr, size_t bytes) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!d_ptr) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    HIPCHK(hipMalloc(d_ptr, bytes ? bytes : 1));
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }
int gh_dev_free(void* d_ptr) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!d_ptr) return GH_OK;
    HIPCHK(hipFree(d_ptr));
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }
int gh_dev_upload(void* d_dst, const void* h_src, size_t bytes) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!bytes) return GH_OK;
    if (!d_dst || !h_src) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }
int gh_dev_download(void* h_dst, const void* d_src, size_t bytes) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!bytes) return GH_OK;
    if (!h_dst || !d_src) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }
int gh_dev_trim(void) try {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.ready) return GH_OK;
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipStreamSynchronize(g.stream_acc));
    HIPCHK(hipStreamSynchronize(g.stream_red));
    pool_release("");
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }
int gh_dev_sync(void) try {
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_proj_add(gh_curve_t curve, uint64_t* acc_xyz, const uint64_t* p_xyz) try {
    if (!acc_xyz || !p_xyz) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(curve);
    if (!ops) return GH_E_BAD_ARG;
    return ops->proj_add(acc_xyz, p_xyz);
} catch (...) { return gh_rt::api_exception(); }

int gh_proj_mul(gh_curve_t curve, const uint64_t* p_xyz, const uint64_t* scalar12, uint64_t* out_xyz) try {
    if (!p_xyz || !scalar12 || !out_xyz) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(curve);
    if (!ops) return GH_E_BAD_ARG;
    return ops->proj_mul(p_xyz, scalar12, out_xyz);
} catch (...) { return gh_rt::api_exception(); }

int gh_proj_neg(gh_curve_t curve, uint64_t* xyz) try {
    if (!xyz) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(curve);
    if (!ops) return GH_E_BAD_ARG;
    return ops->proj_neg(xyz);
} catch (...) { return gh_rt::api_exception(); }

int gh_proj_to_affine(gh_curve_t curve, const uint64_t* xyz, uint64_t* out_xy, uint8_t* is_infinity) try {
    if (!xyz || !out_xy || !is_infinity) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(curve);
    if (!ops) return GH_E_BAD_ARG;
    return ops->to_affine(xyz, out_xy, is_infinity);
} catch (...) { return gh_rt::api_exception(); }

}  // extern "C"

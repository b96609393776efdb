// ginger_hip.hip -- the C ABI declared in include/ginger_hip.h plus the process-wide runtime
// (device context, workspace pool, prefix scan).  Per-curve MSM code lives in msm_<curve>.hip,
// the transforms in ntt.hip.  Build: __graft_entry__.py build() (hipcc --offload-arch=gfx950).
#include <thread>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "runtime.h"
#include "scan_kernels.h"

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
    }
    for (auto& ev : g.ev) HIPCHK(hipEventCreate(&ev));
    for (auto& sl : g.pev) for (auto& ev : sl) HIPCHK(hipEventCreate(&ev));
    g.ready = true;
    return GH_OK;
}

int pool_get(const char* name, size_t bytes, void** out) {
    DevBuf& b = g.pool[name];
    if (b.cap < bytes) {
        if (b.p) { HIPCHK(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
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
    hipLaunchKernelGGL(scan_partials_kernel, dim3((unsigned)nblocks), dim3(SCAN_BLOCK), 0, stream, in, sums, n);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, stream, sums, nblocks);
    hipLaunchKernelGGL(scan_final_kernel, dim3((unsigned)nblocks), dim3(SCAN_BLOCK), 0, stream, in, sums, out, n);
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

int gh_init(const int* devices, int n_devices) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g.ready) return GH_OK;
    if (devices && n_devices > 0) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { g_err = "no HIP device visible"; return GH_E_NO_DEVICE; }
        if (devices[0] < 0 || devices[0] >= count) { g_err = "device index out of range"; return GH_E_BAD_ARG; }
        g_device_req = devices[0];
    }
    return ensure_init();
}

int gh_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    dist_teardown_locked();              // a communicator must not outlive the streams and the device binding it was made on
    if (!g.ready) return GH_OK;
    hipStreamSynchronize(g.stream);
    hipStreamSynchronize(g.stream_acc);
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
        }
        g.domains[f].clear();
    }
    for (auto& ev : g.ev) hipEventDestroy(ev);
    for (auto& sl : g.pev) for (auto& ev : sl) hipEventDestroy(ev);
    hipStreamDestroy(g.stream_acc);
    hipStreamDestroy(g.stream_red);
    hipStreamDestroy(g.stream);
    g.ready = false;
    return GH_OK;
}

const char* gh_last_error(void) {
    static thread_local std::string tl;
    std::lock_guard<std::mutex> lk(g_mu);
    tl = g_err;
    return tl.c_str();
}
const char* gh_device_name(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (ensure_init()) return "";
    return g_devname;
}

int gh_msm(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases, const uint64_t* scalars,
           size_t n_scalars, uint64_t* out_xyz) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out_xyz || (n_bases && !bases) || (n_scalars && !scalars)) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(curve);
    if (!ops) return GH_E_BAD_ARG;
    int rc = ensure_init();
    if (rc) return rc;
    return ops->host(bases, infinity, n_bases, scalars, n_scalars, out_xyz);
}

int gh_bases_upload(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases, gh_bases_t* out_handle) {
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
}

// GroupAffine::write (short_weierstrass_projective.rs:185-192): x || y || infinity byte, every base-field
// coefficient as 96 little-endian bytes of its CANONICAL integer (Fp768::write = into_repr().write, fp_768.rs:784-789).
int gh_bases_upload_wire(gh_curve_t curve, const uint8_t* bytes, size_t n_points, gh_bases_t* out_handle) {
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
}

int gh_bases_free(gh_bases_t handle) {
    std::lock_guard<std::mutex> lk(g_mu);
    BasesBase* h = reinterpret_cast<BasesBase*>(handle);
    if (!h || h->magic != 0x6768424au) { g_err = "bad bases handle"; return GH_E_BAD_HANDLE; }
    if (h->d_points) hipFree(h->d_points);
    if (h->d_inf) hipFree(h->d_inf);
    if (h->d_table) hipFree(h->d_table);
    h->magic = 0;
    delete h;
    return GH_OK;
}

int gh_bases_precompute_rows(gh_bases_t handle, int window_bits, int max_rows) {
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
}
int gh_bases_precompute(gh_bases_t handle, int window_bits) { return gh_bases_precompute_rows(handle, window_bits, 0); }
int gh_bases_table_rows(gh_bases_t handle) {
    BasesBase* h = reinterpret_cast<BasesBase*>(handle);
    return (h && h->magic == 0x6768424au && h->d_table) ? h->pre_W : 0;
}

int gh_bases_precomputed_window(gh_bases_t handle) {
    BasesBase* h = reinterpret_cast<BasesBase*>(handle);
    return (h && h->magic == 0x6768424au && h->d_table) ? h->pre_c : 0;
}

size_t gh_bases_len(gh_bases_t handle) {
    BasesBase* h = reinterpret_cast<BasesBase*>(handle);
    return (h && h->magic == 0x6768424au) ? h->n : 0;
}

int gh_msm_resident_dev(gh_bases_t handle, const void* d_scalars, size_t n_scalars, uint64_t* out_xyz) {
    std::lock_guard<std::mutex> lk(g_mu);
    BasesBase* h = reinterpret_cast<BasesBase*>(handle);
    if (!h || h->magic != 0x6768424au) { g_err = "bad bases handle"; return GH_E_BAD_HANDLE; }
    if (!out_xyz || (n_scalars && !d_scalars)) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(h->curve);
    if (!ops) return GH_E_BAD_ARG;
    int rc = ensure_init();
    if (rc) return rc;
    return ops->run(h, d_scalars, n_scalars, out_xyz);
}

int gh_msm_resident_dev_batch(const gh_bases_t* handles, const void* const* d_scalars, const size_t* n_scalars, int count,
                              uint64_t* out_xyz) {
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
}

int gh_msm_resident(gh_bases_t handle, const uint64_t* scalars, size_t n_scalars, uint64_t* out_xyz) {
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
}

// ------------------------------------------------------------------------------------------
// Content-addressed resident keys.  VariableBaseMSM::multi_scalar_mul (variable_base.rs:85-90) is a pure function of its
// two slices, and the prover calls it with the same proving-key queries proof after proof.  gh_msm_cached keeps that
// contract and still moves only the scalars on a repeat: the bases are identified by a 128-bit hash over EVERY limb and
// infinity flag (never by their address: a buffer reused with other bases is another key), a hit is served from the
// resident copy, a miss is uploaded and remembered.  The cache is bounded (LRU by device bytes); the shift table is
// built from the `table_after`-th sighting on (default 2: what is seen twice is a proving key, what is seen once pays
// exactly what gh_msm pays).
namespace {
struct KeyHash { uint64_t a, b; };
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
// two independent multiply-rotate lanes over 64-bit words (not cryptographic: the inputs are not adversarial, the point
// is that equal hashes mean equal bases for any two keys a process meets)
inline void hash_words(const uint64_t* w, size_t n, uint64_t& h1, uint64_t& h2) {
    for (size_t i = 0; i < n; i++) {
        h1 = rotl64((h1 ^ w[i]) * 0x9E3779B97F4A7C15ull, 29) + 0xD6E8FEB86659FD93ull;
        h2 = (rotl64(h2, 31) + w[i]) * 0xC2B2AE3D27D4EB4Full ^ (h2 >> 33);
    }
}
KeyHash content_hash(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n) {
    const int deg = curve == GH_MNT4753_G2 ? 2 : (curve == GH_MNT6753_G2 ? 3 : 1);
    const size_t words = n * (size_t)24 * deg;
    // chunks hashed in parallel, then the chunk hashes are hashed in order
    const size_t chunk = (size_t)1 << 16;
    const size_t n_chunks = (words + chunk - 1) / chunk;
    std::vector<uint64_t> ch(2 * n_chunks + 4);
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 64) nt = 64;
    if (nt > n_chunks) nt = n_chunks ? (unsigned)n_chunks : 1;
    auto work = [&](unsigned t) {
        for (size_t c = t; c < n_chunks; c += nt) {
            uint64_t h1 = 0x243F6A8885A308D3ull + c, h2 = 0x13198A2E03707344ull ^ c;
            const size_t lo = c * chunk, len = words - lo < chunk ? words - lo : chunk;
            hash_words(bases + lo, len, h1, h2);
            ch[2 * c] = h1; ch[2 * c + 1] = h2;
        }
    };
    if (nt <= 1) work(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    uint64_t h1 = 0xA4093822299F31D0ull ^ (uint64_t)curve, h2 = 0x082EFA98EC4E6C89ull + n;
    hash_words(ch.data(), 2 * n_chunks, h1, h2);
    if (infinity) {   // flags as 0 / 1 words, eight per word; an all-zero flag array hashes like a missing one
        uint64_t acc = 0, any = 0;
        std::vector<uint64_t> fw((n + 63) / 64 + 1, 0);
        for (size_t i = 0; i < n; i++) if (infinity[i]) { fw[i >> 6] |= 1ull << (i & 63); any = 1; }
        (void)acc;
        if (any) hash_words(fw.data(), fw.size(), h1, h2);
    }
    return KeyHash{h1, h2};
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
    size_t max_bytes = (size_t)64 << 30;     // of the 288 GB
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
    return h->n * pt + (h->d_inf ? h->n : 0) + (h->d_table ? (size_t)h->pre_W * h->n * pt : 0);
}
void free_key(BasesBase* h) {
    if (!h) return;
    if (h->d_points) hipFree(h->d_points);
    if (h->d_inf) hipFree(h->d_inf);
    if (h->d_table) hipFree(h->d_table);
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

uint64_t gh_bases_content_hash(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases, uint64_t* hi) {
    if (n_bases && !bases) return 0;
    const KeyHash h = content_hash(curve, bases, infinity, n_bases);
    if (hi) *hi = h.b;
    return h.a;
}

int gh_key_cache_config(size_t max_bytes, int table_after) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (table_after < 0) { g_err = "table_after must be >= 0"; return GH_E_BAD_ARG; }
    kc.max_bytes = max_bytes;
    kc.table_after = table_after;
    if (g.ready) cache_fit(nullptr);
    return GH_OK;
}
int gh_key_cache_clear(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g.ready) { hipStreamSynchronize(g.stream); hipStreamSynchronize(g.stream_acc); hipStreamSynchronize(g.stream_red); }
    cache_drop_all();
    return GH_OK;
}
int gh_key_cache_stats(gh_key_cache_stats_t* out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out) { g_err = "null argument"; return GH_E_BAD_ARG; }
    *out = kc.st;
    return GH_OK;
}

int gh_msm_cached(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases, const uint64_t* scalars,
                  size_t n_scalars, uint64_t* out_xyz) {
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
    CachedKey* hit = nullptr;
    for (auto& k : kc.e) if (k.curve == curve && k.n == n && k.h.a == hh.a && k.h.b == hh.b) { hit = &k; break; }
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
}

int gh_msm_set_window(int c) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (c < 0 || c > 24 || c == 1) { g_err = "window must be 0 (auto) or in [2, 24]"; return GH_E_BAD_ARG; }
    g.window_override = c;
    return GH_OK;
}
int gh_msm_set_affine(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (on < 0 || on > 2) { g_err = "affine mode must be 0 (off), 1 (on) or 2 (automatic)"; return GH_E_BAD_ARG; }
    g.affine_mode = on;
    return GH_OK;
}
int gh_msm_get_window(gh_curve_t curve, size_t n) {
    std::lock_guard<std::mutex> lk(g_mu);
    return auto_window(n, curve == GH_MNT4753_G2 ? 2 : (curve == GH_MNT6753_G2 ? 3 : 1));
}
int gh_msm_batch_timing(int index, gh_msm_timing_t* out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (index < 0 || (size_t)index >= g.batch_tm.size()) { g_err = "no such MSM in the last batch"; return GH_E_BAD_ARG; }
    if (out) *out = g.batch_tm[(size_t)index];
    return GH_OK;
}
int gh_msm_last_timing(gh_msm_timing_t* out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (out) *out = g.last_msm;
    return GH_OK;
}

int gh_domain_supported(gh_field_t field, size_t num_coeffs, uint32_t* log_n) {
    size_t size = 1;
    uint32_t lg = 0;
    while (size < num_coeffs) { size <<= 1; lg++; }
    if (log_n) *log_n = lg;
    int two_adicity = field == GH_MNT4753_FR ? GH_P6_TWO_ADICITY : GH_P4_TWO_ADICITY;
    return (int)lg < two_adicity ? 1 : 0;
}

int gh_fft_dev(gh_field_t field, void* d_data, uint32_t log_n, uint32_t flags) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!d_data) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    return fft_run(field, d_data, log_n, flags);
}

int gh_fft(gh_field_t field, const uint64_t* in, size_t n_in, uint64_t* out, uint32_t log_n, uint32_t flags) {
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
}

int gh_witness_map_dev(gh_field_t field, void* d_a, void* d_b, void* d_c, uint32_t log_n, const uint64_t* d1,
                       const uint64_t* d2, const uint64_t* d3, void* d_h) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!d_a || !d_b || !d_c || !d_h || !d1 || !d2 || !d3) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    return witness_map(field, d_a, d_b, d_c, log_n, d1, d2, d3, d_h);
}

int gh_witness_map(gh_field_t field, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint32_t log_n,
                   const uint64_t* d1, const uint64_t* d2, const uint64_t* d3, uint64_t* h) {
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
}

int gh_sap_witness_map_dev(gh_field_t field, void* d_a, void* d_c, uint32_t log_n, const uint64_t* d1, const uint64_t* d2, void* d_h) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!d_a || !d_c || !d_h || !d1 || !d2) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    return sap_witness_map(field, d_a, d_c, log_n, d1, d2, d_h);
}

int gh_sap_witness_map(gh_field_t field, const uint64_t* a, const uint64_t* c, uint32_t log_n, const uint64_t* d1, const uint64_t* d2, uint64_t* h) {
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
}

int gh_batch_inverse_dev(gh_field_t field, void* d_a, size_t n) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (n && !d_a) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    return batch_inverse(field, d_a, n);
}

int gh_batch_inverse(gh_field_t field, uint64_t* a, size_t n) {
    if (n == 0) return GH_OK;
    if (!a) { std::lock_guard<std::mutex> lk(g_mu); g_err = "null argument"; return GH_E_BAD_ARG; }
    void* da = nullptr;
    int rc = gh_dev_alloc(&da, n * 96);
    if (!rc) rc = gh_dev_upload(da, a, n * 96);
    if (!rc) rc = gh_batch_inverse_dev(field, da, n);
    if (!rc) rc = gh_dev_download(a, da, n * 96);
    gh_dev_free(da);
    return rc;
}

int gh_lagrange_coefficients_dev(gh_field_t field, uint32_t log_n, const uint64_t* tau12, void* d_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!tau12 || !d_out) { g_err = "null argument"; return GH_E_BAD_ARG; }
    if (log_n >= 31) { g_err = "domain too large"; return GH_E_UNSUPPORTED; }
    int rc = ensure_init();
    if (rc) return rc;
    return lagrange_coefficients(field, log_n, tau12, d_out);
}

int gh_lagrange_coefficients(gh_field_t field, uint32_t log_n, const uint64_t* tau12, uint64_t* out) {
    if (!tau12 || !out) { std::lock_guard<std::mutex> lk(g_mu); g_err = "null argument"; return GH_E_BAD_ARG; }
    if (log_n >= 31) { std::lock_guard<std::mutex> lk(g_mu); g_err = "domain too large"; return GH_E_UNSUPPORTED; }
    const size_t bytes = ((size_t)1 << log_n) * 96;
    void* d = nullptr;
    int rc = gh_dev_alloc(&d, bytes);
    if (!rc) rc = gh_lagrange_coefficients_dev(field, log_n, tau12, d);
    if (!rc) rc = gh_dev_download(out, d, bytes);
    gh_dev_free(d);
    return rc;
}

int gh_fft_last_kernel_ms(float* ms) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (ms) *ms = g.last_fft_ms;
    return GH_OK;
}

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

int gh_vec_mul(gh_field_t field, uint64_t* a, const uint64_t* b, size_t n) {
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
}
int gh_vec_scale(gh_field_t field, uint64_t* a, const uint64_t* scalar12, size_t n) {
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
}

int gh_dev_alloc(void** d_ptr, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!d_ptr) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    HIPCHK(hipMalloc(d_ptr, bytes ? bytes : 1));
    return GH_OK;
}
int gh_dev_free(void* d_ptr) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!d_ptr) return GH_OK;
    HIPCHK(hipFree(d_ptr));
    return GH_OK;
}
int gh_dev_upload(void* d_dst, const void* h_src, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!bytes) return GH_OK;
    if (!d_dst || !h_src) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return GH_OK;
}
int gh_dev_download(void* h_dst, const void* d_src, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!bytes) return GH_OK;
    if (!h_dst || !d_src) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return GH_OK;
}
int gh_dev_trim(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.ready) return GH_OK;
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipStreamSynchronize(g.stream_acc));
    HIPCHK(hipStreamSynchronize(g.stream_red));
    pool_release("");
    return GH_OK;
}
int gh_dev_sync(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    return GH_OK;
}

int gh_proj_add(gh_curve_t curve, uint64_t* acc_xyz, const uint64_t* p_xyz) {
    if (!acc_xyz || !p_xyz) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(curve);
    if (!ops) return GH_E_BAD_ARG;
    return ops->proj_add(acc_xyz, p_xyz);
}

int gh_proj_mul(gh_curve_t curve, const uint64_t* p_xyz, const uint64_t* scalar12, uint64_t* out_xyz) {
    if (!p_xyz || !scalar12 || !out_xyz) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(curve);
    if (!ops) return GH_E_BAD_ARG;
    return ops->proj_mul(p_xyz, scalar12, out_xyz);
}

int gh_proj_neg(gh_curve_t curve, uint64_t* xyz) {
    if (!xyz) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(curve);
    if (!ops) return GH_E_BAD_ARG;
    return ops->proj_neg(xyz);
}

int gh_proj_to_affine(gh_curve_t curve, const uint64_t* xyz, uint64_t* out_xy, uint8_t* is_infinity) {
    if (!xyz || !out_xy || !is_infinity) { g_err = "null argument"; return GH_E_BAD_ARG; }
    const MsmOps* ops = ops_of(curve);
    if (!ops) return GH_E_BAD_ARG;
    return ops->to_affine(xyz, out_xy, is_infinity);
}

}  // extern "C"

// msm_impl.h -- host-side launch sequence of the MSM (templated on the curve policy); included
// by the four msm_<curve>.hip translation units.
#pragma once
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <type_traits>
#include <vector>
#include "runtime.h"
#include "msm_kernels.h"
#include <algorithm>
#include "asm_kernels.h"
#include "host_math.h"

#ifndef GH_F3S_TRIPLE
#define GH_F3S_TRIPLE 2   // Fq3 tower product in the projective accumulation / reduction kernels (F3S in msm_kernels.h):
                          //   0 = three plain products in a rolled loop (three reductions, 4056 mads), 2 waves/SIMD
                          //   1 = triple product with ONE reduction (fp_mul3, 2704 mads) inlined at every site: hipcc did not finish
                          //       that kernel in 25 minutes (eleven sites) -- kept for the record
                          //   2 = the same triple product as ONE out-of-line device function (f3s_mul_outlined): compiles in seconds;
                          //       2^19 pairs: projective accumulation 222 -> 167 ms, bucket reduction 17.4 -> 13.6 ms
#endif
#ifndef GH_AFF_F3S_TRIPLE
#define GH_AFF_F3S_TRIPLE 2   // the same choice for the Fq3 affine rounds: 0 = rolled (2^19 pairs: rounds + finish 133 ms), 1 = inlined
                              // triple product (not compiled after 70 CPU-minutes), 2 = out-of-line triple product at one wave per
                              // SIMD (98 ms; at two waves per SIMD, GH_F3S_CALL_WAVES=2, the caller spills 1.4 KB: 140 ms)
#endif
#ifndef GH_AFF_F2S_DUAL
#define GH_AFF_F2S_DUAL 1   // Fq2 affine rounds: 1 = dual product, one wave per SIMD (2^20 pairs: 68 ms); 0 = two plain products, two waves
                            // per SIMD (784 B of spills: 91 ms)
#endif
#ifndef GH_F2S_DUAL
#define GH_F2S_DUAL 1   // Fq2 accumulation: 1 = dual product at 1 wave/SIMD (119 ms at 2^20 pairs); 0 = two plain products per
                        // lane at 2 waves/SIMD, measured 166 ms (1.8 KB of spills: both shuffled operand sets stay live)
#endif

namespace gh_rt {
using namespace gh;

template <class C> struct CurveId;
template <> struct CurveId<Mnt4G1> { static constexpr gh_curve_t id = GH_MNT4753_G1; };
template <> struct CurveId<Mnt4G2> { static constexpr gh_curve_t id = GH_MNT4753_G2; };
template <> struct CurveId<Mnt6G1> { static constexpr gh_curve_t id = GH_MNT6753_G1; };
template <> struct CurveId<Mnt6G2> { static constexpr gh_curve_t id = GH_MNT6753_G2; };

template <class C> void proj_to_abi_host(uint64_t* out, const Proj<C>& p) {
    typedef typename C::F F;
    uint32_t* w = reinterpret_cast<uint32_t*>(out);
    F::to_abi(w, p.x);
    F::to_abi(w + 24 * F::DEG, p.y);
    F::to_abi(w + 48 * F::DEG, p.z);
}
template <class C> Proj<C> proj_from_abi_host(const uint64_t* in) {
    typedef typename C::F F;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(in);
    Proj<C> p;
    p.x = F::from_abi(w);
    p.y = F::from_abi(w + 24 * F::DEG);
    p.z = F::from_abi(w + 48 * F::DEG);
    return p;
}

// generator constants (ABI Montgomery limbs): curves/mnt{4,6}753/{g1,g2}.rs AFFINE_GENERATOR_COEFFS
template <class C> struct GenConst;
template <> struct GenConst<Mnt4G1> { static void get(uint64_t* xy) { static const uint64_t x[12] = GH_MNT4753_G1_GX0_M_64, y[12] = GH_MNT4753_G1_GY0_M_64; memcpy(xy, x, 96); memcpy(xy + 12, y, 96); } };
template <> struct GenConst<Mnt6G1> { static void get(uint64_t* xy) { static const uint64_t x[12] = GH_MNT6753_G1_GX0_M_64, y[12] = GH_MNT6753_G1_GY0_M_64; memcpy(xy, x, 96); memcpy(xy + 12, y, 96); } };
template <> struct GenConst<Mnt4G2> { static void get(uint64_t* xy) {
    static const uint64_t x0[12] = GH_MNT4753_G2_GX0_M_64, x1[12] = GH_MNT4753_G2_GX1_M_64, y0[12] = GH_MNT4753_G2_GY0_M_64, y1[12] = GH_MNT4753_G2_GY1_M_64;
    memcpy(xy, x0, 96); memcpy(xy + 12, x1, 96); memcpy(xy + 24, y0, 96); memcpy(xy + 36, y1, 96); } };
template <> struct GenConst<Mnt6G2> { static void get(uint64_t* xy) {
    static const uint64_t x0[12] = GH_MNT6753_G2_GX0_M_64, x1[12] = GH_MNT6753_G2_GX1_M_64, x2[12] = GH_MNT6753_G2_GX2_M_64;
    static const uint64_t y0[12] = GH_MNT6753_G2_GY0_M_64, y1[12] = GH_MNT6753_G2_GY1_M_64, y2[12] = GH_MNT6753_G2_GY2_M_64;
    memcpy(xy, x0, 96); memcpy(xy + 12, x1, 96); memcpy(xy + 24, x2, 96); memcpy(xy + 36, y0, 96); memcpy(xy + 48, y1, 96); memcpy(xy + 60, y2, 96); } };

// scalar-field modulus of the curve (MNT4: r = p6, MNT6: r = p4) as 24 LE 32-bit words
template <class C> MsmModulus scalar_modulus() {
    static const uint32_t r4[24] = GH_P6_P_32, r6[24] = GH_P4_P_32;
    const bool mnt4 = CurveId<C>::id == GH_MNT4753_G1 || CurveId<C>::id == GH_MNT4753_G2;
    MsmModulus m;
    memcpy(m.w, mnt4 ? r4 : r6, sizeof m.w);
    return m;
}

template <class C> int make_salts(Aff<C>* out) {
    typedef typename C::F F;
    uint64_t xy[72];
    GenConst<C>::get(xy);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(xy);
    out[0].x = F::from_abi(w);
    out[0].y = F::from_abi(w + 24 * F::DEG);
    Proj<C> g2 = proj_dbl<C>(Proj<C>{out[0].x, out[0].y, F::one()});
    typename F::T zi = host_inv<F>(g2.z);
    out[1].x = F::mul(g2.x, zi);
    out[1].y = F::mul(g2.y, zi);
    return GH_OK;
}

// salt points S0 = G, S1 = 2G (internal affine form) for the accumulate kernels' detour, resident once per curve
template <class C> int device_salts(Aff<C>** out) {
    static Aff<C>* d_salts = nullptr;
    if (!d_salts) {
        Aff<C> hs[2];
        if (int src = make_salts<C>(hs)) return src;
        HIPCHK(hipMalloc((void**)&d_salts, sizeof(hs)));
        HIPCHK(hipMemcpy(d_salts, hs, sizeof(hs), hipMemcpyHostToDevice));
        g.at_shutdown.push_back([] { if (d_salts) hipFree(d_salts); d_salts = nullptr; });
    }
    *out = d_salts;
    return GH_OK;
}

template <class C>
int upload_bases(const uint64_t* bases, const uint8_t* infinity, size_t n, int canonical, BasesBase** out) {
    typedef typename C::F F;
    BasesBase* h = new BasesBase();
    h->curve = CurveId<C>::id;
    h->n = n;
    if (n > 0) {
        const size_t in_bytes = n * (size_t)(48 * F::DEG) * 4;
        void* d_in = nullptr;
        HIPCHK(hipMalloc(&h->d_points, n * sizeof(Aff<C>)));
        HIPCHK(hipMalloc(&d_in, in_bytes));
        HIPCHK(hipMemcpyAsync(d_in, bases, in_bytes, hipMemcpyHostToDevice, g.stream));
        GH_LAUNCH((msm_convert_bases_kernel<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g.stream,
                           (const uint32_t*)d_in, (Aff<C>*)h->d_points, n, canonical);
        HIPCHK(hipGetLastError());
        if (infinity) {
            bool any = false;
            for (size_t i = 0; i < n && !any; i++) any = infinity[i] != 0;
            if (any) {
                HIPCHK(hipMalloc((void**)&h->d_inf, n));
                HIPCHK(hipMemcpyAsync(h->d_inf, infinity, n, hipMemcpyHostToDevice, g.stream));
            }
        }
        HIPCHK(hipStreamSynchronize(g.stream));
        HIPCHK(hipFree(d_in));
    }
    *out = h;
    return GH_OK;
}

// Precomputed shift table for a resident key (msm_kernels.h section 0): rows w = 0 .. W-1 of
// 2^(c w) P_i.  c == 0 picks the window from n.  The table costs W x the bases' footprint
// (n = 2^20 G1, c = 21: 36 x 218 MB = 7.8 GB of the 288 GB), built once per key in slabs.
inline int precompute_window(size_t n, int deg) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    if (g.window_override > 0) return g.window_override;
    // Measured on MI355X (profiles/r01_precompute_sweep.txt).  Only window sizes whose top window
    // is well filled are used: 752 mod c = 14 (c = 18), 12 (20), 17 (21), 16 (23).  With 752 mod c = 4
    // (c = 17, 22) the top window's n digits land on 16 counters and the bucket sort's atomics
    // serialise (sort time x 2.5); c = 16 divides 752 and would add a carry-only window.
    int c;
    // G2: with the affine rounds the accumulation costs 6 tower products per addition instead of 11, while the bucket
    // reduction (2^(c-1) buckets, projective) keeps its price: c = 21 at 2^20 pairs left 26 ms of reduction next to 80 ms
    // of accumulation; c = 19 has a quarter of the buckets for 11 % more additions.
    // (round 3, profiles/r03_shard_sweep.txt: MNT6 G2 2^19 c = 19 5.77 M/s vs c = 18 5.63; 2^22 c = 21 7.26 vs c = 19 6.74 -- at 4 M pairs the
    //  accumulation is long enough to carry the 2^20-bucket reduction)
    if (deg > 1) c = lg <= 18 ? 18 : (lg <= 21 ? 19 : 21);
    else if (lg <= 17) c = 18;
    else if (lg == 18) c = 20;
    else if (lg <= 22) c = 21;
    else c = 23;
    return c;
}

// Groups of equal bases (msm_kernels.h "equal bases"): hashed on the device, grouped on the host, verified limb for limb on the
// device.  Optional: any failure leaves the key without groups (every base its own).  Called when the shift table is built --
// a key that gets a table is a key that is used again.  GH_DEDUP=0 switches it off (A/B).
template <class C>
int dedup_bases(BasesBase* h) {
    if (h->d_dup_starts) { (void)hipFree(h->d_dup_starts); h->d_dup_starts = nullptr; }
    if (h->d_dup_members) { (void)hipFree(h->d_dup_members); h->d_dup_members = nullptr; }
    if (h->d_dup_chunks) { (void)hipFree(h->d_dup_chunks); h->d_dup_chunks = nullptr; }
    h->n_dup_groups = h->n_dup_members = h->n_dup_chunks = 0;
    static const bool off = getenv("GH_DEDUP") && atoi(getenv("GH_DEDUP")) == 0;
    const size_t n = h->n;
    if (off || !g.dedup_mode || n < 2 || n >= ((size_t)1 << 31)) return GH_OK;
    hipStream_t st = g.stream;
    uint64_t* d_hash = nullptr;
    int rc;
    if ((rc = pool_get("dedup_hash", n * 16, (void**)&d_hash))) return rc;
    GH_LAUNCH((msm_base_hash_kernel<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const Aff<C>*)h->d_points, (const uint8_t*)h->d_inf, n, d_hash);
    std::vector<uint64_t> hh(2 * n);
    HIPCHK(hipMemcpyAsync(hh.data(), d_hash, n * 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // group by hash: indices sorted by (h1, h2, index); runs of equal hashes with at least two members are groups
    std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; i++) idx[i] = (uint32_t)i;
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
        if (hh[2 * (size_t)a] != hh[2 * (size_t)b]) return hh[2 * (size_t)a] < hh[2 * (size_t)b];
        if (hh[2 * (size_t)a + 1] != hh[2 * (size_t)b + 1]) return hh[2 * (size_t)a + 1] < hh[2 * (size_t)b + 1];
        return a < b;
    });
    std::vector<uint32_t> starts, members;
    for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        const uint64_t a1 = hh[2 * (size_t)idx[i]], a2 = hh[2 * (size_t)idx[i] + 1];
        while (j < n && hh[2 * (size_t)idx[j]] == a1 && hh[2 * (size_t)idx[j] + 1] == a2) j++;
        if (j - i >= 2 && !(a1 == 0 && a2 == 0)) {            // (0, 0): infinity bases -- the digits stage skips them anyway
            starts.push_back((uint32_t)members.size());
            for (size_t k = i; k < j; k++) members.push_back(idx[k]);      // ascending: the canonical base is the smallest index
        }
        i = j;
    }
    if (starts.empty()) return GH_OK;
    starts.push_back((uint32_t)members.size());
    uint32_t *d_st = nullptr, *d_mem = nullptr;
    uint8_t* d_flags = nullptr;
    HIPCHK(hipMalloc((void**)&d_st, starts.size() * 4));
    if (hipMalloc((void**)&d_mem, members.size() * 4) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d_st); return GH_OK; }
    if ((rc = pool_get("dedup_flags", members.size() + 16, (void**)&d_flags))) { (void)hipFree(d_st); (void)hipFree(d_mem); return GH_OK; }
    const uint32_t ng = (uint32_t)starts.size() - 1;
    hipError_t e = hipMemcpyAsync(d_st, starts.data(), starts.size() * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_mem, members.data(), members.size() * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, members.size(), st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((msm_dup_verify_kernel<C>), dim3(ng), dim3(256), 0, st, (const Aff<C>*)h->d_points, (const uint32_t*)d_st, ng, d_mem, d_flags);
        e = hipGetLastError();
    }
    std::vector<uint8_t> flags(members.size());
    if (e == hipSuccess) e = hipMemcpyAsync(flags.data(), d_flags, members.size(), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(members.data(), d_mem, members.size() * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d_st); (void)hipFree(d_mem); return GH_OK; }
    bool collision = false;
    for (uint8_t f : flags) collision |= f != 0;
    if (collision) {      // equal 128-bit hashes over different abscissae: drop those members and rebuild the lists
        std::vector<uint32_t> st2, mem2;
        for (uint32_t gi = 0; gi < ng; gi++) {
            const size_t b0 = mem2.size();
            for (uint32_t j = starts[gi]; j < starts[gi + 1]; j++) if (!flags[j]) mem2.push_back(members[j]);
            if (mem2.size() - b0 >= 2) st2.push_back((uint32_t)b0); else mem2.resize(b0);
        }
        (void)hipFree(d_st); (void)hipFree(d_mem);
        if (st2.empty()) return GH_OK;
        st2.push_back((uint32_t)mem2.size());
        HIPCHK(hipMalloc((void**)&d_st, st2.size() * 4));
        if (hipMalloc((void**)&d_mem, mem2.size() * 4) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d_st); return GH_OK; }
        HIPCHK(hipMemcpy(d_st, st2.data(), st2.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_mem, mem2.data(), mem2.size() * 4, hipMemcpyHostToDevice));
        starts.swap(st2); members.swap(mem2);
    }
    // chunks of at most MSM_DUP_CHUNK members for the summation (msm_merge_scalars_kernel), then the chunk offsets per group
    const uint32_t ngf = (uint32_t)starts.size() - 1;
    std::vector<uint32_t> ch, goff(ngf + 1);
    for (uint32_t gi = 0; gi < ngf; gi++) {
        goff[gi] = (uint32_t)(ch.size() / 3);
        for (uint32_t lo = starts[gi]; lo < starts[gi + 1]; lo += MSM_DUP_CHUNK) {
            const uint32_t hi = starts[gi + 1] - lo > MSM_DUP_CHUNK ? lo + MSM_DUP_CHUNK : starts[gi + 1];
            ch.push_back(lo); ch.push_back(hi); ch.push_back(gi);
        }
    }
    goff[ngf] = (uint32_t)(ch.size() / 3);
    const uint32_t nch = goff[ngf];
    ch.insert(ch.end(), goff.begin(), goff.end());
    uint32_t* d_ch = nullptr;
    if (hipMalloc((void**)&d_ch, ch.size() * 4) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d_st); (void)hipFree(d_mem); return GH_OK; }
    if (hipMemcpy(d_ch, ch.data(), ch.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError(); (void)hipFree(d_st); (void)hipFree(d_mem); (void)hipFree(d_ch);
        return GH_OK;
    }
    h->d_dup_starts = d_st;
    h->d_dup_members = d_mem;
    h->d_dup_chunks = d_ch;
    h->n_dup_groups = ngf;
    h->n_dup_members = (uint32_t)members.size();
    h->n_dup_chunks = nch;
    return GH_OK;
}

template <class C>
int precompute_bases(BasesBase* h, int c_req, int max_rows) {
    typedef typename C::FC::T FT;
    if (h->d_table) { HIPCHK(hipFree(h->d_table)); h->d_table = nullptr; h->pre_c = h->pre_W = 0; h->pre_G = 1; }
    const size_t n = h->n;
    if (n == 0) return GH_OK;
    // Partial table (max_rows > 0, or GH_TABLE_ROWS for every table of the process): at most that many rows, row j = 2^(c G j) P with
    // G = ceil(windows / max_rows) bucket sets -- window w = j G + g reads row j and files into set g; the G set sums are
    // folded with c doublings each (finish()).  For keys whose full table does not fit next to the others (four 2^24-base
    // G1 queries: 4 x 126 GB at c = 21): 8 rows are 28 GB.  A capped table keeps its sets at 2^20 buckets (c = 21) where the
    // full table of a large key would take c = 23: the sets multiply the bucket reduction.
    const int env_rows = getenv("GH_TABLE_ROWS") ? atoi(getenv("GH_TABLE_ROWS")) : 0;
    const int cap = max_rows > 0 ? max_rows : env_rows;
    int c = c_req > 0 ? c_req : precompute_window(n, C::F::DEG);
    if (c_req <= 0 && cap > 0 && cap < 752 / c + 1 && c > 21) c = 21;
    if (c < 2 || c > 24) { g_err = "precompute window must be in [2, 24]"; return GH_E_BAD_ARG; }
    const int windows = 752 / c + 1;
    const int G = cap > 0 && cap < windows ? (windows + cap - 1) / cap : 1;
    const int W = (windows + G - 1) / G;          // rows of the table
    const int c_row = c * G;                      // doublings from one row to the next
    if ((size_t)W * n >= ((size_t)1 << 31)) { g_err = "precomputed table too large for 31-bit entries"; return GH_E_UNSUPPORTED; }
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const size_t slab = n < ((size_t)1 << 20) ? n : ((size_t)1 << 20);
    const size_t need = (size_t)W * n * sizeof(Aff<C>) + 2 * (size_t)(W - 1) * slab * sizeof(FT) + ((size_t)1 << 30);
    if (need > free_b) {   // the scratch caches of earlier calls (bucket lists, affine-round lists) are only caches: drop them
        HIPCHK(hipStreamSynchronize(g.stream));
        HIPCHK(hipStreamSynchronize(g.stream_acc));
        HIPCHK(hipStreamSynchronize(g.stream_red));
        pool_release("");
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
    }
    if (need > free_b) { g_err = "not enough device memory for the precomputed table"; return GH_E_NOMEM; }
    if (const char* t = getenv("GH_TEST_TABLE_NOMEM"); t && atoi(t) != 0) {
        // fault injection (include/ginger_hip.h gh_test_hooks): the path a table build takes when the card is full -- every pooled
        // scratch buffer is dropped, the key stays on the per-window path.  tests/test_gpu_parity.py runs gh_msm_cached through it
        // on every GPU run (the round-3 fault: a pooled scalar buffer freed here under a running copy).
        HIPCHK(hipStreamSynchronize(g.stream));
        HIPCHK(hipStreamSynchronize(g.stream_acc));
        HIPCHK(hipStreamSynchronize(g.stream_red));
        pool_release("");
        g_err = "not enough device memory for the precomputed table (GH_TEST_TABLE_NOMEM)";
        return GH_E_NOMEM;
    }
    Aff<C>* table = nullptr;
    FT *zs = nullptr, *zp = nullptr;
    uint32_t* bad = nullptr;
    int rc;
    HIPCHK(hipMalloc((void**)&table, (size_t)W * n * sizeof(Aff<C>)));
    if ((rc = pool_get("pre_zs", (size_t)(W - 1) * slab * sizeof(FT) + 8, (void**)&zs)) ||
        (rc = pool_get("pre_zp", (size_t)(W - 1) * slab * sizeof(FT) + 8, (void**)&zp)) ||
        (rc = pool_get("pre_bad", 16, (void**)&bad))) { hipFree(table); return rc; }
    hipStream_t st = g.stream;
    hipError_t e = hipMemcpyAsync(table, h->d_points, n * sizeof(Aff<C>), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(bad, 0, 4, st);
    for (size_t i0 = 0; i0 < n && e == hipSuccess; i0 += slab) {
        const size_t cnt = n - i0 < slab ? n - i0 : slab;
        static const bool pre_jac = !(getenv("GH_PRE_JAC") && atoi(getenv("GH_PRE_JAC")) == 0);
        {   // the table builders carry 2-9 KB of stack per lane: no dispatch the card cannot back with scratch (runtime.h scratch_guard)
            const void* kfn = !pre_jac ? (const void*)(msm_precompute_kernel<C>)
                              : (C::F::DEG == 1 ? (const void*)(msm_precompute_jac_kernel<C, typename C::F>)
                                                : (const void*)(msm_precompute_jac_kernel<C, typename C::FC>));
            if (int grc = scratch_guard(kfn, (cnt + 255) / 256 * 256)) { hipFree(table); return grc; }
        }
        if (pre_jac) {
            if constexpr (C::F::DEG == 1)
                hipLaunchKernelGGL((msm_precompute_jac_kernel<C, typename C::F>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st,
                                   table, (const uint8_t*)h->d_inf, n, i0, cnt, slab, c_row, W, zs, zp, bad);
            else
                hipLaunchKernelGGL((msm_precompute_jac_kernel<C, typename C::FC>), dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, st,
                                   table, (const uint8_t*)h->d_inf, n, i0, cnt, slab, c_row, W, zs, zp, bad);
        } else {
            hipLaunchKernelGGL((msm_precompute_kernel<C>), dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, st,
                               table, (const uint8_t*)h->d_inf, n, i0, cnt, slab, c_row, W, zs, zp, bad);
        }
        e = hipGetLastError();
    }
    uint32_t hbad = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { hipFree(table); g_err = std::string("precompute failed: ") + hipGetErrorString(e); return GH_E_HIP; }
    if (hbad) {   // a base of 2-power order: 2^(c w) P hits infinity, which an affine table cannot hold
        hipFree(table);
        g_err = "precompute: a base has 2-power order; the key stays on the per-window path";
        return GH_E_UNSUPPORTED;
    }
    h->d_table = table;
    h->pre_c = c;
    h->pre_W = W;
    h->pre_G = G;
    (void)dedup_bases<C>(h);      // equal bases of the key: their scalars are added up before every MSM (optional: failures leave none)
    return GH_OK;
}

// Horner over windows, high to low (variable_base.rs:73-82).  Per window the device delivers
// (T, PW, PS, PA, PB) with  R_w = PW 2^(u+6) + PS 2^u + PA 2^6 + PB  and T = plain sum of the
// window's buckets; the terms of acc * 2^c + R_w are folded by descending exponent so that the
// powers of two cost no doubling beyond the c per window that the Horner step needs anyway.
// top_unsigned: window W-1 is "region b" of window W-2 (slot offset 2^(c-1)):
//   R_top = R_(W-2) + R_(W-1) + 2^(c-1) T_(W-1),  weight 2^(c (W-2)).
// HC is the curve policy the fold runs on: the fast 64-bit-limb host field for G1, the generic
// rr29 code otherwise.
template <class HC> struct FoldTerm { int ex; const Proj<HC>* pt; };

template <class HC>
Proj<HC> fold_terms(FoldTerm<HC>* t, int nt) {   // sum pt * 2^ex
    for (int a = 1; a < nt; a++) for (int b = a; b > 0 && t[b].ex > t[b - 1].ex; b--) { FoldTerm<HC> x = t[b]; t[b] = t[b - 1]; t[b - 1] = x; }
    Proj<HC> val = *t[0].pt;
    int cur = t[0].ex;
    for (int k = 1; k < nt; k++) {
        for (int d = 0; d < cur - t[k].ex; d++) val = proj_dbl<HC>(val);
        cur = t[k].ex;
        val = proj_add<HC>(val, *t[k].pt);
    }
    for (int d = 0; d < cur; d++) val = proj_dbl<HC>(val);
    return val;
}

template <class HC>
Proj<HC> fold_generic(const std::vector<Proj<HC>>& hw, int W, int c, int u, int sw, int top_unsigned) {
    auto PT = [&](int which, int w, int k) { return &hw[(size_t)(which * W + w) * 3 + k]; };
    auto window_terms = [&](int w, FoldTerm<HC>* t) {
        t[0] = FoldTerm<HC>{u + sw, PT(0, w, 1)};  // PW   (sw = log2 of the items per wave: 6, or 5 / 4 for G2)
        t[1] = FoldTerm<HC>{u, PT(0, w, 2)};       // PS
        t[2] = FoldTerm<HC>{sw, PT(1, w, 0)};      // PA
        t[3] = FoldTerm<HC>{0, PT(2, w, 0)};       // PB
    };
    Proj<HC> acc = proj_zero<HC>();
    int w = W - 1;
    if (top_unsigned) {
        FoldTerm<HC> t[9];
        window_terms(W - 2, t);
        window_terms(W - 1, t + 4);
        t[8] = FoldTerm<HC>{c - 1, PT(0, W - 1, 0)};   // 2^(c-1) * T_(W-1)
        acc = fold_terms<HC>(t, 9);
        w = W - 3;
    }
    for (; w >= 0; w--) {
        FoldTerm<HC> t[5];
        window_terms(w, t);
        t[4] = FoldTerm<HC>{c, &acc};
        Proj<HC> val = fold_terms<HC>(t, 5);
        acc = val;
    }
    if (proj_is_zero<HC>(acc)) acc = proj_zero<HC>();   // canonical (0, 1, 0) like the reference's zero()
    return acc;
}

template <class C>
void fold_windows(const std::vector<Proj<C>>& hw, int W, int c, int u, int sw, int top_unsigned, uint64_t* out_xyz) {
    typedef typename HostCurveOf<C>::type HC;
    if constexpr (HostCurveOf<C>::fast) {
        std::vector<Proj<HC>> h64(hw.size());
        for (size_t i = 0; i < hw.size(); i++) {   // internal -> ABI Montgomery limbs == host representation
            proj_to_abi_host<C>(reinterpret_cast<uint64_t*>(&h64[i]), hw[i]);
        }
        Proj<HC> acc = fold_generic<HC>(h64, W, c, u, sw, top_unsigned);
        memcpy(out_xyz, &acc, sizeof(acc));
    } else {
        Proj<C> acc = fold_generic<C>(hw, W, c, u, sw, top_unsigned);
        proj_to_abi_host<C>(out_xyz, acc);
    }
}

// Merged windows (precomputed shift table): ONE bucket set of nb = 2^(c-1) slots (slot s = digit magnitude s + 1), cut
// into Wp pseudo-windows of Q = 2^q slots for the two-level wave reduction; slot s = w' Q + k, so
//   sum_s s B_s = sum_w' R_w' + Q sum_w' w' T_w'
// with R_w' = PW 2^(u+6) + PS 2^u + PA 2^6 + PB as above and T_w' the plain sum of pseudo-window w'.
// With a PARTIAL table the buckets form `sets` such sets (set g: the windows w = j sets + g, weight 2^(c g) on top of the
// rows' own 2^(c sets j)): every set is folded as above over its Wp / sets pseudo-windows, then Horner over the sets.
template <class HC>
Proj<HC> fold_merged_generic(const std::vector<Proj<HC>>& hw, int Wp_all, int q, int u, int sw, int sets, int c) {
  Proj<HC> total_acc = proj_zero<HC>();
  const int Wp = Wp_all / sets;
  for (int gset = sets - 1; gset >= 0; gset--) {
    const int w0 = gset * Wp;
    auto PT = [&](int which, int w, int k) -> const Proj<HC>& { return hw[(size_t)(which * Wp_all + w0 + w) * 3 + k]; };
    Proj<HC> spw = proj_zero<HC>(), sps = proj_zero<HC>(), spa = proj_zero<HC>(), spb = proj_zero<HC>();
    Proj<HC> run = proj_zero<HC>(), st = proj_zero<HC>();
    for (int w = Wp - 1; w >= 0; w--) {
        spw = proj_add<HC>(spw, PT(0, w, 1));
        sps = proj_add<HC>(sps, PT(0, w, 2));
        spa = proj_add<HC>(spa, PT(1, w, 0));
        spb = proj_add<HC>(spb, PT(2, w, 0));
        if (w >= 1) { run = proj_add<HC>(run, PT(0, w, 0)); st = proj_add<HC>(st, run); }   // sum_w' w' T_w'
    }
    // slot s carries digit magnitude s + 1: sum (s + 1) B_s = sum s B_s + sum_w' T_w'   (run holds T_1 + .. + T_(Wp-1) here)
    spb = proj_add<HC>(spb, proj_add<HC>(run, PT(0, 0, 0)));
    FoldTerm<HC> t[6] = {{u + sw, &spw}, {u, &sps}, {sw, &spa}, {0, &spb}, {q, &st}, {c, &total_acc}};   // (sets above this one) * 2^c + this set
    Proj<HC> acc = fold_terms<HC>(t, gset == sets - 1 ? 5 : 6);
    total_acc = acc;
  }
    if (proj_is_zero<HC>(total_acc)) total_acc = proj_zero<HC>();
    return total_acc;
}
template <class C>
void fold_merged(const std::vector<Proj<C>>& hw, int Wp, int q, int u, int sw, int sets, int c, uint64_t* out_xyz) {
    typedef typename HostCurveOf<C>::type HC;
    if constexpr (HostCurveOf<C>::fast) {
        std::vector<Proj<HC>> h64(hw.size());
        for (size_t i = 0; i < hw.size(); i++) proj_to_abi_host<C>(reinterpret_cast<uint64_t*>(&h64[i]), hw[i]);
        Proj<HC> acc = fold_merged_generic<HC>(h64, Wp, q, u, sw, sets, c);
        memcpy(out_xyz, &acc, sizeof(acc));
    } else {
        Proj<C> acc = fold_merged_generic<C>(hw, Wp, q, u, sw, sets, c);
        proj_to_abi_host<C>(out_xyz, acc);
    }
}

// One MSM as a sequence of stages, so that several MSMs can be pipelined over HIP streams
// (msm_batch below): sort -> [host reads the chunk plan] -> accumulate -> reduce -> [host fold].
// Every stage works on the buffers of one of two slots.
template <class C>
struct MsmJob {
    BasesBase* h = nullptr;
    const void* d_scalars = nullptr;
    uint64_t* out_xyz = nullptr;
    size_t n = 0;
    int slot = 0;
    bool merged = false;
    int c = 0, W = 0, top_unsigned = 0, RW = 0, L1 = 0, L2 = 0;
    int tpw = 64, sw = 6;           // items per wave of the reduction programs (G2 lane groups: 32 / 16)
    uint32_t nb = 0, Q = 0, win_stride = 0, segs_per_window = 0, heavy_thr = 0, heavy_chunk = 0;
    size_t total = 0, slots = 0, max_heavy = 0, max_chunks = 0;
    uint32_t n_heavy = 0, n_chunks = 0;
    int32_t* digits = nullptr;
    uint32_t *counts = nullptr, *starts = nullptr, *cursor = nullptr, *sorted = nullptr, *order = nullptr, *size_hist = nullptr,
             *size_cursor = nullptr, *chunk_start = nullptr, *plan = nullptr;
    Proj<C>*buckets = nullptr, *seg_out = nullptr, *win_out = nullptr, *partials = nullptr;
    bool solo = false;               // a batch of one (set by msm_batch): nothing runs beside this MSM
    bool last = false;               // the last MSM of its batch: its reduction has nothing to hide behind
    int es = 0;                      // event set (g.pev[es]): the job's index in its batch mod 4, so that the sort of job k+1 can be
                                     // issued while job k-1 (same buffer slot) still waits for its window sums
    int sets = 1;                    // bucket sets (window w -> set w % sets, table row w / sets)
    bool lean = false;               // bucket reduction in its lane-level form (launch_reduce)
    Proj<C>* lane_out = nullptr;
    // affine rounds (aff_kernels.h)
    bool tree = false;
    int tree_rounds = 0;
    uint32_t *aff_cnt = nullptr, *aff_st = nullptr, *aff_nout = nullptr;
    bool aff_sticky_pending = false;
    uint32_t* hplan = nullptr;      // pinned
    Proj<C>* hw = nullptr;          // pinned, 9 RW points
    Aff<C>* salts = nullptr;
    std::chrono::steady_clock::time_point t_begin;
    gh_msm_timing_t tm{};

    // staging buffers per job in flight (set = the job's event set, k & 3): job k+1 is prepared while job k-1 -- same buffer slot --
    // still has its window sums on the way, so the two must not share (or re-allocate) a pinned buffer
    static int pinned(int slot, int which, size_t bytes, void** out) {
        static void* p[4][2] = {};
        static size_t cap[4][2] = {};
        static bool registered = false;
        if (!registered) {   // gh_shutdown releases the staging buffers
            registered = true;
            g.at_shutdown.push_back([] {
                for (auto& sl : p) for (auto& q : sl) { if (q) hipHostFree(q); q = nullptr; }
                for (auto& sl : cap) for (auto& q : sl) q = 0;
                registered = false;
            });
        }
        if (cap[slot][which] < bytes) {
            if (p[slot][which]) HIPCHK(hipHostFree(p[slot][which]));
            p[slot][which] = nullptr; cap[slot][which] = 0;
            HIPCHK(hipHostMalloc(&p[slot][which], bytes + 256, hipHostMallocDefault));
            cap[slot][which] = bytes + 256;
        }
        *out = p[slot][which];
        return GH_OK;
    }

    int prepare(BasesBase* h_, const void* d_scalars_, size_t n_scalars, uint64_t* out, int slot_) {
        h = h_; d_scalars = d_scalars_; out_xyz = out; slot = slot_;
        n = h->n < n_scalars ? h->n : n_scalars;
        t_begin = std::chrono::steady_clock::now();
        if (n == 0) return GH_OK;
        // merged: the key carries a precomputed shift table -> all windows share one bucket set
        merged = h->d_table != nullptr && (g.window_override == 0 || g.window_override == h->pre_c);
        c = merged ? h->pre_c : auto_window(n, C::F::DEG);
        // after sign folding the scalar magnitudes are below 2^752 (msm_kernels.h, digits kernel)
        W = 752 / c + 1;
        top_unsigned = (!merged && 752 % c == 0 && W >= 2) ? 1 : 0;
        nb = (1u << (c - 1)) + (merged ? 0u : 1u);   // merged: slot = |digit| - 1 (msm_kernels.h, digits kernel), weight slot + 1
        // bucket sets the reduction sees: W windows of nb slots, or (merged) RW pseudo-windows of Q slots
        const int q = 15;
        Q = merged ? (nb <= (1u << q) + 1 ? nb : (1u << q)) : nb;
        sets = merged ? h->pre_G : W;                          // bucket sets: 1 with a full table, pre_G with a partial one, W without
        RW = merged ? sets * (int)((nb + Q - 1) / Q) : W;      // (merged: every set is cut into pseudo-windows of Q slots)
        total = (size_t)sets * nb;                             // buckets that exist
        slots = (size_t)RW * Q;                                // bucket array incl. padding
        win_stride = nb;
        static const int env_L1 = getenv("GH_REDUCE_L") ? atoi(getenv("GH_REDUCE_L")) : 0;
        // items per lane, level 1 (power of two).  The wave programs are latency chains (2 L1 + 17 steps,
        // then 2 L2 + 17): as long as the launch stays within one wave per SIMD (1024 on MI355X) a shorter
        // L1 only shortens the chain; beyond that the steps of co-resident waves add up again
        // (measured at 2^20 + 1 buckets: L1 = 16 -> 6.2 ms, 8 -> 6.8, 4 -> 8.2, 32 -> 7.9 -- the one bucket beyond the power of two
        //  added a 1025th / 2049th / 4097th wave program, which ran beside or after another one on its SIMD and doubled the
        //  launch; with the merged set at exactly 2^(c-1) slots level 1 takes 4.3 ms (L1 = 16), level 2 1.1 ms).
        tpw = 64;
        if (C::F::DEG == 2) tpw = 32;     // lane pairs  (msm_kernels.h 5b); the one-lane G2 programs (6-14 KB of stack per lane) are no longer built
        if (C::F::DEG == 3) tpw = 16;     // lane triples, 48 lanes busy
        sw = tpw == 64 ? 6 : (tpw == 32 ? 5 : 4);
        auto programs = [&](int l1) { return (size_t)RW * ((Q + (uint32_t)tpw * l1 - 1) / ((uint32_t)tpw * l1)); };
        L1 = MSM_REDUCE_L;
        while (L1 > 4 && programs(L1 / 2) <= 1024) L1 >>= 1;
        // more programs than SIMDs even at L1 = 16 (the per-window path: 48 windows x 32 segments at 2^20 pairs; every path at
        // 2^24): twice the segment length halves the programs -- 768 instead of 1536 at 2^20, so that no SIMD carries two -- and
        // the tree / scan steps per bucket (round 3: reduce 6.7 -> 5.9 ms at 2^20 per-window, 32.3 -> 29.8 ms at 2^24)
        if (L1 == MSM_REDUCE_L && programs(L1) > 1024) L1 = 2 * MSM_REDUCE_L;
        if (env_L1 >= 4 && env_L1 <= 128 && (env_L1 & (env_L1 - 1)) == 0) L1 = env_L1;
        // Lean reduction (G1, inside a batch): level 1 stops after its serial part and hands every LANE's two sums to level 2
        // (msm_kernels.h, mode 2) -- 2 L1 - 1 steps per segment instead of 2 L1 + 17, a third fewer wave instructions for
        // the reduction, which inside a batch cost the accumulation beside it 3.2 of its 23.4 ms per MSM at 2^20 (measured by
        // leaving the reduction out).  The chain is longer (level 2 then folds 64 x as many items per window: 6.5 + 8.3 ms inside
        // a batch at 2^20 against 8.8 + 3.0), so an MSM that runs alone and the last one of a batch keep the segment form, and
        // so do short accumulations the longer chain would not fit behind (2^18 pairs: 12.8 instead of 8.0 ms per MSM).
        static const int env_lean = getenv("GH_REDUCE_LEAN") ? atoi(getenv("GH_REDUCE_LEAN")) : -1;
        lean = C::F::DEG == 1 && tpw == 64 && (env_lean >= 0 ? env_lean != 0 : (!solo && !last && (size_t)W * n >= ((size_t)1 << 25)));
        const uint32_t seg_slots = (uint32_t)tpw * (uint32_t)L1;
        segs_per_window = (Q + seg_slots - 1) / seg_slots;
        L2 = (int)((segs_per_window + tpw - 1) / tpw);         // items per lane group, level 2 (one wave per window)
        if ((size_t)W * n >= ((size_t)1 << 31) || total >= ((size_t)1 << 31)) {
            g_err = "MSM too large for 31-bit list entries";
            return GH_E_UNSUPPORTED;
        }
        // Bucket sums by affine rounds (aff_kernels.h): g.affine_mode 0 = never, 1 = always, 2 = where they are measured
        // faster: on G2 (6 tower products per addition instead of 11: MNT4 G2 2^20 119 -> 80 ms, MNT6 G2 2^19 200 -> 130 ms)
        // once the list is long enough to fill the chip (a round costs at least one inversion's latency, ~0.3 ms).  On G1
        // the rounds tie with the projective kernel alone (22.9 vs 22.6 ms at 2^20: 0.7 x the instructions, but round 0 is
        // bound by its table gathers and every round pays an inversion per lane) and lose inside a pipelined batch
        // (30.4 vs 28.0 ms per MSM), so G1 stays projective unless asked.
        {
            static const int env_aff = getenv("GH_AFFINE") ? atoi(getenv("GH_AFFINE")) : -1;
            const int mode = env_aff >= 0 ? env_aff : g.affine_mode;
            tree = mode == 1 || (mode == 2 && C::F::DEG >= 2 && (size_t)W * n >= ((size_t)1 << 21));
        }
        if (int src = device_salts<C>(&salts)) return src;
        // Heavy threshold.  Buckets are walked longest first, one per thread at ~78 us per addition
        // (2 waves / SIMD), so a bucket of s entries is free as long as s * 78 us stays well inside the
        // kernel's own duration (~ W n / 1.65e9 s); beyond that it would be the tail, and is split.
        // (merged windows: at least twice the mean bucket W n / 2^(c-1), so that chunking stays the exception)
        heavy_thr = merged ? (uint32_t)(((2 * (size_t)W * n) / (size_t)sets) >> (c - 1)) : (uint32_t)((4 * n) >> (c - 1));
        {
            const uint32_t by_duration = (uint32_t)((double)W * (double)n * 3.1e-6);
            if (heavy_thr < by_duration) heavy_thr = by_duration;
        }
        if (heavy_thr < 128) heavy_thr = 128;
        if (heavy_thr > (uint32_t)MSM_MAX_HEAVY_THRESHOLD) heavy_thr = MSM_MAX_HEAVY_THRESHOLD;
        max_heavy = ((size_t)W * n) / (heavy_thr + 1) + 1;          // buckets with > thr entries
        heavy_chunk = heavy_thr;                                    // chunk = a bucket of threshold size
        max_chunks = ((size_t)W * n) / heavy_chunk + max_heavy + 1;
        int rc;
        char nm[48];
#define POOL(name, ptr, bytes)                                      \
    snprintf(nm, sizeof nm, "%s#%d", name, slot);                   \
    if ((rc = pool_get(nm, bytes, (void**)&ptr))) return rc;
        POOL("digits", digits, (size_t)W * n * 4)
        POOL("counts", counts, total * 4)
        POOL("starts", starts, total * 4)
        POOL("cursor", cursor, total * 4)
        POOL("sorted", sorted, (size_t)W * n * 4)
        POOL("order", order, total * 4)
        POOL("size_hist", size_hist, MSM_SIZE_BINS * 4)
        POOL("size_cursor", size_cursor, MSM_SIZE_BINS * 4)
        POOL("chunk_start", chunk_start, (max_heavy + 2) * 4)
        POOL("plan", plan, 64)
        POOL("buckets", buckets, slots * sizeof(Proj<C>))
        POOL("seg_out", seg_out, (size_t)RW * segs_per_window * 3 * sizeof(Proj<C>))
        POOL("win_out", win_out, (size_t)3 * RW * 3 * sizeof(Proj<C>))
        // (also for the last MSM of a batch, which does not use it: a buffer that is first allocated in the middle of a later batch
        //  costs that batch a device-wide wait -- 9 ms at 2^20)
        if (lean || (C::F::DEG == 1 && tpw == 64 && !solo && (size_t)W * n >= ((size_t)1 << 25))) {
            POOL("lane_out", lane_out, (size_t)RW * segs_per_window * 64 * 2 * sizeof(Proj<C>))
        }
#undef POOL
        if ((rc = pinned(es, 0, 512, (void**)&hplan))) return rc;
        if ((rc = pinned(es, 1, (size_t)9 * RW * sizeof(Proj<C>), (void**)&hw))) return rc;
        return GH_OK;
    }

    // stage 1 (stream st): digits + histogram, scan, bucket order by size, heavy plan, scatter; plan -> host
    int launch_sort(hipStream_t st) {
        if (n == 0) return GH_OK;
        int rc;
        // keys a wave combines into one atomic each before falling back to per-lane atomics (wave_agg_inc)
        static const int env_agg = getenv("GH_AGG_ITERS") ? atoi(getenv("GH_AGG_ITERS")) : -1;
        const int agg_iters = env_agg >= 0 ? env_agg : 12;
        HIPCHK(hipEventRecord(g.pev[es][0], st));
        HIPCHK(hipMemsetAsync(size_hist, 0, MSM_SIZE_BINS * 4, st));
        HIPCHK(hipMemsetAsync(plan, 0, 64, st));
        if (h->n_dup_groups) {     // the scalars of equal bases, added up (msm_kernels.h "equal bases"): the MSM sees the distinct bases only
            char nm[48];
            uint32_t* merged_s = nullptr;
            snprintf(nm, sizeof nm, "merged_scalars#%d", slot);
            if ((rc = pool_get(nm, n * 96, (void**)&merged_s))) return rc;
            HIPCHK(hipMemcpyAsync(merged_s, d_scalars, n * 96, hipMemcpyDeviceToDevice, st));
            uint32_t* partial = nullptr;
            snprintf(nm, sizeof nm, "merged_partial#%d", slot);
            if ((rc = pool_get(nm, (size_t)h->n_dup_chunks * 96 + 96, (void**)&partial))) return rc;
            GH_LAUNCH(msm_merge_scalars_kernel, dim3(h->n_dup_chunks), dim3(256), 0, st, (const uint32_t*)d_scalars, merged_s, n,
                      (const uint32_t*)h->d_dup_starts, (const uint32_t*)h->d_dup_members, (const uint32_t*)h->d_dup_chunks, h->n_dup_chunks,
                      partial, scalar_modulus<C>());
            GH_LAUNCH(msm_merge_groups_kernel, dim3((h->n_dup_groups + 63) / 64), dim3(64), 0, st, merged_s, n, (const uint32_t*)h->d_dup_starts,
                      (const uint32_t*)h->d_dup_members, (const uint32_t*)(h->d_dup_chunks + 3 * (size_t)h->n_dup_chunks), h->n_dup_groups,
                      (const uint32_t*)partial, scalar_modulus<C>());
            d_scalars = merged_s;
        }
        // Bucket lists.  Large inputs: two-level counting sort with LDS atomics only (msm_kernels.h 2a); small ones: histogram +
        // scatter with device-scope atomics (fewer launches).  GH_SORT=atomic / part forces one of them where it applies.
        const size_t entries = (size_t)W * n;
        static const char* env_sort = getenv("GH_SORT");
        static const bool sort_ordered = !(getenv("GH_SORT_ORDERED") && atoi(getenv("GH_SORT_ORDERED")) == 0);
        const uint32_t tile = entries > ((size_t)1 << 27) ? 65536u : 16384u;
        uint32_t bin_shift = 8;
        auto bins_at = [&](uint32_t sh) { return (total + ((size_t)1 << sh) - 1) >> sh; };
        while (bins_at(bin_shift) > 1024) bin_shift++;
        // more than 2^23 buckets (2^24 pairs per window at c = 19: 40 x 2^18): up to MSM_PART_MAX_BINS bins of 2^13 buckets rather
        // than the device-scope atomics (sort 65 ms there)
        if (bin_shift > 13 && bins_at(13) <= (size_t)MSM_PART_MAX_BINS) bin_shift = 13;
        bool part_sort = entries >= ((size_t)1 << 22) && n >= tile && bin_shift <= 13;
        if (env_sort && !strcmp(env_sort, "atomic")) part_sort = false;
        if (env_sort && !strcmp(env_sort, "part") && n >= tile && bin_shift <= 13) part_sort = true;
        const bool use_part = part_sort;
        if (use_part) {
            MsmPartArgs a;
            a.digits = digits; a.entries = entries; a.n = n;
            a.win_stride = win_stride; a.row_stride = merged ? (uint32_t)h->n : 0u; a.slot_shift = merged ? 1u : 0u;
            a.sets = (uint32_t)sets;
            a.bin_shift = bin_shift; a.n_bins = (uint32_t)((total + ((size_t)1 << bin_shift) - 1) >> bin_shift);
            a.tile = tile; a.n_blocks = (uint32_t)((entries + tile - 1) / tile);
            const size_t cells = (size_t)a.n_bins * a.n_blocks + 1;
            uint32_t *block_hist = nullptr, *block_off = nullptr;
            uint2* part = nullptr;
            char nm[48];
            snprintf(nm, sizeof nm, "part_hist#%d", slot);
            if ((rc = pool_get(nm, cells * 4, (void**)&block_hist))) return rc;
            snprintf(nm, sizeof nm, "part_off#%d", slot);
            if ((rc = pool_get(nm, cells * 4, (void**)&block_off))) return rc;
            snprintf(nm, sizeof nm, "part_pairs#%d", slot);
            if ((rc = pool_get(nm, entries * 8, (void**)&part))) return rc;
            GH_LAUNCH(msm_digits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               (const uint32_t*)d_scalars, (const uint8_t*)h->d_inf, n, c, W, win_stride, top_unsigned, scalar_modulus<C>(), digits,
                               (uint32_t*)nullptr, agg_iters, merged ? 1u : 0u, (uint32_t)sets);
            HIPCHK(hipMemsetAsync(block_hist + (cells - 1), 0, 4, st));
            GH_LAUNCH(msm_part_hist_kernel, dim3(a.n_blocks), dim3(MSM_PART_THREADS), 0, st, a, block_hist);
            HIPCHK(hipGetLastError());
            snprintf(nm, sizeof nm, "scan_tmp3#%d", slot);
            if ((rc = device_scan(block_hist, block_off, cells, nm, st))) return rc;
            GH_LAUNCH(msm_part_scatter_kernel, dim3(a.n_blocks), dim3(MSM_PART_THREADS), 0, st, a, (const uint32_t*)block_off, part);
            GH_LAUNCH(msm_bin_sort_kernel, dim3(a.n_bins), dim3(MSM_BIN_THREADS), (size_t)4 << bin_shift, st, (const uint2*)part,
                               (const uint32_t*)block_off, a.n_blocks, bin_shift, (uint32_t)total, counts, starts, sorted, sort_ordered ? 1u : 0u);
            HIPCHK(hipGetLastError());
        } else {
            HIPCHK(hipMemsetAsync(counts, 0, total * 4, st));
            GH_LAUNCH(msm_digits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               (const uint32_t*)d_scalars, (const uint8_t*)h->d_inf, n, c, W, win_stride, top_unsigned, scalar_modulus<C>(), digits, counts, agg_iters,
                               merged ? 1u : 0u, (uint32_t)sets);
            HIPCHK(hipGetLastError());
            if ((rc = device_scan(counts, starts, total, "scan_tmp", st))) return rc;
            HIPCHK(hipMemcpyAsync(cursor, starts, total * 4, hipMemcpyDeviceToDevice, st));
        }
        GH_LAUNCH(msm_size_hist_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, counts, total, heavy_thr, size_hist, plan + 4);
        if ((rc = device_scan(size_hist, size_cursor, MSM_SIZE_BINS, "scan_tmp2", st))) return rc;
        GH_LAUNCH(msm_size_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, counts, total, heavy_thr, size_cursor, order);
        GH_LAUNCH(msm_heavy_plan_kernel, dim3(1), dim3(1), 0, st, (const uint32_t*)size_hist, (const uint32_t*)counts,
                           (const uint32_t*)order, (const uint32_t*)starts, (uint32_t)total, heavy_chunk, chunk_start, plan);
        if (!use_part)
            GH_LAUNCH(msm_scatter_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)W), dim3(256), 0, st,
                               (const int32_t*)digits, n, W, win_stride, merged ? (uint32_t)h->n : 0u, cursor, sorted, agg_iters, merged ? 1u : 0u,
                               (uint32_t)sets);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(hplan, plan, 32, hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(g.pev[es][1], st));
        return GH_OK;
    }

    // stage 2 (stream st): waits on the host for the plan of stage 1, then the accumulation launch
    int launch_accumulate(hipStream_t st) {
        if (n == 0) return GH_OK;
        int rc;
        HIPCHK(hipEventSynchronize(g.pev[es][1]));
        n_heavy = hplan[0]; n_chunks = hplan[1];
        tm.accumulate_madds = hplan[2];
        if (n_heavy > max_heavy || n_chunks > max_chunks) { g_err = "internal: heavy-bucket plan out of range"; return GH_E_HIP; }
        const size_t lds_wave = 64 * sizeof(Proj<C>);
        partials = nullptr;
        char nm[48];
        snprintf(nm, sizeof nm, "partials#%d", slot);
        if (n_heavy > 0 && (rc = pool_get(nm, (size_t)n_chunks * sizeof(Proj<C>), (void**)&partials))) return rc;
        // 2 waves / SIMD (256 VGPRs, 184 B scratch) measured 29.3 ms vs 34.6 ms for 1 wave (297 registers) at 2^20
        static const int acc_waves = getenv("GH_ACC_WAVES") ? atoi(getenv("GH_ACC_WAVES")) : 2;
        const void* src_points = merged ? h->d_table : h->d_points;
        if (slots > total)   // padding slots of the last pseudo-window: infinity (Z = 0)
            HIPCHK(hipMemsetAsync((void*)(buckets + total), 0, (slots - total) * sizeof(Proj<C>), st));
        HIPCHK(hipEventRecord(g.pev[es][2], st));
        if (tree) {   // may clear `tree` when its scratch does not fit next to the key: the projective kernel takes over
            if ((rc = launch_tree(st))) return rc;
        }
        if (!tree) {
        {
            // one launch: the chunks of the heavy buckets first, then every other bucket, longest first
            const size_t tasks = (size_t)n_chunks + (total - n_heavy);
            // G2: one coefficient per lane, 2 (Fq2) / 3 (Fq3) lanes per task (msm_kernels.h 4b)
            constexpr bool is_g2 = std::is_same<C, Mnt4G2>::value || std::is_same<C, Mnt6G2>::value;
            if constexpr (is_g2) {
                {
                    typedef typename std::conditional<std::is_same<C, Mnt4G2>::value, F2S<P4, 13, GH_F2S_DUAL != 0>, F3S<P6, 11, GH_F3S_TRIPLE>>::type FS;
                    constexpr int LANES = FS::LANES;
                    const size_t waves = (tasks + (64 / LANES) - 1) / (64 / LANES);
                    GH_LAUNCH((msm_accumulate_split_kernel<C, FS, LANES>), dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, st,
                                       (const Aff<C>*)src_points, (const uint32_t*)sorted, (const uint32_t*)starts,
                                       (const uint32_t*)counts, (const uint32_t*)order, (uint32_t)total, (const Aff<C>*)salts, buckets,
                                       (const uint32_t*)chunk_start, n_heavy, n_chunks, heavy_chunk, partials);
                }
            }
            // G1: XYZZ accumulators (msm_kernels.h 4a: 10 multiplications / 9 reductions per update); GH_ACC_XYZZ=0 selects the
            // homogeneous-projective kernel (madd-1998-cmo, 11 / 11) for A/B measurements
            static const bool acc_xyzz = !(getenv("GH_ACC_XYZZ") && atoi(getenv("GH_ACC_XYZZ")) == 0);
            bool done_xyzz = false;
            if constexpr (C::F::DEG == 1) {
                if (acc_xyzz && acc_waves >= 2 && gh_asm::enabled()) {
                    // the assembly kernel (asmgen/g1_xyzz.py): the same updates on a fixed register plan, 0 B of scratch
                    gh_asm::AccTask* tk = nullptr;
                    snprintf(nm, sizeof nm, "acc_tasks#%d", slot);
                    if ((rc = pool_get(nm, tasks * sizeof(gh_asm::AccTask), (void**)&tk))) return rc;
                    GH_LAUNCH((msm_acc_tasks_kernel<C>), dim3((unsigned)((tasks + 255) / 256)), dim3(256), 0, st,
                                       (const uint32_t*)starts, (const uint32_t*)counts, (const uint32_t*)order, (uint32_t)total, buckets,
                                       (const uint32_t*)chunk_start, n_heavy, n_chunks, heavy_chunk, partials, (AccTaskRec*)tk);
                    if ((rc = gh_asm::acc_g1_launch(std::is_same<typename C::PF, P6>::value ? 6 : 4, src_points, (const uint32_t*)sorted, tk,
                                                    salts, (uint32_t)tasks, st))) return rc;
                    done_xyzz = true;
                } else if (acc_xyzz && acc_waves >= 2) {
                    GH_LAUNCH((msm_accumulate_xyzz_kernel<C>), dim3((unsigned)((tasks + 255) / 256)), dim3(256), 0, st,
                                       (const Aff<C>*)src_points, (const uint32_t*)sorted, (const uint32_t*)starts,
                                       (const uint32_t*)counts, (const uint32_t*)order, (uint32_t)total, (const Aff<C>*)salts, buckets,
                                       (const uint32_t*)chunk_start, n_heavy, n_chunks, heavy_chunk, partials, 0u, 0u);
                    done_xyzz = true;
                }
            }
            if constexpr (!is_g2) {      // (the one-lane G2 instances, 4-8 KB of stack per lane, are no longer built: G2 is always split)
            if (!done_xyzz) {
                if (acc_waves >= 2)
                    GH_LAUNCH((msm_accumulate_kernel<C, 2>), dim3((unsigned)((tasks + 255) / 256)), dim3(256), 0, st,
                                       (const Aff<C>*)src_points, (const uint32_t*)sorted, (const uint32_t*)starts,
                                       (const uint32_t*)counts, (const uint32_t*)order, (uint32_t)total, (const Aff<C>*)salts, buckets,
                                       (const uint32_t*)chunk_start, n_heavy, n_chunks, heavy_chunk, partials);
                else
                    GH_LAUNCH((msm_accumulate_kernel<C, 1>), dim3((unsigned)((tasks + 255) / 256)), dim3(256), 0, st,
                                       (const Aff<C>*)src_points, (const uint32_t*)sorted, (const uint32_t*)starts,
                                       (const uint32_t*)counts, (const uint32_t*)order, (uint32_t)total, (const Aff<C>*)salts, buckets,
                                       (const uint32_t*)chunk_start, n_heavy, n_chunks, heavy_chunk, partials);
            }
            }
        }
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(g.pev[es][3], st));
        if (n_heavy > 0) {   // one wave per heavy bucket adds its chunk sums
            GH_LAUNCH((msm_heavy_combine_kernel<C>), dim3(n_heavy), dim3(64), lds_wave, st, (const Proj<C>*)partials,
                               (const uint32_t*)order, (const uint32_t*)chunk_start, buckets);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(g.pev[es][4], st));
        return GH_OK;
    }

    // Bucket sums by affine rounds (aff_kernels.h) on stream st: plan (per-round bucket sizes and offsets: R small
    // scans), R rounds (descriptor kernel + round kernel), then the projective kernel over what is left per bucket.
    // lane-group field of the rounds: one lane per element (G1), lane pairs with the dual product (Fq2), lane triples with
    // the single-reduction triple product (Fq3: six product sites per addition, where the projective kernel's eleven did
    // not get through hipcc unrolled)
    typedef typename std::conditional<C::F::DEG == 1, F1S<typename C::PF>,
            typename std::conditional<C::F::DEG == 2, F2S<P4, 13, GH_AFF_F2S_DUAL != 0>, F3S<P6, 11, GH_AFF_F3S_TRIPLE>>::type>::type TreeFS;

    int launch_tree(hipStream_t st) {
        typedef TreeFS FS;
        constexpr int LANES = FS::LANES;
        constexpr uint32_t TPW = 64 / LANES;
        int rc;
        const uint32_t n0 = hplan[2], maxc = hplan[4];
        static const int env_R = getenv("GH_AFF_ROUNDS") ? atoi(getenv("GH_AFF_ROUNDS")) : 0;
        static const int env_bmin = getenv("GH_AFF_BMIN") ? atoi(getenv("GH_AFF_BMIN")) : 8;
        static const int env_fin = getenv("GH_AFF_FINISH_MAX") ? atoi(getenv("GH_AFF_FINISH_MAX")) : 64;
        // rounds: down to ~env_left points per bucket on average (the late rounds are short batches -- one inversion per
        // lane and round -- while the projective finish is dense work), and no bucket left with more than env_fin points
        // (round 3, profiles/r03_g2_knobs.txt: on the towers the projective finish costs 11 tower products per point against the rounds' 6,
        //  so fewer points are left to it: Fq3 1.5 (MNT6 G2 2^19: 5.75 -> 5.95 M/s together with the one-chunk scratch budget), Fq2 2.5)
        static const double env_left = getenv("GH_AFF_LEFTOVER") ? atof(getenv("GH_AFF_LEFTOVER")) : (C::F::DEG == 3 ? 1.5 : (C::F::DEG == 2 ? 2.5 : 4.5));
        int R = 1;
        {
            const double mean = (double)n0 / (double)(total > 1 ? total - 1 : 1);
            while (R < AFF_MAX_ROUNDS && (double)(1u << R) * env_left < mean) R++;
            if (env_R > 0) R = env_R;
            while (R < AFF_MAX_ROUNDS && (maxc >> R) > (uint32_t)env_fin) R++;
        }
        tree_rounds = R;
        const size_t stride = (total + 63) & ~(size_t)63;
        char nm[48];
#define POOLT(name, ptr, bytes)                                     \
    snprintf(nm, sizeof nm, "%s#%d", name, slot);                   \
    if ((rc = pool_get(nm, bytes, (void**)&ptr))) return rc;
        POOLT("aff_cnt", aff_cnt, (size_t)R * stride * 4)
        POOLT("aff_st", aff_st, (size_t)R * stride * 4)
        GH_LAUNCH(aff_counts_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           (const uint32_t*)counts, (uint32_t)total, R, stride, aff_cnt);
        snprintf(nm, sizeof nm, "aff_scan#%d", slot);
        for (int r = 1; r <= R; r++)
            if ((rc = device_scan(aff_cnt + (size_t)(r - 1) * stride, aff_st + (size_t)(r - 1) * stride, total, nm, st))) return rc;
        // Chunks of buckets: the scratch lists of the rounds are sized per chunk, so that a 2^24-pair key (or a G2 key with
        // its shift table) does not need 300 GB of them.  ~420 bytes x lanes per list entry: the staged inputs, the two
        // output lists, the running products and the descriptors of a chunk.
        // (default 64 GB since round 3: a 2^20-pair G2 MSM then runs as ONE chunk -- 14.0 -> 14.4 M/s on MNT4 G2; the budget is cut to what
        //  is free next to the key anyway)
        static const double env_scratch_gb = getenv("GH_AFF_SCRATCH_GB") ? atof(getenv("GH_AFF_SCRATCH_GB")) : 64.0;
        uint32_t K = 1;
        {
            size_t free_b = 0, total_b = 0;
            HIPCHK(hipMemGetInfo(&free_b, &total_b));
            size_t have = 0;
            const char* names[6] = {"aff_desc", "aff_ptsA", "aff_ptsB", "aff_prefix", "aff_stage1", "aff_stage2"};
            for (int i = 0; i < 6; i++) { snprintf(nm, sizeof nm, "%s#%d", names[i], slot); have += pool_cap(nm); }
            double budget = env_scratch_gb * 1073741824.0;
            const double avail = ((double)free_b + (double)have - 3.0 * 1073741824.0) * 0.9;     // what this slot may hold at most
            if (budget > avail) budget = avail;
            const double need = 430.0 * LANES * (double)n0 * 1.13;                                // incl. the pool's 1/8 slack
            if (budget < 256.0 * 1048576.0) {       // no room at all: the projective kernel runs
                for (int i = 0; i < 6; i++) { snprintf(nm, sizeof nm, "%s#%d", names[i], slot); pool_release(nm); }
                tree = false;
                return GH_OK;
            }
            while ((double)K * budget < need && K < 4096) K++;
        }
        uint32_t *d_bq, *d_tab;
        POOLT("aff_bq", d_bq, ((size_t)K + 2) * 4)
        POOLT("aff_tab", d_tab, ((size_t)K + 2) * (R + 1) * 4)
        GH_LAUNCH(aff_chunks_kernel, dim3((K + 1 + 63) / 64), dim3(64), 0, st, (const uint32_t*)starts, (const uint32_t*)counts,
                           (const uint32_t*)aff_st, (const uint32_t*)aff_cnt, (uint32_t)total, R, stride, K, d_bq, d_tab);
        HIPCHK(hipGetLastError());
        std::vector<uint32_t> bq((size_t)K + 1), tab(((size_t)K + 1) * (R + 1));
        HIPCHK(hipMemcpyAsync(bq.data(), d_bq, bq.size() * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(tab.data(), d_tab, tab.size() * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        auto T = [&](uint32_t j, int r) { return tab[(size_t)j * (R + 1) + r]; };
        if (T(K, 0) != n0) { g_err = "internal: affine plan disagrees with the sort stage"; return GH_E_HIP; }
        // the largest chunk sizes every list
        uint32_t max_n1 = 0, max_n2 = 0;
        size_t max_desc = 0;
        for (uint32_t j = 0; j < K; j++) {
            const uint32_t n1 = T(j + 1, 1) - T(j, 1), n2 = R >= 2 ? T(j + 1, 2) - T(j, 2) : 0;
            if (n1 > max_n1) max_n1 = n1;
            if (n2 > max_n2) max_n2 = n2;
            size_t dsum = 0;
            for (int r = 1; r <= R; r++) dsum += T(j + 1, r) - T(j, r);
            if (dsum > max_desc) max_desc = dsum;
        }
        uint32_t* desc;
        void *ptsA, *ptsB, *prefix, *stage1, *stage2;     // T64 lists (aff_kernels.h)
        auto tiles = [&](uint32_t n_el) { return ((size_t)n_el + TPW - 1) / TPW + 1; };
#undef POOLT
#define POOLBIG(name, ptr, bytes)                                   \
    snprintf(nm, sizeof nm, "%s#%d", name, slot);                   \
    rc = pool_get(nm, bytes, (void**)&ptr);                         \
    if (rc == GH_E_NOMEM) { (void)hipGetLastError(); tree = false; return GH_OK; } \
    if (rc) return rc;
        POOLBIG("aff_desc", desc, (max_desc + 64) * 4)
        POOLBIG("aff_ptsA", ptsA, t64_bytes(tiles(max_n1), T64_PT_CHUNKS))
        POOLBIG("aff_ptsB", ptsB, t64_bytes(tiles(max_n2), T64_PT_CHUNKS))
        POOLBIG("aff_prefix", prefix, t64_bytes(tiles(max_n1), T64_FP_CHUNKS))
        POOLBIG("aff_stage1", stage1, t64_bytes(tiles(max_n1), T64_PT_CHUNKS))
        POOLBIG("aff_stage2", stage2, t64_bytes(tiles(max_n1), T64_PT_CHUNKS))
#undef POOLBIG
        const uint32_t max_waves = (uint32_t)g.num_cus * 4u * (uint32_t)FS::WAVES;
        const bool aff_asm = (C::F::DEG >= 2 ? gh_asm::aff_enabled() : gh_asm::aff_g1_enabled()) && !h->aff_asm_off;
        const int asm_kind = std::is_same<C, Mnt4G2>::value ? 0 : (std::is_same<C, Mnt6G2>::value ? 1 : (std::is_same<C, Mnt6G1>::value ? 3 : 2));
        // the assembly kernels run two waves per SIMD; GH_AFF_WAVES_MUL x that many waves are launched so that the blocks (equal
        // work each) are dealt out dynamically instead of as one exact fill of the chip
        static const int env_wmul = getenv("GH_AFF_WAVES_MUL") ? atoi(getenv("GH_AFF_WAVES_MUL")) : 1;
        const uint32_t asm_max_waves = (uint32_t)g.num_cus * 4u * 2u * (uint32_t)(env_wmul > 0 && env_wmul <= 16 ? env_wmul : 1);
        void* asm_accs = nullptr;
        uint32_t* asm_flag = nullptr;
        // Two copies of the per-round control data: a large round is issued as two halves (below)
        const size_t accs_half = t64_bytes((size_t)asm_max_waves + 4, T64_FP_CHUNKS);
        const size_t flag_words = 16 + (size_t)AFF_FIX_CAP;
        if (aff_asm) {
            snprintf(nm, sizeof nm, "aff_accs#%d", slot);
            if ((rc = pool_get(nm, 2 * accs_half, &asm_accs))) return rc;
            snprintf(nm, sizeof nm, "aff_flag#%d", slot);
            if ((rc = pool_get(nm, 2 * 4 * flag_words, (void**)&asm_flag))) return rc;   // control block + exception list, per half
            HIPCHK(hipMemsetAsync(asm_flag, 0, 64, st));                    // word 4: "a round of this MSM was redone" (sticky)
            HIPCHK(hipMemsetAsync(asm_flag + flag_words, 0, 64, st));
        }
        // A round = forward kernel, tower inversion of the lane groups' running products, backward kernel.  The inversion is
        // 0.4 ms of latency with the card nearly idle.  A large round therefore goes out as two halves of its output range on two
        // streams, the second half one kernel behind the first: the inversion of either half runs beside a forward / backward
        // kernel of the other (GH_AFF_SPLIT=0: one piece; halves are whole tiles, so every list keeps its layout).
        static const int env_split = getenv("GH_AFF_SPLIT") ? atoi(getenv("GH_AFF_SPLIT")) : 1;
        static const int env_split_b = getenv("GH_AFF_SPLIT_B") ? atoi(getenv("GH_AFF_SPLIT_B")) : 32;   // smallest batch per lane group worth splitting
        const Aff<C>* rows = (const Aff<C>*)(merged ? h->d_table : h->d_points);
        // one piece of a round: outputs [o0, o0 + n_piece) (o0 a multiple of the tile size); which: 0 / 1 = control block, stream role
        struct Piece {
            AffRoundArgs<C> a;
            gh_asm::AffArgs q;
            uint32_t waves_cpp, aw, Bq;
            uint32_t* flag;
            void* accs;
        };
        auto make_piece = [&](int r, uint32_t j, size_t doff, const void* in, void* out, uint32_t o0, uint32_t n_piece, int which) {
            Piece P;
            const size_t t0 = o0 / TPW;                                    // first tile of the piece in every list
            auto off = [&](void* base, int chunks) { return (void*)((char*)base + t64_bytes(t0, chunks)); };
            uint32_t waves = (n_piece + TPW * (uint32_t)env_bmin - 1) / (TPW * (uint32_t)env_bmin);
            if (waves > max_waves) waves = max_waves;
            waves = (waves + 3u) & ~3u;
            AffRoundArgs<C>& a = P.a;
            a.rows = rows; a.in = in; a.sorted = r == 0 ? sorted : nullptr; a.desc = desc + doff + o0; a.n_out = n_piece; a.in_base = T(j, r);
            a.prefix = off(prefix, T64_FP_CHUNKS); a.out = off(out, T64_PT_CHUNKS);
            a.stage1 = off(stage1, T64_PT_CHUNKS); a.stage2 = off(stage2, T64_PT_CHUNKS);
            a.groups = waves * TPW; a.bmin = (uint32_t)env_bmin;
            a.run_if = nullptr;
            P.waves_cpp = waves;
            uint32_t aw = (n_piece + TPW * (uint32_t)env_bmin - 1) / (TPW * (uint32_t)env_bmin);
            if (aw > asm_max_waves) aw = asm_max_waves;
            aw = (aw + 3u) & ~3u;
            uint32_t Bq = (n_piece + aw * TPW - 1) / (aw * TPW);
            if (Bq < (uint32_t)env_bmin) Bq = (uint32_t)env_bmin;
            P.aw = aw; P.Bq = Bq;
            P.flag = asm_flag ? asm_flag + (size_t)which * flag_words : nullptr;
            P.accs = asm_accs ? (void*)((char*)asm_accs + (size_t)which * accs_half) : nullptr;
            gh_asm::AffArgs& q = P.q;
            q.in = r == 0 ? (const void*)rows : in; q.sorted = sorted; q.desc = a.desc; q.prefix = a.prefix;
            q.stage1 = a.stage1; q.stage2 = a.stage2; q.out = a.out; q.accs = P.accs; q.flag = P.flag;
            q.n_out = n_piece; q.in_base = T(j, r); q.B = Bq; q.pad = 0;
            return P;
        };
        // The assembly kernels (asmgen/g2_rounds.py): forward pass, tower inversion of the lane groups' running products, backward
        // pass -- 256 registers, two waves per SIMD, no scratch, no out-of-line product.  Elements on the group law's rare branches
        // go through an exception list (aff_fix_kernel); only if the list overflowed, the whole piece once more on the C++ kernel.
        auto issue_fwd = [&](Piece& P, int r, hipStream_t s_) -> int {
            HIPCHK(hipMemsetAsync(P.flag, 0, 16, s_));
            return gh_asm::aff_launch(asm_kind, true, r == 0, P.q, P.aw, s_);
        };
        auto issue_rest = [&](Piece& P, int r, hipStream_t s_) -> int {
            int rc2;
            GH_LAUNCH((aff_inv_kernel<FS>), dim3(P.aw / 4), dim3(256), 0, s_, P.accs, P.aw, P.a.n_out, P.Bq, (const uint32_t*)P.flag);
            if ((rc2 = gh_asm::aff_launch(asm_kind, false, r == 0, P.q, P.aw, s_))) return rc2;
            if (r == 0) GH_LAUNCH((aff_fix_kernel<C, FS, true>), dim3(16), dim3(256), 0, s_, P.a, (const uint32_t*)P.flag);
            else GH_LAUNCH((aff_fix_kernel<C, FS, false>), dim3(16), dim3(256), 0, s_, P.a, (const uint32_t*)P.flag);
            P.a.run_if = P.flag;
            return GH_OK;
        };
        auto issue_cpp = [&](Piece& P, int r, hipStream_t s_) -> int {   // the C++ round kernel: the whole piece, or (run_if) its fallback
            if (r == 0) GH_LAUNCH((aff_round_kernel<C, FS, true>), dim3(P.waves_cpp / 4), dim3(256), 0, s_, P.a);
            else GH_LAUNCH((aff_round_kernel<C, FS, false>), dim3(P.waves_cpp / 4), dim3(256), 0, s_, P.a);
            return GH_OK;
        };
        for (uint32_t j = 0; j < K; j++) {
            size_t doff = 0;
            const void* in = nullptr;
            for (int r = 0; r < R; r++) {
                const uint32_t n_out = T(j + 1, r + 1) - T(j, r + 1);
                void* out = (r & 1) ? ptsB : ptsA;
                if (n_out > 0) {
                    unsigned dgrid = (n_out + 255) / 256;
                    if (dgrid > 16384) dgrid = 16384;
                    const uint32_t* st_in = r == 0 ? starts : aff_st + (size_t)(r - 1) * stride;
                    const uint32_t* m_in = r == 0 ? counts : aff_cnt + (size_t)(r - 1) * stride;
                    GH_LAUNCH(aff_desc_kernel, dim3(dgrid), dim3(256), 0, st, st_in, m_in,
                                       (const uint32_t*)(aff_st + (size_t)r * stride), (uint32_t)total, T(j, r + 1), n_out, desc + doff);
                    // halves: whole tiles, the first one a multiple of four tiles
                    uint32_t nA = ((n_out / 2 + 4 * TPW - 1) / (4 * TPW)) * (4 * TPW);
                    bool split = aff_asm && env_split != 0 && nA < n_out;
                    if (split) {
                        const uint32_t whole = (uint32_t)(((size_t)n_out + (size_t)asm_max_waves * TPW - 1) / ((size_t)asm_max_waves * TPW));
                        split = whole >= (uint32_t)env_split_b;          // batch per lane group if the round went out in one piece
                    }
                    if (!aff_asm) {
                        Piece P = make_piece(r, j, doff, in, out, 0, n_out, 0);
                        if ((rc = issue_cpp(P, r, st))) return rc;
                    } else if (!split) {
                        Piece P = make_piece(r, j, doff, in, out, 0, n_out, 0);
                        if ((rc = issue_fwd(P, r, st))) return rc;
                        if ((rc = issue_rest(P, r, st))) return rc;
                        static const bool aff_debug = getenv("GH_AFF_DEBUG") != nullptr;
                        if (aff_debug) {      // how many elements of the round went through the exception list / whether it overflowed
                            uint32_t fw[4] = {0, 0, 0, 0};
                            HIPCHK(hipStreamSynchronize(st));
                            HIPCHK(hipMemcpy(fw, P.flag, 16, hipMemcpyDeviceToHost));
                            fprintf(stderr, "[gh aff] chunk %u round %d n_out %u waves %u B %u in_base %u redo %u exceptions %u\n", j, r, n_out, P.aw, P.Bq,
                                    P.q.in_base, fw[0], fw[1]);
                        }
                        if ((rc = issue_cpp(P, r, st))) return rc;
                    } else {
                        Piece A = make_piece(r, j, doff, in, out, 0, nA, 0);
                        Piece B = make_piece(r, j, doff, in, out, nA, n_out - nA, 1);
                        hipStream_t st2 = g.stream_acc2;
                        if ((rc = issue_fwd(A, r, st))) return rc;
                        HIPCHK(hipEventRecord(g.tev[0], st));                 // the round's inputs are complete and A's forward pass is out
                        HIPCHK(hipStreamWaitEvent(st2, g.tev[0], 0));
                        if ((rc = issue_fwd(B, r, st2))) return rc;
                        if ((rc = issue_rest(A, r, st))) return rc;
                        if ((rc = issue_rest(B, r, st2))) return rc;
                        if ((rc = issue_cpp(A, r, st))) return rc;
                        if ((rc = issue_cpp(B, r, st2))) return rc;
                        HIPCHK(hipEventRecord(g.tev[1], st2));
                        HIPCHK(hipStreamWaitEvent(st, g.tev[1], 0));          // join: the next round reads both halves
                    }
                }
                doff += n_out;
                in = out;
            }
            // what is left of the chunk's buckets (a few points each): projective, one bucket per thread / lane group
            const uint32_t nbk = bq[j + 1] - bq[j];
            if (nbk == 0) continue;
            const uint32_t* stR = aff_st + (size_t)(R - 1) * stride;
            const uint32_t* mR = aff_cnt + (size_t)(R - 1) * stride;
            if constexpr (C::F::DEG == 1) {
                GH_LAUNCH((msm_accumulate_xyzz_kernel<C, true>), dim3((nbk + 255) / 256), dim3(256), 0, st, (const Aff<C>*)in,
                                   (const uint32_t*)nullptr, stR, mR, (const uint32_t*)nullptr, nbk, (const Aff<C>*)salts, buckets,
                                   (const uint32_t*)nullptr, 0u, 0u, heavy_chunk, (Proj<C>*)nullptr, bq[j], T(j, R));
            } else {
                typedef typename std::conditional<C::F::DEG == 2, F2S<P4, 13, GH_F2S_DUAL != 0>, F3S<P6, 11, GH_F3S_TRIPLE>>::type FA;
                const size_t fwaves = ((size_t)nbk + TPW - 1) / TPW;
                GH_LAUNCH((msm_accumulate_split_kernel<C, FA, LANES, true>), dim3((unsigned)((fwaves * 64 + 255) / 256)), dim3(256), 0, st,
                                   (const Aff<C>*)in, (const uint32_t*)nullptr, stR, mR, (const uint32_t*)nullptr, nbk, (const Aff<C>*)salts, buckets,
                                   (const uint32_t*)nullptr, 0u, 0u, heavy_chunk, (Proj<C>*)nullptr, bq[j], T(j, R));
            }
        }
        HIPCHK(hipGetLastError());
        if (aff_asm) {        // read with the window sums in finish(): a key that overflows the exception list leaves the assembly rounds
            HIPCHK(hipMemcpyAsync(&hplan[16], asm_flag + 4, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(&hplan[17], asm_flag + flag_words + 4, 4, hipMemcpyDeviceToHost, st));
            aff_sticky_pending = true;
        }
        n_heavy = 0;   // no chunk sums to combine
        return GH_OK;
    }

    // stage 3 (stream st): the two wave-program levels of the bucket reduction; window sums -> host
    int launch_reduce(hipStream_t st) {
        if (n == 0) return GH_OK;
        const size_t lds_wave = 64 * sizeof(Proj<C>);
        // level 1: one wave per segment of 64 * L1 bucket slots -> (runW, A, Bv) per segment
        WaveReduceIn<C> i0{buckets, 1, 0, Q, 0, (uint32_t)total}, none{nullptr, 0, 0, 0, 0, 0};
        static const int env_wpb = getenv("GH_REDUCE_WPB") ? atoi(getenv("GH_REDUCE_WPB")) : 1;
        int wpb = env_wpb >= 1 && (size_t)env_wpb * lds_wave <= 65536 && env_wpb <= 4 ? env_wpb : 1;   // waves per block
        const unsigned nb1 = (unsigned)(RW * segs_per_window), nb2 = (unsigned)(3 * RW);
        const uint32_t all = 0xFFFFFFFFu;
        // level 2: one wave per window and per array: weighted program on runW, plain sums of A and Bv
        WaveReduceIn<C> r0{seg_out, 3, 0, segs_per_window, 0, all}, r1{seg_out, 3, 1, segs_per_window, 1, all}, r2{seg_out, 3, 2, segs_per_window, 1, all};
        bool launched = false;
        if constexpr (C::F::DEG >= 2) {
            if (tpw != 64) {
                typedef typename std::conditional<C::F::DEG == 2, F2S<P4, 13>, F3S<P6, 11, GH_F3S_TRIPLE>>::type FS;
                constexpr int LANES = FS::LANES, TPW = C::F::DEG == 2 ? 32 : 16;
                const size_t lds_split = 64 * sizeof(P3);
                uint32_t* slabs = nullptr;
                char nm[48];
                snprintf(nm, sizeof nm, "reduce_slabs#%d", slot);
                int rc = pool_get(nm, (size_t)(nb1 > nb2 ? nb1 : nb2) * P3Slab::WORDS * 4, (void**)&slabs);
                if (rc) return rc;
                GH_LAUNCH((msm_wave_reduce_split_kernel<C, FS, LANES, TPW>), dim3(nb1), dim3(64), lds_split, st,
                                   i0, none, none, nb1, 1u, segs_per_window, L1, (const Aff<C>*)salts, seg_out, slabs);
                GH_LAUNCH((msm_wave_reduce_split_kernel<C, FS, LANES, TPW>), dim3(nb2), dim3(64), lds_split, st,
                                   r0, r1, r2, (uint32_t)RW, 3u, 1u, L2, (const Aff<C>*)salts, win_out, slabs);
                launched = true;
            }
        }
        if constexpr (C::F::DEG == 1) {
        if (!launched) {
            // the programs' accumulators live in a slab of global memory each (msm_kernels.h, ReduceSlab)
            uint32_t* slabs = nullptr;
            {
                char nm[48];
                snprintf(nm, sizeof nm, "reduce_slabs#%d", slot);
                const size_t progs = nb1 > nb2 ? nb1 : nb2;
                int rc = pool_get(nm, progs * ReduceSlab<C>::WORDS * 4, (void**)&slabs);
                if (rc) return rc;
            }
            // A stand-alone MSM has the chip to itself: the 512-register build of the program (one wave per SIMD, 88 B of spills
            // per lane instead of 680) -- inside a batch the reduction must fit beside the accumulation's waves (256 registers).
            static const int env_w = getenv("GH_REDUCE_WAVES") ? atoi(getenv("GH_REDUCE_WAVES")) : 0;
            const bool one_wave = env_w ? env_w == 1 : solo;
            if (lean) {
                // level 1: serial part only, (run, wacc) per lane; level 2 per window: the weighted program over the lanes' run (item =
                // segment * 64 + lane, so its A = sum segment * run and Bv = sum lane * run) and the plain sum of their wacc
                const uint32_t lanes_per_window = (uint32_t)segs_per_window * 64u;
                WaveReduceIn<C> i0l{buckets, 1, 0, Q, 2, (uint32_t)total};
                WaveReduceIn<C> l0{lane_out, 2, 0, lanes_per_window, 0, all}, l1{lane_out, 2, 1, lanes_per_window, 1, all};
                const unsigned nb2l = (unsigned)(2 * RW);
                if (one_wave) {
                    GH_LAUNCH((msm_wave_reduce_kernel<C, 1>), dim3((nb1 + wpb - 1) / wpb), dim3(64 * wpb), lds_wave * wpb, st,
                                       i0l, none, none, nb1, 1u, segs_per_window, L1, (const Aff<C>*)salts, lane_out, slabs);
                    GH_LAUNCH((msm_wave_reduce_kernel<C, 1>), dim3((nb2l + wpb - 1) / wpb), dim3(64 * wpb), lds_wave * wpb, st,
                                       l0, l1, none, (uint32_t)RW, 2u, 1u, (int)segs_per_window, (const Aff<C>*)salts, win_out, slabs);
                } else {
                    GH_LAUNCH((msm_wave_reduce_kernel<C, 2>), dim3((nb1 + wpb - 1) / wpb), dim3(64 * wpb), lds_wave * wpb, st,
                                       i0l, none, none, nb1, 1u, segs_per_window, L1, (const Aff<C>*)salts, lane_out, slabs);
                    GH_LAUNCH((msm_wave_reduce_kernel<C, 2>), dim3((nb2l + wpb - 1) / wpb), dim3(64 * wpb), lds_wave * wpb, st,
                                       l0, l1, none, (uint32_t)RW, 2u, 1u, (int)segs_per_window, (const Aff<C>*)salts, win_out, slabs);
                }
            } else if (one_wave) {
                GH_LAUNCH((msm_wave_reduce_kernel<C, 1>), dim3((nb1 + wpb - 1) / wpb), dim3(64 * wpb), lds_wave * wpb, st,
                                   i0, none, none, nb1, 1u, segs_per_window, L1, (const Aff<C>*)salts, seg_out, slabs);
                GH_LAUNCH((msm_wave_reduce_kernel<C, 1>), dim3((nb2 + wpb - 1) / wpb), dim3(64 * wpb), lds_wave * wpb, st,
                                   r0, r1, r2, (uint32_t)RW, 3u, 1u, L2, (const Aff<C>*)salts, win_out, slabs);
            } else {
                GH_LAUNCH((msm_wave_reduce_kernel<C, 2>), dim3((nb1 + wpb - 1) / wpb), dim3(64 * wpb), lds_wave * wpb, st,
                                   i0, none, none, nb1, 1u, segs_per_window, L1, (const Aff<C>*)salts, seg_out, slabs);
                GH_LAUNCH((msm_wave_reduce_kernel<C, 2>), dim3((nb2 + wpb - 1) / wpb), dim3(64 * wpb), lds_wave * wpb, st,
                                   r0, r1, r2, (uint32_t)RW, 3u, 1u, L2, (const Aff<C>*)salts, win_out, slabs);
            }
        }
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(g.pev[es][5], st));
        HIPCHK(hipMemcpyAsync(hw, win_out, (size_t)9 * RW * sizeof(Proj<C>), hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(g.pev[es][6], st));
        return GH_OK;
    }

    // stage 4 (host): wait for the window sums, fold
    int finish() {
        if (n == 0) {
            proj_to_abi_host<C>(out_xyz, proj_zero<C>());
            g.last_msm = gh_msm_timing_t{};
            g.batch_tm.push_back(g.last_msm);
            return GH_OK;
        }
        HIPCHK(hipEventSynchronize(g.pev[es][6]));
        if (aff_sticky_pending && (hplan[16] != 0 || hplan[17] != 0)) h->aff_asm_off = 1;
        auto t_fold0 = std::chrono::steady_clock::now();
        std::vector<Proj<C>> hwv(hw, hw + (size_t)9 * RW);
        if (lean) {
            // level 2 wrote (T_w, A2, Bv2) for the lanes' run and PA' = sum of the lanes' wacc:
            //   R_w = 64 PA' + 64 L1 A2 + Bv2   ->   the fold's slots PW = 0, PS = A2 (weight 2^u = 64 L1), PA = PA', PB = Bv2
            for (int w = 0; w < RW; w++) {
                const Proj<C> t = hw[(size_t)w * 3], a2 = hw[(size_t)w * 3 + 1], bv2 = hw[(size_t)w * 3 + 2], pa = hw[(size_t)(RW + w) * 3];
                hwv[(size_t)w * 3] = t; hwv[(size_t)w * 3 + 1] = proj_zero<C>(); hwv[(size_t)w * 3 + 2] = a2;
                hwv[(size_t)(RW + w) * 3] = pa;
                hwv[(size_t)(2 * RW + w) * 3] = bv2;
            }
        }
        // Window sum R_w = 64 PA + PB + U (64 PW + PS), U = 64 L1 = 2^u, with
        //   PW, PS = (A, Bv) of the weighted level-2 program over the runW's, PA = sum A, PB = sum Bv.
        // Horner over windows, high to low (variable_base.rs:73-82), with the powers of two of R_w
        // merged into the c doublings between windows:
        //   acc*2^c + R_w = (((acc*2^(c-u-6) + PW)*2^6 + PS)*2^(u-6) + PA)*2^6 + PB        (c >= u + 6)
        int u = sw;
        while ((1 << (u - sw)) < L1) u++;
        if (merged) {
            int lq = 0;
            while ((1u << lq) < Q) lq++;            // RW > 1 only with Q = 2^q; for RW == 1 the term is empty
            fold_merged<C>(hwv, RW, lq, u, sw, sets, c, out_xyz);
        } else {
            fold_windows<C>(hwv, W, c, u, sw, top_unsigned, out_xyz);
        }
        auto t_end = std::chrono::steady_clock::now();
        HIPCHK(hipEventElapsedTime(&tm.sort_ms, g.pev[es][0], g.pev[es][1]));
        HIPCHK(hipEventElapsedTime(&tm.accumulate_ms, g.pev[es][2], g.pev[es][3]));   // brackets exactly the accumulation launch
        HIPCHK(hipEventElapsedTime(&tm.heavy_ms, g.pev[es][3], g.pev[es][4]));
        HIPCHK(hipEventElapsedTime(&tm.reduce_ms, g.pev[es][4], g.pev[es][5]));
        tm.heavy_buckets = n_heavy;
        tm.fold_ms = std::chrono::duration<float, std::milli>(t_end - t_fold0).count();
        tm.total_ms = std::chrono::duration<float, std::milli>(t_end - t_begin).count();
        tm.window_bits = c;
        tm.num_windows = W;
        g.last_msm = tm;
        g.batch_tm.push_back(tm);
        return GH_OK;
    }
};

// The accumulation kernels over caller-made lists (fixed_base.hip: one "bucket" per scalar, its list the table entries its
// digits select): out[g] = sum of points[sorted[starts[g] + k] & 0x7FFFFFFF] (bit 31: negated), k < counts[g], for the
// `total` lists in the order `order` names them; no heavy-bucket chunks.  Same kernels, same complete addition (doubling through
// the salt detour, P + (-P), infinity) as the MSM's projective path.
template <class C>
int accumulate_lists(const void* points, const uint32_t* sorted, const uint32_t* starts, const uint32_t* counts,
                     const uint32_t* order, uint32_t total, void* out_proj, hipStream_t st) {
    if (total == 0) return GH_OK;
    Aff<C>* salts = nullptr;
    if (int rc = device_salts<C>(&salts)) return rc;
    if constexpr (C::F::DEG == 1) {
      if (gh_asm::enabled()) {
        gh_asm::AccTask* tk = nullptr;
        if (int rc = pool_get("acc_tasks#lists", (size_t)total * sizeof(gh_asm::AccTask), (void**)&tk)) return rc;
        GH_LAUNCH((msm_acc_tasks_kernel<C>), dim3((unsigned)(((size_t)total + 255) / 256)), dim3(256), 0, st, starts, counts, order,
                           total, (Proj<C>*)out_proj, (const uint32_t*)nullptr, 0u, 0u, 0u, (Proj<C>*)nullptr, (AccTaskRec*)tk);
        if (int rc = gh_asm::acc_g1_launch(std::is_same<typename C::PF, P6>::value ? 6 : 4, points, sorted, tk, salts, total, st)) return rc;
      } else {
        GH_LAUNCH((msm_accumulate_xyzz_kernel<C>), dim3((unsigned)(((size_t)total + 255) / 256)), dim3(256), 0, st,
                           (const Aff<C>*)points, sorted, starts, counts, order, total, (const Aff<C>*)salts, (Proj<C>*)out_proj,
                           (const uint32_t*)nullptr, 0u, 0u, 0u, (Proj<C>*)nullptr, 0u, 0u);
      }
    } else {
        typedef typename std::conditional<std::is_same<C, Mnt4G2>::value, F2S<P4, 13, GH_F2S_DUAL != 0>, F3S<P6, 11, GH_F3S_TRIPLE>>::type FS;
        constexpr int LANES = FS::LANES;
        const size_t waves = ((size_t)total + (64 / LANES) - 1) / (64 / LANES);
        GH_LAUNCH((msm_accumulate_split_kernel<C, FS, LANES>), dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, st,
                           (const Aff<C>*)points, sorted, starts, counts, order, total, (const Aff<C>*)salts, (Proj<C>*)out_proj,
                           (const uint32_t*)nullptr, 0u, 0u, 0u, (Proj<C>*)nullptr);
    }
    HIPCHK(hipGetLastError());
    return GH_OK;
}

// `count` MSMs back to back.  With count > 1 the stages are pipelined over three streams and two
// buffer slots: while MSM k accumulates (stream_acc), the bucket sort of MSM k+1 (g.stream) and the
// bucket reduction + host fold of MSM k-1 (stream_red, host) run beside it -- the sort is
// atomics/memory bound and the reduction's wave programs are latency chains, so both fit into the
// issue slots the accumulation leaves.  Slot reuse is ordered by events (one event set per job in flight, g.pev[k & 3]):
//   sort(k+1) waits for acc(k-1) (lists of that slot), acc(k+1) for reduce(k-1) (its buckets).
// Not free: leaving the reduction out of a batch (measurement only) takes the 2^20-pair G1 MSM from 23.4 to 20.1 ms -- the
// reduction's instructions are issued at the accumulation's expense.  Hence the lean form of the reduction for the MSMs
// of a batch that have a successor to hide its longer chain behind (MsmJob::prepare).
template <class C>
int msm_batch(BasesBase* const* hs, const void* const* d_scalars, const size_t* n_scalars, int count, uint64_t* out_xyz) {
    const size_t out_stride = (size_t)36 * C::F::DEG;
    std::vector<MsmJob<C>> jobs((size_t)count);
    int rc;
    auto issue_sort = [&](int k) -> int {
        MsmJob<C>& j = jobs[(size_t)k];
        j.solo = count == 1;
        j.last = k == count - 1;
        j.es = k & 3;
        if ((rc = j.prepare(hs[k], d_scalars[k], n_scalars[k], out_xyz + (size_t)k * out_stride, k & 1))) return rc;
        if (k >= 2) {
            HIPCHK(hipStreamWaitEvent(g.stream, g.pev[(k - 2) & 3][4], 0));   // acc(k-2) has consumed this slot's lists
        }
        return j.launch_sort(g.stream);
    };
    g.batch_tm.clear();
    if (count <= 0) return GH_OK;
    if ((rc = issue_sort(0))) return rc;
    for (int k = 0; k < count; k++) {
        MsmJob<C>& j = jobs[(size_t)k];
        if (k >= 2 && j.n) HIPCHK(hipStreamWaitEvent(g.stream_acc, g.pev[(k - 2) & 3][6], 0));   // reduce(k-2) is done with this slot's buckets
        if ((rc = j.launch_accumulate(g.stream_acc))) return rc;
        if (j.n) HIPCHK(hipStreamWaitEvent(g.stream_red, g.pev[k & 3][4], 0));
        if ((rc = j.launch_reduce(g.stream_red))) return rc;
        // sort(k+1) goes out before the host waits for the window sums of k-1: the two share a buffer slot but no buffer
        // (lists and plan of the slot were consumed by acc(k-1); buckets, segment sums and the pinned window sums are the
        // reduction's) and, since round 3, no events -- so the sort of the next MSM no longer starts only when the
        // reduction of the previous one has ended (11 ms into acc(k), more with the lean reduction's longer chain).
        if (k + 1 < count && (rc = issue_sort(k + 1))) return rc;
        if (k >= 1 && (rc = jobs[(size_t)k - 1].finish())) return rc;
    }
    return jobs[(size_t)count - 1].finish();
}

template <class C>
int msm_run(BasesBase* h, const void* d_scalars, size_t n_scalars, uint64_t* out_xyz) {
    return msm_batch<C>(&h, &d_scalars, &n_scalars, 1, out_xyz);
}

template <class C>
int msm_host(const uint64_t* bases, const uint8_t* infinity, size_t n_bases, const uint64_t* scalars, size_t n_scalars,
             uint64_t* out_xyz) {
    size_t n = n_bases < n_scalars ? n_bases : n_scalars;
    BasesBase* h = nullptr;
    int rc = upload_bases<C>(bases, infinity, n, 0, &h);
    if (rc) return rc;
    void* d_s = nullptr;
    if (n > 0) {
        rc = pool_get("scalars", n * 96, &d_s);
        if (!rc) {
            hipError_t e = hipMemcpyAsync(d_s, scalars, n * 96, hipMemcpyHostToDevice, g.stream);
            if (e != hipSuccess) { g_err = hipGetErrorString(e); rc = GH_E_HIP; }
        }
    }
    if (!rc) rc = msm_run<C>(h, d_s, n, out_xyz);
    if (h->d_points) hipFree(h->d_points);
    if (h->d_inf) hipFree(h->d_inf);
    delete h;
    return rc;
}


template <class C> int proj_add_host(uint64_t* acc_xyz, const uint64_t* p_xyz) {
    Proj<C> a = proj_from_abi_host<C>(acc_xyz), b = proj_from_abi_host<C>(p_xyz);
    proj_to_abi_host<C>(acc_xyz, proj_add<C>(a, b));
    return GH_OK;
}

// out = k * p for one point (the prover's r * delta_g1, s * g_a, ... of prover.rs:278-330): double-and-add
// from the top bit like GroupProjective::mul_assign (short_weierstrass_projective.rs:521-540), on the
// 64-bit-limb host field.  Host side; ~1 ms.
template <class C> int proj_mul_host(const uint64_t* p_xyz, const uint64_t* scalar12, uint64_t* out_xyz) {
    typedef typename HostCurveOf<C>::type HC;
    static_assert(HostCurveOf<C>::fast, "host curve on ABI limbs");
    Proj<HC> p, res = proj_zero<HC>();
    memcpy(&p, p_xyz, sizeof(p));
    bool found_one = false;
    for (int bit = 767; bit >= 0; bit--) {
        const bool b = (scalar12[bit >> 6] >> (bit & 63)) & 1u;
        if (found_one) res = proj_dbl<HC>(res);
        if (b) { res = proj_add<HC>(res, p); found_one = true; }
    }
    if (proj_is_zero<HC>(res)) res = proj_zero<HC>();
    memcpy(out_xyz, &res, sizeof(res));
    return GH_OK;
}

template <class C> int proj_neg_host(uint64_t* xyz) {   // (X, Y, Z) -> (X, -Y, Z)   (swp.rs Neg)
    typedef typename HostCurveOf<C>::type HC;
    Proj<HC> p;
    memcpy(&p, xyz, sizeof(p));
    if (!proj_is_zero<HC>(p)) p.y = HC::F::neg(p.y);
    memcpy(xyz, &p, sizeof(p));
    return GH_OK;
}

template <class C> int to_affine_host(const uint64_t* xyz, uint64_t* out_xy, uint8_t* is_infinity) {
    typedef typename C::F F;
    Proj<C> p = proj_from_abi_host<C>(xyz);
    uint32_t* w = reinterpret_cast<uint32_t*>(out_xy);
    if (proj_is_zero<C>(p)) {  // GroupAffine::zero() = (0, 1, infinity)  (swp.rs:130-132)
        *is_infinity = 1;
        F::to_abi(w, F::zero());
        F::to_abi(w + 24 * F::DEG, F::one());
        return GH_OK;
    }
    *is_infinity = 0;
    typename F::T zi = host_inv<F>(p.z);
    F::to_abi(w, F::mul(p.x, zi));
    F::to_abi(w + 24 * F::DEG, F::mul(p.y, zi));
    return GH_OK;
}

#define GH_DEFINE_MSM_OPS(CURVE, NAME)                                                        \
    namespace gh_rt {                                                                          \
    const MsmOps* NAME() {                                                                     \
        static const MsmOps ops = {&upload_bases<CURVE>, &msm_run<CURVE>, &msm_host<CURVE>,    \
                                   &proj_add_host<CURVE>, &to_affine_host<CURVE>,              \
                                   &precompute_bases<CURVE>, &msm_batch<CURVE>,                \
                                   &proj_mul_host<CURVE>, &proj_neg_host<CURVE>, &accumulate_lists<CURVE>};           \
        return &ops;                                                                           \
    }                                                                                          \
    }

}  // namespace gh_rt

// msm_impl.h -- host-side launch sequence of the MSM (templated on the curve policy); included
// by the four msm_<curve>.hip translation units.
#pragma once
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#include "runtime.h"
#include "msm_kernels.h"
#include "host_math.h"

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

template <class C>
int upload_bases(const uint64_t* bases, const uint8_t* infinity, size_t n, BasesBase** out) {
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
        hipLaunchKernelGGL((msm_convert_bases_kernel<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g.stream,
                           (const uint32_t*)d_in, (Aff<C>*)h->d_points, n);
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

template <class C>
int msm_run(BasesBase* h, const void* d_scalars, size_t n_scalars, uint64_t* out_xyz) {
    size_t n = h->n < n_scalars ? h->n : n_scalars;
    auto t_begin = std::chrono::steady_clock::now();
    gh_msm_timing_t tm{};
    if (n == 0) {
        proj_to_abi_host<C>(out_xyz, proj_zero<C>());
        g.last_msm = tm;
        return GH_OK;
    }
    const int c = auto_window(n);
    const int W = 753 / c + 1;
    const uint32_t nb = (1u << (c - 1)) + 1;
    const size_t total = (size_t)W * nb;
    const uint32_t seg_slots = 64 * MSM_REDUCE_L;
    const uint32_t nbp = ((nb + seg_slots - 1) / seg_slots) * seg_slots;
    const uint32_t segs_per_window = nbp / seg_slots;
    if ((size_t)W * n >= ((size_t)1 << 32) || total >= ((size_t)1 << 31)) {
        g_err = "MSM too large for 32-bit bucket offsets";
        return GH_E_UNSUPPORTED;
    }
    // salt points S0 = G, S1 = 2G (internal affine form) for the accumulate kernel's detour
    static Aff<C>* salts = nullptr;
    if (!salts) {
        Aff<C> hs[2];
        if (int src = make_salts<C>(hs)) return src;
        HIPCHK(hipMalloc((void**)&salts, sizeof(hs)));
        HIPCHK(hipMemcpy(salts, hs, sizeof(hs), hipMemcpyHostToDevice));
    }
    // heavy threshold: 4x the mean bucket load, within [128, MSM_MAX_HEAVY_THRESHOLD]
    uint32_t heavy_thr = (uint32_t)((4 * n) >> (c - 1));
    if (heavy_thr < 128) heavy_thr = 128;
    if (heavy_thr > (uint32_t)MSM_MAX_HEAVY_THRESHOLD) heavy_thr = MSM_MAX_HEAVY_THRESHOLD;
    const size_t max_heavy = ((size_t)W * n) / (heavy_thr + 1) + 1;          // buckets with > thr entries
    const size_t max_chunks = ((size_t)W * n) / MSM_HEAVY_CHUNK + max_heavy + 1;
    int32_t* digits; uint32_t *counts, *starts, *cursor, *sorted, *order, *size_hist, *size_cursor, *chunk_start, *plan;
    Proj<C>*buckets, *seg_run, *seg_wacc, *wsums, *partials;
    int rc;
#define POOL(name, ptr, bytes) if ((rc = pool_get(name, bytes, (void**)&ptr))) return rc;
    POOL("digits", digits, (size_t)W * n * 4)
    POOL("counts", counts, total * 4)
    POOL("starts", starts, total * 4)
    POOL("cursor", cursor, total * 4)
    POOL("sorted", sorted, (size_t)W * n * 4)
    POOL("order", order, total * 4)
    POOL("size_hist", size_hist, MSM_SIZE_BINS * 4)
    POOL("size_cursor", size_cursor, MSM_SIZE_BINS * 4)
    POOL("chunk_start", chunk_start, (max_heavy + 2) * 4)
    POOL("plan", plan, 16)
    POOL("buckets", buckets, total * sizeof(Proj<C>))
    POOL("seg_run", seg_run, (size_t)W * segs_per_window * sizeof(Proj<C>))
    POOL("seg_wacc", seg_wacc, (size_t)W * segs_per_window * sizeof(Proj<C>))
    POOL("wsums", wsums, (size_t)W * sizeof(Proj<C>))
#undef POOL
    hipStream_t st = g.stream;
    static const bool dbg = getenv("GH_DEBUG") != nullptr;
#define TRACE(msg)                                                                   \
    if (dbg) {                                                                       \
        HIPCHK(hipStreamSynchronize(st));                                            \
        fprintf(stderr, "[gh] msm %s (n=%zu c=%d W=%d)\n", msg, n, c, W);            \
        fflush(stderr);                                                              \
    }
    TRACE("begin")
    HIPCHK(hipEventRecord(g.ev[0], st));
    HIPCHK(hipMemsetAsync(counts, 0, total * 4, st));
    HIPCHK(hipMemsetAsync(size_hist, 0, MSM_SIZE_BINS * 4, st));
    hipLaunchKernelGGL(msm_digits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const uint32_t*)d_scalars, (const uint8_t*)h->d_inf, n, c, W, nb, digits, counts);
    HIPCHK(hipGetLastError());
    TRACE("digits done")
    if ((rc = device_scan(counts, starts, total, "scan_tmp"))) return rc;
    TRACE("scan done")
    HIPCHK(hipMemcpyAsync(cursor, starts, total * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(msm_size_hist_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, counts, total, heavy_thr, size_hist);
    if ((rc = device_scan(size_hist, size_cursor, MSM_SIZE_BINS, "scan_tmp2"))) return rc;
    hipLaunchKernelGGL(msm_size_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, counts, total, heavy_thr, size_cursor, order);
    hipLaunchKernelGGL(msm_heavy_plan_kernel, dim3(1), dim3(1), 0, st, (const uint32_t*)size_hist, (const uint32_t*)counts,
                       (const uint32_t*)order, chunk_start, plan);
    hipLaunchKernelGGL(msm_scatter_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)W), dim3(256), 0, st,
                       (const int32_t*)digits, n, W, nb, cursor, sorted);
    HIPCHK(hipGetLastError());
    TRACE("scatter done")
    uint32_t hplan[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(hplan, plan, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(g.ev[1], st));
    HIPCHK(hipStreamSynchronize(st));
    const uint32_t n_heavy = hplan[0], n_chunks = hplan[1];
    if (n_heavy > max_heavy || n_chunks > max_chunks) { g_err = "internal: heavy-bucket plan out of range"; return GH_E_HIP; }
    const size_t lds_wave = 64 * sizeof(Proj<C>);
    HIPCHK(hipEventRecord(g.ev[2], st));
    {
        size_t rest = total - n_heavy;
        static const int acc_waves = getenv("GH_ACC_WAVES") ? atoi(getenv("GH_ACC_WAVES")) : 2;  // measured: 29.3 ms vs 34.6 ms at 2^20
        if (acc_waves >= 2 && C::F::DEG == 1)
            hipLaunchKernelGGL((msm_accumulate_kernel<C, (C::F::DEG == 1 ? 2 : 1)>), dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, st,
                               (const Aff<C>*)h->d_points, (const uint32_t*)sorted, (const uint32_t*)starts,
                               (const uint32_t*)counts, (const uint32_t*)order, n_heavy, (uint32_t)total,
                               (const Aff<C>*)salts, buckets);
        else
            hipLaunchKernelGGL((msm_accumulate_kernel<C, 1>), dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, st,
                               (const Aff<C>*)h->d_points, (const uint32_t*)sorted, (const uint32_t*)starts,
                               (const uint32_t*)counts, (const uint32_t*)order, n_heavy, (uint32_t)total,
                               (const Aff<C>*)salts, buckets);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(g.ev[3], st));
    TRACE("accumulate done")
    if (n_heavy > 0) {
        if ((rc = pool_get("partials", (size_t)n_chunks * sizeof(Proj<C>), (void**)&partials))) return rc;
        hipLaunchKernelGGL((msm_heavy_chunk_kernel<C>), dim3(n_chunks), dim3(64), lds_wave, st, (const Aff<C>*)h->d_points,
                           (const uint32_t*)sorted, (const uint32_t*)starts, (const uint32_t*)counts, (const uint32_t*)order,
                           (const uint32_t*)chunk_start, n_heavy, partials);
        hipLaunchKernelGGL((msm_heavy_combine_kernel<C>), dim3(n_heavy), dim3(64), lds_wave, st, (const Proj<C>*)partials,
                           (const uint32_t*)order, (const uint32_t*)chunk_start, buckets);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(g.ev[4], st));
    TRACE("heavy done")
    hipLaunchKernelGGL((msm_reduce1_kernel<C>), dim3((unsigned)(W * segs_per_window)), dim3(64), lds_wave, st,
                       (const Proj<C>*)buckets, nb, nbp, seg_run, seg_wacc);
    TRACE("reduce1 done")
    int log_u = 0;
    while ((1u << log_u) < seg_slots) log_u++;
    hipLaunchKernelGGL((msm_reduce2_kernel<C>), dim3((unsigned)W), dim3(64), lds_wave, st, (const Proj<C>*)seg_run,
                       (const Proj<C>*)seg_wacc, segs_per_window, log_u, wsums);
    HIPCHK(hipGetLastError());
    TRACE("reduce2 done")
    HIPCHK(hipEventRecord(g.ev[5], st));
    std::vector<Proj<C>> hw(W);
    HIPCHK(hipMemcpyAsync(hw.data(), wsums, (size_t)W * sizeof(Proj<C>), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    auto t_fold0 = std::chrono::steady_clock::now();
    // window fold, high to low (variable_base.rs:73-82)
    Proj<C> acc = hw[W - 1];
    for (int w = W - 2; w >= 0; w--) {
        for (int d = 0; d < c; d++) acc = proj_dbl<C>(acc);
        acc = proj_add<C>(acc, hw[w]);
    }
    if (proj_is_zero<C>(acc)) acc = proj_zero<C>();   // canonical (0, 1, 0) like the reference's zero()
    proj_to_abi_host<C>(out_xyz, acc);
    TRACE("fold done")
#undef TRACE
    auto t_end = std::chrono::steady_clock::now();
    HIPCHK(hipEventElapsedTime(&tm.sort_ms, g.ev[0], g.ev[1]));
    HIPCHK(hipEventElapsedTime(&tm.accumulate_ms, g.ev[2], g.ev[3]));   // brackets exactly msm_accumulate_kernel
    HIPCHK(hipEventElapsedTime(&tm.heavy_ms, g.ev[3], g.ev[4]));
    HIPCHK(hipEventElapsedTime(&tm.reduce_ms, g.ev[4], g.ev[5]));
    tm.heavy_buckets = n_heavy;
    tm.fold_ms = std::chrono::duration<float, std::milli>(t_end - t_fold0).count();
    tm.total_ms = std::chrono::duration<float, std::milli>(t_end - t_begin).count();
    tm.window_bits = c;
    tm.num_windows = W;
    uint32_t last_start = 0, last_count = 0;
    HIPCHK(hipMemcpy(&last_start, starts + total - 1, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&last_count, counts + total - 1, 4, hipMemcpyDeviceToHost));
    tm.accumulate_madds = (unsigned long long)last_start + last_count;
    g.last_msm = tm;
    return GH_OK;
}

template <class C>
int msm_host(const uint64_t* bases, const uint8_t* infinity, size_t n_bases, const uint64_t* scalars, size_t n_scalars,
             uint64_t* out_xyz) {
    size_t n = n_bases < n_scalars ? n_bases : n_scalars;
    BasesBase* h = nullptr;
    int rc = upload_bases<C>(bases, infinity, n, &h);
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
                                   &proj_add_host<CURVE>, &to_affine_host<CURVE>};             \
        return &ops;                                                                           \
    }                                                                                          \
    }

}  // namespace gh_rt

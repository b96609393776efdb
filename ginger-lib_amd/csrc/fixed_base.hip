// fixed_base.hip -- FixedBaseMSM (algebra/src/msm/fixed_base.rs:7-79): v[i] * g for a vector of scalars and ONE base,
// through a window table of g.  Used by the Groth16 parameter generator (proof-systems/src/groth16/generator.rs:225-296:
// a_query, b_g1_query, b_g2_query, h_query, l_query, gamma_abc_g1 are each one such call).
//
// Reference: table[outer][inner] = inner * 2^(window outer) * g, outerc = ceil(scalar_size / window) rows of 2^window
// projective points; windowed_mul adds one table entry per row (:45-66).  Here the table is kept in AFFINE form in HBM
// (entry 0 = infinity is never read), one thread per scalar walks its outerc digits with mixed additions
// (add_assign_mixed, short_weierstrass_projective.rs:481-519, complete: doubling and infinity handled), and the result
// is the same group element as the reference's (a projective representative; into_affine() is canonical).
// The sums run on the MSM's own accumulation kernels (round 3): scalar i is "bucket" i and its list the table entries its
// digits select (fixed_digits_kernel), so the per-row additions are the inlined XYZZ updates (G1) / lane-group updates (G2) of
// msm_kernels.h instead of one out-of-line add_assign_mixed per row and thread -- 2^20 G1 scalars at window 14: 75 -> 31 ms.
// GH_FIXED_NAIVE=1 (or a table with infinity entries: g of small order) keeps the one-thread-per-scalar kernel below.
// Table building, normalisation and the chain generator use the out-of-line curve functions of ec29.h (small code, any curve).
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "runtime.h"
#include "msm_kernels.h"
#include "host_math.h"

namespace gh {

// table entry (outer, inner), inner >= 1: inner * g_outer by double-and-add from the top bit, then one inversion
template <class C>
__global__ void __launch_bounds__(64)
fixed_table_kernel(const Proj<C>* __restrict__ g_outer /* outerc points: 2^(window outer) g */, int window, uint32_t outerc,
                   uint32_t last_in_window, Aff<C>* __restrict__ table, uint32_t* __restrict__ mark_flag) {
    typedef typename C::FC F;
    const uint32_t in_window = 1u << window;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)outerc * in_window) return;
    const uint32_t outer = (uint32_t)(t >> window), inner = (uint32_t)(t & (in_window - 1));
    const uint32_t cur = outer == outerc - 1 ? last_in_window : in_window;
    Aff<C> out{F::zero(), F::zero()};
    if (inner >= 1 && inner < cur) {
        const Proj<C> g = ld_words(g_outer + outer);
        Proj<C> acc = proj_zero<C>();
        for (int b = window - 1; b >= 0; b--) {
            acc = proj_dbl_call<C>(acc);
            if ((inner >> b) & 1u) acc = proj_add_call<C>(acc, g);
        }
        if (!F::is_zero(acc.z)) {
            const typename F::T zi = DevInv<F>::inv(acc.z);
            out.x = F::mul(acc.x, zi);
            out.y = F::mul(acc.y, zi);
        } else {
            F::comp(out.x, 0).l[0] = AFF_MARK;       // inner * g_outer = infinity (g of small order): marked, skipped below
            *mark_flag = 1u;
        }
    }
    st_words(table + t, out);
}

// one thread per scalar: res = sum over rows of table[outer][digit]   (windowed_mul :45-66; digit 0 adds table[..][0] = 0)
template <class C>
__global__ void __launch_bounds__(64)
fixed_msm_kernel(const Aff<C>* __restrict__ table, int window, uint32_t outerc, uint32_t scalar_size,
                 const uint32_t* __restrict__ scalars /* n x 24 words, canonical */, size_t n, Proj<C>* __restrict__ out) {
    typedef typename C::FC F;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[25];
#pragma unroll
    for (int k = 0; k < 24; k++) s[k] = scalars[i * 24 + k];
    s[24] = 0;
    Proj<C> acc = proj_zero<C>();
    for (uint32_t outer = 0; outer < outerc; outer++) {
        const uint32_t bit = outer * (uint32_t)window;
        // windowed_mul (:45-66) reads `window` bits per row, but none at or above MODULUS_BITS = 753 (both scalar fields), and a
        // digit of the last row at or beyond last_in_window = 2^(scalar_size - (outerc - 1) window) finds the zero the table was
        // initialised with (:22-33) -- which only happens for scalar_size < 753 with a scalar that has higher bits set
        uint32_t nb = (uint32_t)window;
        if (bit >= 753u) continue;
        if (bit + nb > 753u) nb = 753u - bit;
        const uint32_t wi = bit >> 5, sh = bit & 31;
        const uint64_t two = (uint64_t)s[wi] | ((uint64_t)s[wi + 1] << 32);
        const uint32_t d = (uint32_t)(two >> sh) & ((1u << nb) - 1u);
        if (d == 0) continue;
        if (outer == outerc - 1 && d >= (1u << (scalar_size - (outerc - 1) * (uint32_t)window))) continue;
        const Aff<C> q = ld_words(table + ((size_t)outer << window) + d);
        if (F::comp(q.x, 0).l[0] == AFF_MARK) continue;
        acc = proj_madd_call<C>(acc, q);
    }
    st_words(out + i, acc);
}

// The same digits as bucket lists for the MSM's accumulation kernels (msm_impl.h: accumulate_lists): list i = the table entries
// scalar i selects, at sorted[i * outerc ..], counts[i] of them; order = identity.  The digit rules are fixed_msm_kernel's.
static __global__ void __launch_bounds__(256)
fixed_digits_kernel(int window, uint32_t outerc, uint32_t scalar_size, const uint32_t* __restrict__ scalars, size_t n,
                    uint32_t* __restrict__ sorted, uint32_t* __restrict__ starts, uint32_t* __restrict__ counts, uint32_t* __restrict__ order) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[25];
#pragma unroll
    for (int k = 0; k < 24; k++) s[k] = scalars[i * 24 + k];
    s[24] = 0;
    uint32_t* list = sorted + i * outerc;
    uint32_t cnt = 0;
    for (uint32_t outer = 0; outer < outerc; outer++) {
        const uint32_t bit = outer * (uint32_t)window;
        uint32_t nb = (uint32_t)window;
        if (bit >= 753u) break;
        if (bit + nb > 753u) nb = 753u - bit;
        const uint32_t wi = bit >> 5, sh = bit & 31;
        const uint64_t two = (uint64_t)s[wi] | ((uint64_t)s[wi + 1] << 32);
        const uint32_t d = (uint32_t)(two >> sh) & ((1u << nb) - 1u);
        if (d == 0) continue;
        if (outer == outerc - 1 && d >= (1u << (scalar_size - (outerc - 1) * (uint32_t)window))) continue;
        list[cnt++] = (outer << window) + d;
    }
    starts[i] = (uint32_t)(i * outerc);
    counts[i] = cnt;
    order[i] = (uint32_t)i;
}

// batch_normalization + into_affine of a vector of projective points (short_weierstrass_projective.rs:402-442, :663-678; the
// parameter generator does exactly this to every query: generator.rs:318-335), written in the C ABI's affine layout:
// n x (x || y) coefficients, either Montgomery 2^768 (the in-memory form) or -- canonical != 0 -- plain integers (what
// GroupAffine::write serialises, :185-192), plus the infinity flags; infinity is GroupAffine::zero() = (0, 1, true).
// One thread per run of NORM_RUN points: prefix products of the non-zero Z, ONE inversion, backward sweep (Montgomery's trick).
constexpr int NORM_RUN = 32;
template <class C>
__global__ void __launch_bounds__(64)
fixed_normalize_kernel(const Proj<C>* __restrict__ in, size_t n, int canonical, uint32_t* __restrict__ out_xy, uint8_t* __restrict__ out_inf,
                       typename C::FC::T* __restrict__ zp) {
    typedef typename C::FC F;
    typedef typename C::PF PF;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i0 = t * NORM_RUN;
    if (i0 >= n) return;
    const int cnt = (int)(n - i0 < (size_t)NORM_RUN ? n - i0 : (size_t)NORM_RUN);
    typename F::T run = F::one();
    for (int j = 0; j < cnt; j++) {
        const typename F::T z = ld_words(&in[i0 + j].z);
        st_words(zp + i0 + j, run);                      // product of the non-zero Z before j
        if (!F::is_zero(z)) run = F::mul(run, z);
    }
    typename F::T inv = DevInv<F>::inv(run);
    Fp plain_one = fp_zero();
    plain_one.l[0] = 1;
    auto put = [&](uint32_t* w, const typename F::T& v) {
        for (int d = 0; d < F::DEG; d++) {
            const Fp& c = F::comp(v, d);
            if (canonical) fp_pack(w + 24 * d, fp_mul_call<PF>(c, plain_one));   // internal Montgomery (x 2^754) -> the integer itself
            else fp_to_abi<PF>(w + 24 * d, c);
        }
    };
    for (int j = cnt - 1; j >= 0; j--) {
        const Proj<C> p = ld_words(in + i0 + j);
        uint32_t* w = out_xy + (i0 + j) * (size_t)(48 * F::DEG);
        if (F::is_zero(p.z)) {
            put(w, F::zero());
            put(w + 24 * F::DEG, F::one());
            out_inf[i0 + j] = 1;
            continue;
        }
        const typename F::T zi = F::mul(inv, ld_words(zp + i0 + j));      // 1 / Z_j
        inv = F::mul(inv, p.z);
        put(w, F::mul(p.x, zi));
        put(w + 24 * F::DEG, F::mul(p.y, zi));
        out_inf[i0 + j] = 0;
    }
}

// Synthetic key for benchmarks and full-size tests: n DISTINCT points P_i = P_0 + i H along an addition
// chain, written straight into the resident internal layout.  One thread per run of CHAIN_RUN consecutive points:
// start P_0 + (t RUN) H by double-and-add, RUN - 1 mixed additions of H, one inversion per thread over the run's Z
// (Montgomery's trick, as batch_normalization: short_weierstrass_projective.rs:402-442).  zs: n field elements of scratch.
constexpr int CHAIN_RUN = 32;
template <class C>
__global__ void __launch_bounds__(64)
chain_bases_kernel(Aff<C> p0, Aff<C> h, size_t n, Aff<C>* __restrict__ out, typename C::FC::T* __restrict__ zs) {
    typedef typename C::FC F;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i0 = t * CHAIN_RUN;
    if (i0 >= n) return;
    const int cnt = (int)(n - i0 < (size_t)CHAIN_RUN ? n - i0 : (size_t)CHAIN_RUN);
    // (t RUN) H, top bit first
    Proj<C> acc = proj_zero<C>();
    const Proj<C> hp{h.x, h.y, F::one()};
    for (int b = 40; b >= 0; b--) {
        acc = proj_dbl_call<C>(acc);
        if ((i0 >> b) & 1) acc = proj_add_call<C>(acc, hp);
    }
    acc = proj_madd_call<C>(acc, p0);
    typename F::T run = F::one();
    for (int j = 0; j < cnt; j++) {
        // (a chain that meets infinity -- P_0 a small multiple of -H -- is not a benchmark key; its rows would be (0, 0))
        st_words(out + i0 + j, Aff<C>{acc.x, acc.y});
        st_words(zs + i0 + j, acc.z);
        run = F::mul(run, acc.z);
        acc = proj_madd_call<C>(acc, h);
    }
    typename F::T inv = DevInv<F>::inv(run);
    for (int j = cnt - 1; j >= 0; j--) {
        typename F::T zi = inv;                                  // 1 / (Z_0 .. Z_j) -> 1 / Z_j needs the product of the earlier ones
        typename F::T pre = F::one();
        for (int k = 0; k < j; k++) pre = F::mul(pre, ld_words(zs + i0 + k));     // RUN is small: recompute instead of a second scratch array
        zi = F::mul(inv, pre);
        inv = F::mul(inv, ld_words(zs + i0 + j));
        Aff<C> q = ld_words(out + i0 + j);
        q.x = F::mul(q.x, zi);
        q.y = F::mul(q.y, zi);
        st_words(out + i0 + j, q);
    }
}

}  // namespace gh

namespace gh_rt {
using namespace gh;

template <class C> struct CurveIdOf;
template <> struct CurveIdOf<Mnt4G1> { static constexpr gh_curve_t id = GH_MNT4753_G1; };
template <> struct CurveIdOf<Mnt4G2> { static constexpr gh_curve_t id = GH_MNT4753_G2; };
template <> struct CurveIdOf<Mnt6G1> { static constexpr gh_curve_t id = GH_MNT6753_G1; };
template <> struct CurveIdOf<Mnt6G2> { static constexpr gh_curve_t id = GH_MNT6753_G2; };

struct FixedTable {
    uint32_t magic = 0x67684654u;
    gh_curve_t curve;
    int window = 0;
    uint32_t outerc = 0, scalar_size = 0;
    void* d_table = nullptr;
    bool has_marks = false;     // some entry is the point at infinity (g = 0 or of small order): only fixed_msm_kernel skips those
};

template <class C> int build_table(const uint64_t* g_xyz, size_t scalar_size, int window, FixedTable* t) {
    typedef typename HostCurveOf<C>::type HC;
    const uint32_t outerc = (uint32_t)((scalar_size + window - 1) / window);
    const uint32_t last_in_window = 1u << (scalar_size - (size_t)(outerc - 1) * window);
    // g_outer = 2^(window outer) g on the host (outerc * window <= ~770 dependent doublings)
    std::vector<Proj<C>> gouter(outerc);
    {
        Proj<HC> p;
        memcpy(&p, g_xyz, sizeof(p));
        for (uint32_t o = 0; o < outerc; o++) {
            uint64_t abi[108];
            memcpy(abi, &p, sizeof(p));
            const uint32_t* w = reinterpret_cast<const uint32_t*>(abi);
            typedef typename C::F F;
            gouter[o].x = F::from_abi(w);
            gouter[o].y = F::from_abi(w + 24 * F::DEG);
            gouter[o].z = F::from_abi(w + 48 * F::DEG);
            for (int k = 0; k < window; k++) p = proj_dbl<HC>(p);
        }
    }
    const size_t entries = (size_t)outerc << window;
    Proj<C>* d_g = nullptr;
    uint32_t* d_flag = nullptr;
    uint32_t h_flag = 0;
    if (int rc = pool_get("fixed_flag", 4, (void**)&d_flag)) return rc;
    HIPCHK(hipMalloc(&t->d_table, entries * sizeof(Aff<C>)));
    hipError_t e = hipMalloc((void**)&d_g, outerc * sizeof(Proj<C>));
    if (e == hipSuccess) e = hipMemcpyAsync(d_g, gouter.data(), outerc * sizeof(Proj<C>), hipMemcpyHostToDevice, g.stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, 4, g.stream);
    if (e == hipSuccess) {
        GH_LAUNCH((fixed_table_kernel<C>), dim3((unsigned)((entries + 63) / 64)), dim3(64), 0, g.stream, (const Proj<C>*)d_g,
                           window, outerc, last_in_window, (Aff<C>*)t->d_table, d_flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_flag, d_flag, 4, hipMemcpyDeviceToHost, g.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    t->has_marks = h_flag != 0;
    if (d_g) (void)hipFree(d_g);
    if (e != hipSuccess) {
        (void)hipFree(t->d_table);
        t->d_table = nullptr;
        g_err = std::string("fixed-base table: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? GH_E_NOMEM : GH_E_HIP;
    }
    t->window = window; t->outerc = outerc; t->scalar_size = (uint32_t)scalar_size;
    return GH_OK;
}

template <class C> const MsmOps* fixed_ops();
template <> const MsmOps* fixed_ops<Mnt4G1>() { return msm_ops_mnt4753_g1(); }
template <> const MsmOps* fixed_ops<Mnt4G2>() { return msm_ops_mnt4753_g2(); }
template <> const MsmOps* fixed_ops<Mnt6G1>() { return msm_ops_mnt6753_g1(); }
template <> const MsmOps* fixed_ops<Mnt6G2>() { return msm_ops_mnt6753_g2(); }

// d_out[i] = sum over rows of table[outer][digit_outer(scalar i)] on g.stream (scalars already on the device)
template <class C> int launch_fixed_sums(const FixedTable* t, const void* d_s, size_t n, void* d_o) {
    static const bool naive = getenv("GH_FIXED_NAIVE") && atoi(getenv("GH_FIXED_NAIVE")) != 0;
    if (naive || t->has_marks || n * (size_t)t->outerc >= ((size_t)1 << 31)) {
        GH_LAUNCH((fixed_msm_kernel<C>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, g.stream, (const Aff<C>*)t->d_table,
                           t->window, t->outerc, t->scalar_size, (const uint32_t*)d_s, n, (Proj<C>*)d_o);
        HIPCHK(hipGetLastError());
        return GH_OK;
    }
    uint32_t *d_list = nullptr, *d_meta = nullptr;
    int rc = pool_get("fixed_list", n * (size_t)t->outerc * 4, (void**)&d_list);
    if (!rc) rc = pool_get("fixed_meta", n * 12, (void**)&d_meta);
    if (rc) return rc;
    uint32_t *d_starts = d_meta, *d_counts = d_meta + n, *d_order = d_meta + 2 * n;
    GH_LAUNCH(fixed_digits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g.stream, t->window, t->outerc, t->scalar_size,
                       (const uint32_t*)d_s, n, d_list, d_starts, d_counts, d_order);
    HIPCHK(hipGetLastError());
    return fixed_ops<C>()->acc_lists(t->d_table, d_list, d_starts, d_counts, d_order, (uint32_t)n, d_o, g.stream);
}

template <class C> int run_fixed(const FixedTable* t, const uint64_t* scalars, size_t n, uint64_t* out_xyz) {
    typedef typename C::F F;
    if (n == 0) return GH_OK;
    void *d_s = nullptr, *d_o = nullptr;
    int rc = pool_get("fixed_scalars", n * 96, &d_s);
    if (!rc) rc = pool_get("fixed_out", n * sizeof(Proj<C>), &d_o);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(d_s, scalars, n * 96, hipMemcpyHostToDevice, g.stream));
    if ((rc = launch_fixed_sums<C>(t, d_s, n, d_o))) return rc;
    std::vector<Proj<C>> host(n);
    HIPCHK(hipMemcpyAsync(host.data(), d_o, n * sizeof(Proj<C>), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    for (size_t i = 0; i < n; i++) {                       // internal -> ABI Montgomery limbs
        uint32_t* w = reinterpret_cast<uint32_t*>(out_xyz + i * 36 * F::DEG);
        Proj<C> p = host[i];
        if (proj_is_zero<C>(p)) p = proj_zero<C>();          // canonical (0, 1, 0) like the reference's zero()
        F::to_abi(w, p.x);
        F::to_abi(w + 24 * F::DEG, p.y);
        F::to_abi(w + 48 * F::DEG, p.z);
    }
    return GH_OK;
}

// FixedBaseMSM::multi_scalar_mul + batch_normalization + into_affine, all on the device (generator.rs:247-335 per query)
template <class C> int run_fixed_affine(const FixedTable* t, const uint64_t* scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf, int canonical) {
    typedef typename C::F F;
    typedef typename C::FC::T FT;
    if (n == 0) return GH_OK;
    void *d_s = nullptr, *d_o = nullptr, *d_xy = nullptr, *d_inf = nullptr, *d_zp = nullptr;
    int rc = pool_get("fixed_scalars", n * 96, &d_s);
    if (!rc) rc = pool_get("fixed_out", n * sizeof(Proj<C>), &d_o);
    if (!rc) rc = pool_get("fixed_xy", n * (size_t)(192 * F::DEG), &d_xy);
    if (!rc) rc = pool_get("fixed_inf", n, &d_inf);
    if (!rc) rc = pool_get("fixed_zp", n * sizeof(FT), &d_zp);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(d_s, scalars, n * 96, hipMemcpyHostToDevice, g.stream));
    if ((rc = launch_fixed_sums<C>(t, d_s, n, d_o))) return rc;
    const size_t threads = (n + NORM_RUN - 1) / NORM_RUN;
    GH_LAUNCH((fixed_normalize_kernel<C>), dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, g.stream, (const Proj<C>*)d_o, n, canonical,
                       (uint32_t*)d_xy, (uint8_t*)d_inf, (FT*)d_zp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_xy, d_xy, n * (size_t)(192 * F::DEG), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipMemcpyAsync(out_inf, d_inf, n, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return GH_OK;
}

template <class C> int chain_bases(const uint64_t* p0_xy, const uint64_t* h_xy, size_t n, BasesBase** out) {
    typedef typename C::F F;
    typedef typename C::FC::T FT;
    BasesBase* hb = new BasesBase();
    hb->curve = CurveIdOf<C>::id;
    hb->n = n;
    if (n > 0) {
        Aff<C> p0, h;
        const uint32_t* w = reinterpret_cast<const uint32_t*>(p0_xy);
        p0.x = F::from_abi(w); p0.y = F::from_abi(w + 24 * F::DEG);
        w = reinterpret_cast<const uint32_t*>(h_xy);
        h.x = F::from_abi(w); h.y = F::from_abi(w + 24 * F::DEG);
        FT* zs = nullptr;
        hipError_t e = hipMalloc(&hb->d_points, n * sizeof(Aff<C>));
        if (e == hipSuccess) e = hipMalloc((void**)&zs, n * sizeof(FT));
        if (e == hipSuccess) {
            const size_t threads = (n + CHAIN_RUN - 1) / CHAIN_RUN;
            GH_LAUNCH((chain_bases_kernel<C>), dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, g.stream, p0, h, n,
                               (Aff<C>*)hb->d_points, zs);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
        if (zs) (void)hipFree(zs);
        if (e != hipSuccess) {
            if (hb->d_points) (void)hipFree(hb->d_points);
            delete hb;
            g_err = std::string("chain bases: ") + hipGetErrorString(e);
            return e == hipErrorOutOfMemory ? GH_E_NOMEM : GH_E_HIP;
        }
    }
    *out = hb;
    return GH_OK;
}

template <class C> int download_bases(const BasesBase* hb, size_t first, size_t count, uint64_t* out_xy) {
    typedef typename C::F F;
    std::vector<Aff<C>> host(count);
    HIPCHK(hipMemcpyAsync(host.data(), (const Aff<C>*)hb->d_points + first, count * sizeof(Aff<C>), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    for (size_t i = 0; i < count; i++) {
        uint32_t* w = reinterpret_cast<uint32_t*>(out_xy + i * 24 * F::DEG);
        F::to_abi(w, host[i].x);
        F::to_abi(w + 24 * F::DEG, host[i].y);
    }
    return GH_OK;
}

int fixed_build(gh_curve_t curve, const uint64_t* g_xyz, size_t scalar_size, int window, FixedTable* t) {
    switch (curve) {
        case GH_MNT4753_G1: return build_table<Mnt4G1>(g_xyz, scalar_size, window, t);
        case GH_MNT4753_G2: return build_table<Mnt4G2>(g_xyz, scalar_size, window, t);
        case GH_MNT6753_G1: return build_table<Mnt6G1>(g_xyz, scalar_size, window, t);
        case GH_MNT6753_G2: return build_table<Mnt6G2>(g_xyz, scalar_size, window, t);
    }
    g_err = "unknown curve id";
    return GH_E_BAD_ARG;
}
int fixed_run(const FixedTable* t, const uint64_t* scalars, size_t n, uint64_t* out_xyz) {
    switch (t->curve) {
        case GH_MNT4753_G1: return run_fixed<Mnt4G1>(t, scalars, n, out_xyz);
        case GH_MNT4753_G2: return run_fixed<Mnt4G2>(t, scalars, n, out_xyz);
        case GH_MNT6753_G1: return run_fixed<Mnt6G1>(t, scalars, n, out_xyz);
        case GH_MNT6753_G2: return run_fixed<Mnt6G2>(t, scalars, n, out_xyz);
    }
    return GH_E_BAD_ARG;
}

int fixed_run_affine(const FixedTable* t, const uint64_t* scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf, int canonical) {
    switch (t->curve) {
        case GH_MNT4753_G1: return run_fixed_affine<Mnt4G1>(t, scalars, n, out_xy, out_inf, canonical);
        case GH_MNT4753_G2: return run_fixed_affine<Mnt4G2>(t, scalars, n, out_xy, out_inf, canonical);
        case GH_MNT6753_G1: return run_fixed_affine<Mnt6G1>(t, scalars, n, out_xy, out_inf, canonical);
        case GH_MNT6753_G2: return run_fixed_affine<Mnt6G2>(t, scalars, n, out_xy, out_inf, canonical);
    }
    return GH_E_BAD_ARG;
}

}  // namespace gh_rt

using namespace gh_rt;

extern "C" {

int gh_fixed_base_window(size_t num_scalars) {     // FixedBaseMSM::get_mul_window_size (fixed_base.rs:7-13)
    if (num_scalars < 32) return 3;
    return (int)ceil(log((double)(uint32_t)num_scalars));
}

int gh_fixed_base_table(gh_curve_t curve, const uint64_t* g_xyz, size_t scalar_size, int window, gh_fixed_table_t* out) try {
    std::lock_guard<std::mutex> lk(api_mutex());
    if (!g_xyz || !out) { g_err = "null argument"; return GH_E_BAD_ARG; }
    if (window < 1 || window > 22 || scalar_size < 1 || scalar_size > 768) { g_err = "fixed-base window must be in [1, 22], scalar_size in [1, 768]"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    FixedTable* t = new FixedTable();
    t->curve = curve;
    rc = fixed_build(curve, g_xyz, scalar_size, window, t);
    if (rc) { delete t; return rc; }
    *out = reinterpret_cast<gh_fixed_table_t>(t);
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_fixed_base_msm(gh_fixed_table_t table, const uint64_t* scalars, size_t n, uint64_t* out_xyz) try {
    std::lock_guard<std::mutex> lk(api_mutex());
    FixedTable* t = reinterpret_cast<FixedTable*>(table);
    if (!t || t->magic != 0x67684654u) { g_err = "bad fixed-base table handle"; return GH_E_BAD_HANDLE; }
    if (n && (!scalars || !out_xyz)) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    return fixed_run(t, scalars, n, out_xyz);
} catch (...) { return gh_rt::api_exception(); }

int gh_fixed_base_msm_affine(gh_fixed_table_t table, const uint64_t* scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf, int canonical) try {
    std::lock_guard<std::mutex> lk(api_mutex());
    FixedTable* t = reinterpret_cast<FixedTable*>(table);
    if (!t || t->magic != 0x67684654u) { g_err = "bad fixed-base table handle"; return GH_E_BAD_HANDLE; }
    if (n && (!scalars || !out_xy || !out_inf)) { g_err = "null argument"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    return fixed_run_affine(t, scalars, n, out_xy, out_inf, canonical);
} catch (...) { return gh_rt::api_exception(); }

int gh_bases_generate_chain(gh_curve_t curve, const uint64_t* p0_xy, const uint64_t* step_xy, size_t n, gh_bases_t* out_handle) try {
    std::lock_guard<std::mutex> lk(api_mutex());
    if (!p0_xy || !step_xy || !out_handle) { g_err = "null argument"; return GH_E_BAD_ARG; }
    if (n >= ((size_t)1 << 40)) { g_err = "chain too long"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    BasesBase* h = nullptr;
    switch (curve) {
        case GH_MNT4753_G1: rc = chain_bases<Mnt4G1>(p0_xy, step_xy, n, &h); break;
        case GH_MNT4753_G2: rc = chain_bases<Mnt4G2>(p0_xy, step_xy, n, &h); break;
        case GH_MNT6753_G1: rc = chain_bases<Mnt6G1>(p0_xy, step_xy, n, &h); break;
        case GH_MNT6753_G2: rc = chain_bases<Mnt6G2>(p0_xy, step_xy, n, &h); break;
        default: g_err = "unknown curve id"; return GH_E_BAD_ARG;
    }
    if (rc) return rc;
    *out_handle = reinterpret_cast<gh_bases_t>(h);
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

int gh_bases_download(gh_bases_t handle, size_t first, size_t count, uint64_t* out_xy) try {
    std::lock_guard<std::mutex> lk(api_mutex());
    BasesBase* h = reinterpret_cast<BasesBase*>(handle);
    if (!h || h->magic != 0x6768424au) { g_err = "bad bases handle"; return GH_E_BAD_HANDLE; }
    if (count == 0) return GH_OK;
    if (!out_xy || first > h->n || count > h->n - first) { g_err = "range outside the resident bases"; return GH_E_BAD_ARG; }
    int rc = ensure_init();
    if (rc) return rc;
    switch (h->curve) {
        case GH_MNT4753_G1: return download_bases<Mnt4G1>(h, first, count, out_xy);
        case GH_MNT4753_G2: return download_bases<Mnt4G2>(h, first, count, out_xy);
        case GH_MNT6753_G1: return download_bases<Mnt6G1>(h, first, count, out_xy);
        case GH_MNT6753_G2: return download_bases<Mnt6G2>(h, first, count, out_xy);
    }
    return GH_E_BAD_ARG;
} catch (...) { return gh_rt::api_exception(); }

int gh_fixed_base_free(gh_fixed_table_t table) try {
    std::lock_guard<std::mutex> lk(api_mutex());
    FixedTable* t = reinterpret_cast<FixedTable*>(table);
    if (!t || t->magic != 0x67684654u) { g_err = "bad fixed-base table handle"; return GH_E_BAD_HANDLE; }
    if (t->d_table) (void)hipFree(t->d_table);
    t->magic = 0;
    delete t;
    return GH_OK;
} catch (...) { return gh_rt::api_exception(); }

}  // extern "C"

// aff_kernels.h -- bucket sums in AFFINE coordinates by pairwise rounds over a flat, bucket-ordered list.
//
// What it replaces: the inner loop of msm_inner (algebra/src/msm/variable_base.rs:36-59), n x W calls of
// add_assign_mixed (short_weierstrass_projective.rs:481-519, 11 Fp-mul each).  A bucket sum is a sum of affine
// points in ANY order, and an affine addition costs 1 inversion + 2 M + 1 S; Montgomery's trick (the one
// batch_normalization uses, swp.rs:402-442) shares one inversion over a batch at 3 M per member, so an addition
// comes to 5 M + 1 S + (one safegcd inversion, ~40 M) / batch -- against 11 M projectively.
//
// Round r turns the list P_r (all buckets back to back, bucket b at [st_r[b], st_r[b] + m_r[b])) into P_(r+1) with
// m_(r+1)[b] = ceil(m_r[b] / 2): output j of bucket b is P_r[st_r[b] + 2j] + P_r[st_r[b] + 2j + 1], an odd
// leftover is copied.  The unit of work is the OUTPUT ELEMENT, not the bucket:
//   * a descriptor desc_r[o] = (index of the first input | pair flag << 31) per output o (aff_desc_kernel);
//   * the round kernel gives every lane group B output elements of its wave's contiguous chunk, interleaved
//     (o = chunk + k * TPW + g): identical trip counts in all lanes, no bucket cursor, no divergence, and the 64
//     lanes of a wave read one contiguous span of the list per iteration;
//   * forward pass: running product of the denominators x2 - x1 (parked in `prefix`), ONE inversion per lane,
//     backward pass: 1 / (x2 - x1), lambda, the sum.
// Every case of the group law is handled in place (no fallback list):
//   x1 == x2, y1 == y2 != 0  -> doubling: lambda = (3 x1^2 + a) / (2 y1)      (the reference's P == Q branch, :492)
//   x1 == x2 otherwise       -> the point at infinity, stored as a marker (x.l[0] = 0xFFFFFFFF, not a limb value)
//   marker + Q -> Q
// Both rare paths sit behind a wave-uniform `any` so the common case pays nothing for them.
// After the last round a bucket holds a few points at most; msm_accumulate_kernel<.., AFFIN = true> adds them
// projectively (identity list, markers skipped) and leaves the buckets in the form the reduction expects.
#pragma once
#include "ec29.h"

namespace gh {

constexpr uint32_t AFF_MARK = 0xFFFFFFFFu;   // x.l[0] of the infinity marker
constexpr int AFF_MAX_ROUNDS = 26;

#ifndef GH_LD_ST_FP
#define GH_LD_ST_FP
GH_HD Fp ld_fp(const Fp* p) {
    Fp r;
    const uint2* q = reinterpret_cast<const uint2*>(p);
    GH_UNROLL for (int i = 0; i < NL / 2; i++) { uint2 v = q[i]; r.l[2 * i] = v.x; r.l[2 * i + 1] = v.y; }
    return r;
}
GH_HD void st_fp(Fp* p, const Fp& a) {
    uint2* q = reinterpret_cast<uint2*>(p);
    GH_UNROLL for (int i = 0; i < NL / 2; i++) q[i] = make_uint2(a.l[2 * i], a.l[2 * i + 1]);
}
#endif

// ---- lane-group field policies: an element of the coordinate field lives in LANES adjacent lanes, one Fp
//      coefficient per lane (G1: one lane; Fq2: lane pairs; Fq3: lane triples -- F2S / F3S of msm_kernels.h).
template <class P> struct F1S {
    typedef Fp T;
    static constexpr int LANES = 1;
    static constexpr int WAVES = 2;
    GH_HD static T one() { return fp_one<P>(); }
    GH_HD static T zero() { return fp_zero(); }
    GH_HD static T add(const T& a, const T& b) { return fp_add<P>(a, b); }
    GH_HD static T sub(const T& a, const T& b) { return fp_sub<P>(a, b); }
    GH_HD static T dbl(const T& a) { return fp_dbl<P>(a); }
    GH_HD static T neg(const T& a) { return fp_neg<P>(a); }
    GH_HD static T mul(const T& a, const T& b) { return fp_mul<P>(a, b); }
    GH_HD static T sqr(const T& a) { return fp_sqr<P>(a); }
    GH_HD static bool is_zero(const T& a) { return fp_is_zero(a); }
    GH_HD static bool eq(const T& a, const T& b) { return fp_eq(a, b); }
    GH_HD static T inv(const T& a) { return fp_inv<P>(a); }
};

// wave-uniform "does any lane want the rare path" (host emulation: lanes run one at a time)
GH_HD bool aff_any(bool f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __any(f ? 1 : 0) != 0;
#else
    return f;
#endif
}

// the curve coefficient a as an element of the lane-group field (coefficient `comp` of it)
template <class C, class FS> struct CurveA;
template <class FS> struct CurveA<Mnt4G1, FS> { GH_HD static Fp get(int) { return fp_dbl<P4>(fp_one<P4>()); } };            // a = 2
template <class FS> struct CurveA<Mnt6G1, FS> { GH_HD static Fp get(int) { return fp_mul_small<P6, 11>(fp_one<P6>()); } };  // a = 11
template <class FS> struct CurveA<Mnt4G2, FS> {   // a' = (26, 0)          (curves/mnt4753/g2.rs:57-75)
    GH_HD static Fp get(int comp) { return comp == 0 ? fp_mul_small<P4, 26>(fp_one<P4>()) : fp_zero(); }
};
template <class FS> struct CurveA<Mnt6G2, FS> {   // a' = (0, 0, 11)       (curves/mnt6753/g2.rs:71-100)
    GH_HD static Fp get(int comp) { return comp == 2 ? fp_mul_small<P6, 11>(fp_one<P6>()) : fp_zero(); }
};

template <class C> struct AffRoundArgs {
    const Aff<C>* in;          // round 0: the bases / the shift table; later: the previous round's output
    const uint32_t* sorted;    // round 0: list entries (row index | sign << 31); nullptr afterwards
    const uint32_t* desc;      // per output element: first input index | pair << 31
    const uint32_t* n_out_p;   // number of output elements (device memory: the host only has a bound)
    Fp* prefix;                // n_out x LANES running products
    Aff<C>* out;
    uint32_t groups;           // lane groups in the grid
    uint32_t bmin;             // minimum batch per lane group (one inversion each)
};

// One lane of a round.  t = global lane-group index, comp = this lane's coefficient, live = lane belongs to a group.
template <class C, class FS> struct AffRoundLane {
    static constexpr int LANES = FS::LANES;
    static constexpr uint32_t TPW = 64 / LANES;
    typedef Fp T;

    GH_HD static const Fp* coef(const Aff<C>* pt, int e, int comp) { return reinterpret_cast<const Fp*>(pt) + LANES * e + comp; }

    struct Elem {          // one output element's inputs as this lane sees them
        bool act, pair;
        uint32_t i1, i2;   // input records
        bool n1, n2;       // negate y (round 0 sign bits)
    };
    GH_HD static Elem elem_of(const AffRoundArgs<C>& a, uint32_t o, uint32_t n_out, bool live) {
        Elem e;
        e.act = live && o < n_out;
        const uint32_t de = e.act ? a.desc[o] : 0u;
        const uint32_t ai = de & 0x7FFFFFFFu;
        e.pair = e.act && (de >> 31) != 0;
        if (a.sorted != nullptr) {
            const uint32_t s1 = e.act ? a.sorted[ai] : 0u, s2 = e.pair ? a.sorted[ai + 1] : s1;
            e.i1 = s1 & 0x7FFFFFFFu; e.i2 = s2 & 0x7FFFFFFFu;
            e.n1 = (s1 >> 31) != 0; e.n2 = (s2 >> 31) != 0;
        } else {
            e.i1 = ai; e.i2 = e.pair ? ai + 1 : ai;
            e.n1 = e.n2 = false;
        }
        return e;
    }
    GH_HD static T ld_y(const AffRoundArgs<C>& a, uint32_t idx, bool negate, int comp) {
        T y = ld_fp(coef(a.in + idx, 1, comp));
        const T ny = FS::neg(y);
        GH_UNROLL for (int i = 0; i < NL; i++) y.l[i] = negate ? ny.l[i] : y.l[i];
        return y;
    }

    // kind of an element: 0 copy / pass-through (d = 1), 1 generic addition, 2 doubling, 3 cancellation
    // d: the denominator whose inverse the addition needs (1 where none is needed)
    // The equal-x cases need the y coordinates: they are loaded behind a wave-uniform branch.
    GH_HD static int classify(const AffRoundArgs<C>& a, const Elem& e, const T& x1, const T& x2, int comp, T& d) {
        d = FS::one();
        int kind = 0;
        const bool m1 = x1.l[0] == AFF_MARK, m2 = x2.l[0] == AFF_MARK;
        const bool both = e.pair && !m1 && !m2;
        const bool eqx = both && FS::eq(x1, x2);
        if (both && !eqx) { d = FS::sub(x2, x1); kind = 1; }
        if (aff_any(eqx)) {
            const T y1 = ld_y(a, e.i1, e.n1, comp), y2 = ld_y(a, e.i2, e.n2, comp);
            const bool dbl = eqx && FS::eq(y1, y2) && !FS::is_zero(y1);
            const T y2x = FS::dbl(y1);
            if (dbl) { d = y2x; kind = 2; } else if (eqx) kind = 3;
        }
        return kind;
    }

    GH_HD static void run(const AffRoundArgs<C>& a, uint32_t t, int comp, bool live) {
        const uint32_t n_out = *a.n_out_p;
        uint32_t B = (n_out + a.groups - 1) / a.groups;
        if (B < a.bmin) B = a.bmin;
        const uint32_t wv = t / TPW, g = t % TPW;
        const uint64_t chunk64 = (uint64_t)wv * TPW * B;
        if (chunk64 >= n_out) return;                         // the whole wave: nothing left for it
        const uint32_t chunk = (uint32_t)chunk64;
        // ---- forward: running product of the denominators
        T acc = FS::one();
        for (uint32_t k = 0; k < B; k++) {
            const uint32_t o = chunk + k * TPW + g;
            const Elem e = elem_of(a, o, n_out, live);
            const T x1 = ld_fp(coef(a.in + e.i1, 0, comp)), x2 = ld_fp(coef(a.in + e.i2, 0, comp));
            T d;
            classify(a, e, x1, x2, comp, d);
            acc = FS::mul(acc, d);
            if (e.act) st_fp(a.prefix + (size_t)o * LANES + comp, acc);
        }
        // ---- one inversion for the lane group's whole batch
        T inv = FS::inv(acc);
        // ---- backward
        for (uint32_t k = B; k-- > 0;) {
            const uint32_t o = chunk + k * TPW + g;
            const Elem e = elem_of(a, o, n_out, live);
            const T x1 = ld_fp(coef(a.in + e.i1, 0, comp)), x2 = ld_fp(coef(a.in + e.i2, 0, comp));
            T pk = FS::one();
            if (e.act && k > 0) pk = ld_fp(a.prefix + (size_t)(o - TPW) * LANES + comp);
            T d;
            const int kind = classify(a, e, x1, x2, comp, d);
            const T dinv = FS::mul(inv, pk);                 // 1 / d_k
            inv = FS::mul(inv, d);
            const T y1 = ld_y(a, e.i1, e.n1, comp), y2 = ld_y(a, e.i2, e.n2, comp);
            T num = FS::sub(y2, y1);
            if (aff_any(kind == 2)) {                        // 3 x1^2 + a
                const T xx = FS::sqr(x1);
                const T n2 = FS::add(FS::add(FS::dbl(xx), xx), CurveA<C, FS>::get(comp));
                GH_UNROLL for (int i = 0; i < NL; i++) num.l[i] = kind == 2 ? n2.l[i] : num.l[i];
            }
            const T lam = FS::mul(num, dinv);
            T x3 = FS::sub(FS::sub(FS::sqr(lam), x1), x2);
            T y3 = FS::sub(FS::mul(lam, FS::sub(x1, x3)), y1);
            // select the result: copy / marker + Q / P + marker / cancellation
            const bool m1 = x1.l[0] == AFF_MARK, m2 = x2.l[0] == AFF_MARK;
            const bool take1 = !e.pair || m2, take2 = e.pair && m1 && !m2;
            GH_UNROLL for (int i = 0; i < NL; i++) {
                x3.l[i] = take1 ? x1.l[i] : (take2 ? x2.l[i] : x3.l[i]);
                y3.l[i] = take1 ? y1.l[i] : (take2 ? y2.l[i] : y3.l[i]);
            }
            if (kind == 3) {
                GH_UNROLL for (int i = 0; i < NL; i++) { x3.l[i] = 0; y3.l[i] = 0; }
                x3.l[0] = AFF_MARK;
            }
            if (e.act) {
                Fp* po = reinterpret_cast<Fp*>(a.out + o);
                st_fp(po + comp, x3);
                st_fp(po + LANES + comp, y3);
            }
        }
    }
};

// ---- planning (plain index arithmetic; host-callable bodies for the CPU emulation test)
// cnt[(r - 1) * stride + b] = m_r[b] = ceil(counts[b] / 2^r), r = 1 .. R
GH_HD void aff_counts_body(const uint32_t* counts, uint32_t b, int R, size_t stride, uint32_t* cnt) {
    uint32_t m = counts[b];
    for (int r = 1; r <= R; r++) { m = (m + 1) >> 1; cnt[(size_t)(r - 1) * stride + b] = m; }
}
// descriptor of output o of a round: st_in / m_in describe the input list, st_out the output list
GH_HD uint32_t aff_desc_body(const uint32_t* st_in, const uint32_t* m_in, const uint32_t* st_out, uint32_t total, uint32_t o) {
    uint32_t lo = 0, hi = total;          // last b with st_out[b] <= o
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (st_out[mid] <= o) lo = mid; else hi = mid; }
    const uint32_t j = o - st_out[lo];
    const uint32_t ai = st_in[lo] + 2 * j;
    return ai | ((2 * j + 1 < m_in[lo]) ? 0x80000000u : 0u);
}

#if defined(__HIPCC__)
static __global__ void __launch_bounds__(256)
aff_counts_kernel(const uint32_t* __restrict__ counts, uint32_t total, int R, size_t stride, uint32_t* __restrict__ cnt) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < total) aff_counts_body(counts, b, R, stride, cnt);
}
// n_out[r] = st[last] + m[last] of round r's list (r = 1 .. R), n_out[0] = entries of the sorted list
static __global__ void aff_totals_kernel(const uint32_t* __restrict__ starts, const uint32_t* __restrict__ counts,
                                         const uint32_t* __restrict__ st, const uint32_t* __restrict__ cnt, uint32_t total,
                                         int R, size_t stride, uint32_t* __restrict__ n_out) {
    const int r = threadIdx.x;
    if (r == 0) n_out[0] = starts[total - 1] + counts[total - 1];
    else if (r <= R) n_out[r] = st[(size_t)(r - 1) * stride + total - 1] + cnt[(size_t)(r - 1) * stride + total - 1];
}
static __global__ void __launch_bounds__(256)
aff_desc_kernel(const uint32_t* __restrict__ st_in, const uint32_t* __restrict__ m_in, const uint32_t* __restrict__ st_out,
                uint32_t total, const uint32_t* __restrict__ n_out_p, uint32_t* __restrict__ desc) {
    const uint32_t n_out = *n_out_p;
    for (uint32_t o = blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += gridDim.x * blockDim.x)
        desc[o] = aff_desc_body(st_in, m_in, st_out, total, o);
}

template <class C, class FS>
__global__ void __launch_bounds__(256, FS::WAVES) aff_round_kernel(AffRoundArgs<C> a) {
    constexpr uint32_t TPW = 64 / FS::LANES;
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const bool live = lane < TPW * FS::LANES;
    const uint32_t t = wave * TPW + (live ? lane / FS::LANES : TPW - 1);
    AffRoundLane<C, FS>::run(a, t, (int)(lane % FS::LANES), live && t < a.groups);
}
#endif

}  // namespace gh

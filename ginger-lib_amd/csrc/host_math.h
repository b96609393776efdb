// host_math.h -- the few host-only helpers the library needs on top of fp29.h / ec29.h:
// field inversion (Fermat) for size_inv / g^-1 (domain.rs:83-93) and for the affine conversion
// of a projective result (short_weierstrass_projective.rs:663-678).  Not on the hot path.
#pragma once
#include "ec29.h"

namespace gh {

template <class P> inline Fp host_fp_pow_pm2(const Fp& a) {
    // exponent p - 2 as 26 x 29-bit limbs
    uint32_t e[NL];
    int32_t bw = -2;
    for (int i = 0; i < NL; i++) {
        int32_t x = (int32_t)P::P[i] + bw;
        e[i] = (uint32_t)x & LM;
        bw = x >> 31;
    }
    Fp r = fp_one<P>();
    bool started = false;
    for (int bit = NL * LB - 1; bit >= 0; bit--) {
        if (started) r = fp_sqr<P>(r);
        if ((e[bit / LB] >> (bit % LB)) & 1) {
            r = started ? fp_mul<P>(r, a) : a;
            started = true;
        }
    }
    return r;
}
template <class P> inline Fp host_fp_inv(const Fp& a) { return host_fp_pow_pm2<P>(a); }

template <class F> struct HostInv;
template <class P, bool I> struct HostInv<F1<P, I>> {
    static Fp inv(const Fp& a) { return host_fp_inv<P>(a); }
};
template <class P, int NR, bool I> struct HostInv<F2<P, NR, I>> {
    typedef Fp2T T;
    // (a0 + a1 X)^-1 = (a0 - a1 X) / (a0^2 - NR a1^2)          (fp2.rs inverse)
    static T inv(const T& a) {
        Fp n = fp_sub<P>(fp_sqr<P>(a.c0), fp_mul_small<P, NR>(fp_sqr<P>(a.c1)));
        Fp ni = host_fp_inv<P>(n);
        return T{fp_mul<P>(a.c0, ni), fp_neg<P>(fp_mul<P>(a.c1, ni))};
    }
};
template <class P, int NR, bool I> struct HostInv<F3<P, NR, I>> {
    typedef Fp3T T;
    // norm-based inverse in Fp[X]/(X^3 - NR)                    (fp3.rs inverse)
    static T inv(const T& a) {
        Fp t0 = fp_sqr<P>(a.c0), t1 = fp_sqr<P>(a.c1), t2 = fp_sqr<P>(a.c2);
        Fp t3 = fp_mul<P>(a.c0, a.c1), t4 = fp_mul<P>(a.c0, a.c2), t5 = fp_mul<P>(a.c1, a.c2);
        Fp c0 = fp_sub<P>(t0, fp_mul_small<P, NR>(t5));
        Fp c1 = fp_sub<P>(fp_mul_small<P, NR>(t2), t3);
        Fp c2 = fp_sub<P>(t1, t4);
        Fp n = fp_add<P>(fp_mul<P>(a.c0, c0),
                         fp_mul_small<P, NR>(fp_add<P>(fp_mul<P>(a.c2, c1), fp_mul<P>(a.c1, c2))));
        Fp ni = host_fp_inv<P>(n);
        return T{fp_mul<P>(c0, ni), fp_mul<P>(c1, ni), fp_mul<P>(c2, ni)};
    }
};
template <class F> inline typename F::T host_inv(const typename F::T& a) { return HostInv<F>::inv(a); }

// ------------------------------------------------------------------------------------------
// Host-side field for the window fold of G1 results: 12 x u64 limbs, Montgomery radix 2^768 --
// i.e. exactly the ABI representation -- with a plain CIOS product on unsigned __int128.  The
// fold is ~750 dependent doublings; on the host a 64-bit-limb product costs about a third of the
// 26 x 29-bit one.  Same static interface as F1<P>, so ec29.h's proj_dbl / proj_add apply.
struct H64 { uint64_t l[12]; };
template <class P> struct HostConsts;
template <> struct HostConsts<P4> {
    static constexpr uint64_t MOD[12] = GH_P4_P_64, ONE[12] = GH_P4_R_64;
    static constexpr uint64_t INV = GH_P4_INV64;
};
template <> struct HostConsts<P6> {
    static constexpr uint64_t MOD[12] = GH_P6_P_64, ONE[12] = GH_P6_R_64;
    static constexpr uint64_t INV = GH_P6_INV64;
};
template <class P> struct HF1 {
    typedef H64 T;
    typedef unsigned __int128 u128;
    typedef HostConsts<P> K;
    static constexpr int DEG = 1;
    static T zero() { T r; for (int i = 0; i < 12; i++) r.l[i] = 0; return r; }
    static T one() { T r; for (int i = 0; i < 12; i++) r.l[i] = K::ONE[i]; return r; }
    static bool is_zero(const T& a) { uint64_t o = 0; for (int i = 0; i < 12; i++) o |= a.l[i]; return o == 0; }
    static bool eq(const T& a, const T& b) { uint64_t o = 0; for (int i = 0; i < 12; i++) o |= a.l[i] ^ b.l[i]; return o == 0; }
    static bool geq_mod(const uint64_t* t) {
        for (int i = 11; i >= 0; i--) { if (t[i] > K::MOD[i]) return true; if (t[i] < K::MOD[i]) return false; }
        return true;
    }
    static void sub_mod(uint64_t* t) { uint64_t bw = 0; for (int i = 0; i < 12; i++) { u128 x = (u128)t[i] - K::MOD[i] - bw; t[i] = (uint64_t)x; bw = (uint64_t)(x >> 64) & 1; } }
    static T add(const T& a, const T& b) {
        T r; uint64_t c = 0;
        for (int i = 0; i < 12; i++) { u128 x = (u128)a.l[i] + b.l[i] + c; r.l[i] = (uint64_t)x; c = (uint64_t)(x >> 64); }
        if (geq_mod(r.l)) sub_mod(r.l);   // no carry out: 2p < 2^768
        return r;
    }
    static T dbl(const T& a) { return add(a, a); }
    static T sub(const T& a, const T& b) {
        T r; uint64_t bw = 0;
        for (int i = 0; i < 12; i++) { u128 x = (u128)a.l[i] - b.l[i] - bw; r.l[i] = (uint64_t)x; bw = (uint64_t)(x >> 64) & 1; }
        if (bw) { uint64_t c = 0; for (int i = 0; i < 12; i++) { u128 x = (u128)r.l[i] + K::MOD[i] + c; r.l[i] = (uint64_t)x; c = (uint64_t)(x >> 64); } }
        return r;
    }
    static T neg(const T& a) { return is_zero(a) ? a : sub(zero(), a); }
    static T mul(const T& a, const T& b) {
        uint64_t t[14];
        for (int i = 0; i < 14; i++) t[i] = 0;
        for (int i = 0; i < 12; i++) {
            uint64_t c = 0;
            for (int j = 0; j < 12; j++) { u128 x = (u128)a.l[i] * b.l[j] + t[j] + c; t[j] = (uint64_t)x; c = (uint64_t)(x >> 64); }
            u128 y = (u128)t[12] + c; t[12] = (uint64_t)y; t[13] = (uint64_t)(y >> 64);
            const uint64_t m = t[0] * K::INV;
            c = (uint64_t)(((u128)m * K::MOD[0] + t[0]) >> 64);
            for (int j = 1; j < 12; j++) { u128 x = (u128)m * K::MOD[j] + t[j] + c; t[j - 1] = (uint64_t)x; c = (uint64_t)(x >> 64); }
            y = (u128)t[12] + c; t[11] = (uint64_t)y; t[12] = t[13] + (uint64_t)(y >> 64);
        }
        if (t[12] || geq_mod(t)) sub_mod(t);
        T r; for (int i = 0; i < 12; i++) r.l[i] = t[i];
        return r;
    }
    static T sqr(const T& a) { return mul(a, a); }
    static T mul_sub_mul(const T& a, const T& b, const T& c, const T& d) { return sub(mul(a, b), mul(c, d)); }
    static T mul_small(const T& a, int k) {   // k * a by double-and-add
        T acc = a; int top = 30; while (!((k >> top) & 1)) top--;
        for (int b = top - 1; b >= 0; b--) { acc = dbl(acc); if ((k >> b) & 1) acc = add(acc, a); }
        return acc;
    }
};
// towers over the 64-bit-limb host field, same formulas as F2 / F3 (fp2.rs:389-400, :128-144;
// fp3.rs:453-477, :165-185), for the window fold of G2 results
template <class B, int NR> struct HF2 {
    struct T { H64 c0, c1; };
    static constexpr int DEG = 2;
    static T zero() { return T{B::zero(), B::zero()}; }
    static T one() { return T{B::one(), B::zero()}; }
    static T add(const T& a, const T& b) { return T{B::add(a.c0, b.c0), B::add(a.c1, b.c1)}; }
    static T sub(const T& a, const T& b) { return T{B::sub(a.c0, b.c0), B::sub(a.c1, b.c1)}; }
    static T dbl(const T& a) { return T{B::dbl(a.c0), B::dbl(a.c1)}; }
    static T neg(const T& a) { return T{B::neg(a.c0), B::neg(a.c1)}; }
    static T mul(const T& a, const T& b) {
        H64 v0 = B::mul(a.c0, b.c0), v1 = B::mul(a.c1, b.c1);
        H64 s = B::mul(B::add(a.c0, a.c1), B::add(b.c0, b.c1));
        return T{B::add(v0, B::mul_small(v1, NR)), B::sub(B::sub(s, v0), v1)};
    }
    static T sqr(const T& a) {
        H64 v0 = B::sub(a.c0, a.c1), v3 = B::sub(a.c0, B::mul_small(a.c1, NR)), v2 = B::mul(a.c0, a.c1);
        H64 t = B::mul(v0, v3);
        return T{B::add(B::add(t, v2), B::mul_small(v2, NR)), B::dbl(v2)};
    }
    static T mul_sub_mul(const T& a, const T& b, const T& c, const T& d) { return sub(mul(a, b), mul(c, d)); }
    static bool is_zero(const T& a) { return B::is_zero(a.c0) && B::is_zero(a.c1); }
    static bool eq(const T& a, const T& b) { return B::eq(a.c0, b.c0) && B::eq(a.c1, b.c1); }
};
template <class B, int NR> struct HF3 {
    struct T { H64 c0, c1, c2; };
    static constexpr int DEG = 3;
    static T zero() { return T{B::zero(), B::zero(), B::zero()}; }
    static T one() { return T{B::one(), B::zero(), B::zero()}; }
    static T add(const T& a, const T& b) { return T{B::add(a.c0, b.c0), B::add(a.c1, b.c1), B::add(a.c2, b.c2)}; }
    static T sub(const T& a, const T& b) { return T{B::sub(a.c0, b.c0), B::sub(a.c1, b.c1), B::sub(a.c2, b.c2)}; }
    static T dbl(const T& a) { return T{B::dbl(a.c0), B::dbl(a.c1), B::dbl(a.c2)}; }
    static T neg(const T& a) { return T{B::neg(a.c0), B::neg(a.c1), B::neg(a.c2)}; }
    static T mul(const T& A, const T& Bv) {
        const H64 &a = Bv.c0, &b = Bv.c1, &c = Bv.c2, &d = A.c0, &e = A.c1, &f = A.c2;
        H64 ad = B::mul(d, a), be = B::mul(e, b), cf = B::mul(f, c);
        H64 x = B::sub(B::sub(B::mul(B::add(e, f), B::add(b, c)), be), cf);
        H64 y = B::sub(B::sub(B::mul(B::add(d, e), B::add(a, b)), ad), be);
        H64 z = B::sub(B::add(B::sub(B::mul(B::add(d, f), B::add(a, c)), ad), be), cf);
        return T{B::add(ad, B::mul_small(x, NR)), B::add(y, B::mul_small(cf, NR)), z};
    }
    static T sqr(const T& A) {
        const H64 &a = A.c0, &b = A.c1, &c = A.c2;
        H64 s0 = B::sqr(a), ab = B::mul(a, b), s1 = B::dbl(ab), s2 = B::sqr(B::add(B::sub(a, b), c));
        H64 bc = B::mul(b, c), s3 = B::dbl(bc), s4 = B::sqr(c);
        return T{B::add(s0, B::mul_small(s3, NR)), B::add(s1, B::mul_small(s4, NR)),
                 B::sub(B::sub(B::add(B::add(s1, s2), s3), s0), s4)};
    }
    static T mul_sub_mul(const T& a, const T& b, const T& c, const T& d) { return sub(mul(a, b), mul(c, d)); }
    static bool is_zero(const T& a) { return B::is_zero(a.c0) && B::is_zero(a.c1) && B::is_zero(a.c2); }
    static bool eq(const T& a, const T& b) { return B::eq(a.c0, b.c0) && B::eq(a.c1, b.c1) && B::eq(a.c2, b.c2); }
};
struct HostMnt4G2 {   // a' = (26, 0): curves/mnt4753/g2.rs:113-118
    typedef HF2<HF1<P4>, 13> F; typedef F FC;
    static F::T mul_by_a(const F::T& z) { return F::T{HF1<P4>::mul_small(z.c0, 26), HF1<P4>::mul_small(z.c1, 26)}; }
};
struct HostMnt6G2 {   // a' = (0, 0, 11): curves/mnt6753/g2.rs:149-155
    typedef HF3<HF1<P6>, 11> F; typedef F FC;
    static F::T mul_by_a(const F::T& z) { return F::T{HF1<P6>::mul_small(z.c1, 121), HF1<P6>::mul_small(z.c2, 121), HF1<P6>::mul_small(z.c0, 11)}; }
};
struct HostMnt4G1 { typedef HF1<P4> F; typedef HF1<P4> FC; static H64 mul_by_a(const H64& z) { return HF1<P4>::dbl(z); } };
struct HostMnt6G1 { typedef HF1<P6> F; typedef HF1<P6> FC; static H64 mul_by_a(const H64& z) { return HF1<P6>::mul_small(z, 11); } };
template <class C> struct HostCurveOf { typedef C type; static constexpr bool fast = false; };
template <> struct HostCurveOf<Mnt4G1> { typedef HostMnt4G1 type; static constexpr bool fast = true; };
template <> struct HostCurveOf<Mnt6G1> { typedef HostMnt6G1 type; static constexpr bool fast = true; };
template <> struct HostCurveOf<Mnt4G2> { typedef HostMnt4G2 type; static constexpr bool fast = true; };
template <> struct HostCurveOf<Mnt6G2> { typedef HostMnt6G2 type; static constexpr bool fast = true; };

}  // namespace gh

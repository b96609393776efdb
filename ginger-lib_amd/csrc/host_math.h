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
template <class P> struct HostInv<F1<P>> {
    static Fp inv(const Fp& a) { return host_fp_inv<P>(a); }
};
template <class P, int NR> struct HostInv<F2<P, NR>> {
    typedef typename F2<P, NR>::T T;
    // (a0 + a1 X)^-1 = (a0 - a1 X) / (a0^2 - NR a1^2)          (fp2.rs inverse)
    static T inv(const T& a) {
        Fp n = fp_sub<P>(fp_sqr<P>(a.c0), fp_mul_small<P, NR>(fp_sqr<P>(a.c1)));
        Fp ni = host_fp_inv<P>(n);
        return T{fp_mul<P>(a.c0, ni), fp_neg<P>(fp_mul<P>(a.c1, ni))};
    }
};
template <class P, int NR> struct HostInv<F3<P, NR>> {
    typedef typename F3<P, NR>::T T;
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

}  // namespace gh

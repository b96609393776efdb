// ec29.h -- short-Weierstrass curve arithmetic in homogeneous projective coordinates
// (X:Y:Z), infinity <=> Z == 0, over the fields of fp29.h.  Formulas are the ones the reference
// uses for MNT4/6-753 G1 and G2 (algebra/src/curves/models/short_weierstrass_projective.rs):
//   double_in_place  :444-479  (dbl-2007-bl)
//   add_assign_mixed :481-519  (madd-1998-cmo, with the explicit infinity / P==Q branches)
//   add_assign       :574-617  (add-1998-cmo-2)
// The projective triple an MSM returns depends on the order of additions; only the affine image
// is canonical, so the device is free to order additions for parallelism.
#pragma once
#include "fp29.h"

namespace gh {

// ----- curve policies: base-field tower + multiplication by the curve coefficient a
// F  = field policy with the Fp product inlined (G1 accumulation inner loop);
// FC = the same field with out-of-line products, used by the proj_*_call instances below.
struct Mnt4G1 {  // y^2 = x^3 + 2x + b over p4       (curves/mnt4753/g1.rs:19-50)
    typedef P4 PF;
    typedef F1<P4, true> F;
    typedef F1<P4, false> FC;
    static GH_HD F::T mul_by_a(const F::T& z) { return fp_dbl<P4>(z); }
};
struct Mnt6G1 {  // y^2 = x^3 + 11x + b over p6      (curves/mnt6753/g1.rs:19-52)
    typedef P6 PF;
    typedef F1<P6, true> F;
    typedef F1<P6, false> FC;
    static GH_HD F::T mul_by_a(const F::T& z) { return fp_mul_small<P6, 11>(z); }
};
struct Mnt4G2 {  // twist over Fq2, a' = (26, 0)     (curves/mnt4753/g2.rs:57-75, mul_by_a :113-118)
    typedef P4 PF;
    typedef F2<P4, 13, false> F;   // inlining the 31 Fp products of an Fq2 mixed addition was measured
    typedef F2<P4, 13, false> FC;  // SLOWER (3.7 KB of spills per lane) than out-of-line products
    static GH_HD F::T mul_by_a(const F::T& z) { return F::T{fp_mul_small<P4, 26>(z.c0), fp_mul_small<P4, 26>(z.c1)}; }
};
struct Mnt6G2 {  // twist over Fq3, a' = (0, 0, 11)  (curves/mnt6753/g2.rs:71-100, mul_by_a :149-155)
    typedef P6 PF;
    typedef F3<P6, 11, false> F;
    typedef F3<P6, 11, false> FC;
    static GH_HD F::T mul_by_a(const F::T& z) {
        return F::T{fp_mul_small<P6, 121>(z.c1), fp_mul_small<P6, 121>(z.c2), fp_mul_small<P6, 11>(z.c0)};
    }
};

template <class C> struct Proj {
    typename C::F::T x, y, z;
};
template <class C> struct Aff {
    typename C::F::T x, y;
};

template <class C> GH_HD Proj<C> proj_zero() {  // (0, 1, 0)   swp.rs:372-378
    typedef typename C::F F;
    return Proj<C>{F::zero(), F::one(), F::zero()};
}
template <class C> GH_HD bool proj_is_zero(const Proj<C>& p) { return C::F::is_zero(p.z); }

template <class C, class F = typename C::F> GH_HD Proj<C> proj_dbl(const Proj<C>& p) {
    if (proj_is_zero(p)) return p;
    typename F::T xx = F::sqr(p.x);
    typename F::T zz = F::sqr(p.z);
    typename F::T w = F::add(C::mul_by_a(zz), F::add(xx, F::dbl(xx)));
    typename F::T s = F::dbl(F::mul(p.y, p.z));
    typename F::T sss = F::mul(F::sqr(s), s);
    typename F::T r = F::mul(p.y, s);
    typename F::T rr = F::sqr(r);
    typename F::T b = F::sub(F::sub(F::sqr(F::add(p.x, r)), xx), rr);
    typename F::T h = F::sub(F::sqr(w), F::dbl(b));
    Proj<C> o;
    o.x = F::mul(h, s);
    o.y = F::sub(F::mul(w, F::sub(b, h)), F::dbl(rr));
    o.z = sss;
    return o;
}

template <class C> GH_HD_NOINLINE Proj<C> proj_dbl_call(const Proj<C>& p);

// p += q, q affine and NOT the point at infinity (callers drop infinity bases: swp.rs:482-483)
template <class C, class F = typename C::F> GH_HD Proj<C> proj_madd(const Proj<C>& p, const Aff<C>& q) {
    if (proj_is_zero(p)) return Proj<C>{q.x, q.y, F::one()};
    typename F::T v = F::mul(q.x, p.z);
    typename F::T u = F::mul(q.y, p.z);
    if (F::eq(u, p.y) && F::eq(v, p.x)) return proj_dbl_call<C>(p);
    u = F::sub(u, p.y);
    typename F::T uu = F::sqr(u);
    v = F::sub(v, p.x);
    typename F::T vv = F::sqr(v);
    typename F::T vvv = F::mul(v, vv);
    typename F::T r = F::mul(vv, p.x);
    typename F::T a = F::sub(F::sub(F::mul(uu, p.z), vvv), F::dbl(r));
    Proj<C> o;
    o.x = F::mul(v, a);
    o.y = F::mul_sub_mul(u, F::sub(r, a), vvv, p.y);
    o.z = F::mul(vvv, p.z);
    return o;
}

template <class C, class F = typename C::F> GH_HD Proj<C> proj_add(const Proj<C>& p, const Proj<C>& q) {
    if (proj_is_zero(p)) return q;
    if (proj_is_zero(q)) return p;
    typename F::T y1z2 = F::mul(p.y, q.z);
    typename F::T x1z2 = F::mul(p.x, q.z);
    typename F::T z1z2 = F::mul(p.z, q.z);
    typename F::T u = F::sub(F::mul(p.z, q.y), y1z2);
    typename F::T v = F::sub(F::mul(p.z, q.x), x1z2);
    if (F::is_zero(u) && F::is_zero(v)) return proj_dbl_call<C>(p);  // same point (swp.rs:586)
    typename F::T uu = F::sqr(u);
    typename F::T vv = F::sqr(v);
    typename F::T vvv = F::mul(v, vv);
    typename F::T r = F::mul(vv, x1z2);
    typename F::T a = F::sub(F::sub(F::mul(uu, z1z2), vvv), F::dbl(r));
    Proj<C> o;
    o.x = F::mul(v, a);
    o.y = F::mul_sub_mul(F::sub(r, a), u, vvv, y1z2);
    o.z = F::mul(vvv, z1z2);
    return o;
}

// Out-of-line instances for everything that is not the accumulation inner loop.  They are built
// on the out-of-line Fp product (FC), which keeps each of them a few KB.  This is not only about
// code size: hipcc (ROCm 7.2) relaxes a branch that spans more than 2^15 dwords into
// s_getpc_b64 s[30:31] / s_setpc_b64 and in a LEAF function s[30:31] still holds the return
// address, so a 140 KB leaf proj_dbl with an early `return p` never returned (it looped on its
// own epilogue).  Small callees have no long branches.
template <class C> GH_HD_NOINLINE Proj<C> proj_dbl_call(const Proj<C>& p) { return proj_dbl<C, typename C::FC>(p); }
template <class C> GH_HD_NOINLINE Proj<C> proj_add_call(const Proj<C>& p, const Proj<C>& q) { return proj_add<C, typename C::FC>(p, q); }
template <class C> GH_HD_NOINLINE Proj<C> proj_madd_call(const Proj<C>& p, const Aff<C>& q) { return proj_madd<C, typename C::FC>(p, q); }

template <class C> GH_HD Aff<C> aff_neg(const Aff<C>& q) { return Aff<C>{q.x, C::F::neg(q.y)}; }

// ----- extended Jacobian ("XYZZ") bucket accumulators:  x = X / ZZ,  y = Y / ZZZ,  ZZ^3 = ZZZ^2,  infinity <=> ZZ == 0.
// The reference's bucket update is add_assign_mixed (madd-1998-cmo, swp.rs:481-519: 9 M + 2 S on (X : Y : Z)).  A bucket sum
// is only ever read through its affine image, so the accumulator is free to use the cheaper mixed addition
// madd-2008-s (8 M + 2 S, no curve coefficient), and Y3 = R (Q - X3) - Y1 PPP is ONE dual product (fp_mul2s):
// 10 multiplications / 9 reductions per bucket update instead of 11 / 11.  xyzz_to_proj() hands the sum to the
// projective bucket reduction (3 M per bucket, once).
template <class C> struct Xyzz {
    typename C::F::T x, y, zz, zzz;
};
template <class C> GH_HD Xyzz<C> xyzz_zero() {
    typedef typename C::F F;
    return Xyzz<C>{F::zero(), F::one(), F::zero(), F::zero()};
}
template <class C> GH_HD bool xyzz_is_zero(const Xyzz<C>& p) { return C::F::is_zero(p.zz); }
// p += q for q affine, not infinity, and q != +-p unless p is infinity (P == Q is the caller's business: `same` reports it and
// p is returned unchanged; P == -Q needs no case: PP = 0 makes ZZ3 = ZZZ3 = 0)
template <class C, class F = typename C::F> GH_HD Xyzz<C> xyzz_madd(const Xyzz<C>& p, const Aff<C>& q, bool& same) {
    same = false;
    if (xyzz_is_zero(p)) return Xyzz<C>{q.x, q.y, F::one(), F::one()};
    typename F::T pp = F::sub(F::mul(q.x, p.zz), p.x);         // P = U2 - X1
    typename F::T r = F::sub(F::mul(q.y, p.zzz), p.y);         // R = S2 - Y1
    if (F::is_zero(pp) && F::is_zero(r)) { same = true; return p; }
    typename F::T p2 = F::sqr(pp);
    typename F::T p3 = F::mul(pp, p2);
    typename F::T qq = F::mul(p.x, p2);
    Xyzz<C> o;
    o.x = F::sub(F::sub(F::sqr(r), p3), F::dbl(qq));
    o.y = F::mul_sub_mul(r, F::sub(qq, o.x), p.y, p3);
    o.zz = F::mul(p.zz, p2);
    o.zzz = F::mul(p.zzz, p3);
    return o;
}
// (X : Y : ZZ : ZZZ) -> homogeneous (X ZZZ : Y ZZ : ZZ ZZZ); infinity -> (0, 1, 0)
template <class C, class F = typename C::F> GH_HD Proj<C> xyzz_to_proj(const Xyzz<C>& p) {
    if (xyzz_is_zero(p)) return proj_zero<C>();
    return Proj<C>{F::mul(p.x, p.zzz), F::mul(p.y, p.zz), F::mul(p.zz, p.zzz)};
}

}  // namespace gh

// oracle.hpp -- CPU restatement of the reference's algorithms for the hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product (ginger-lib_amd/, include/) includes, links
// or loads this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and
// only as the checker / the timed CPU baseline.
//
// The reference (ZencashOfficial/ginger-lib) is Rust and cannot be built in this environment
// (no rustc/cargo), so this file restates its algorithms in C++ function by
// function, each citing the file:line it follows (paths relative to /root/reference/algebra/src).
// Pinned by: the reference's own known-answer tests re-expressed as fixtures
// (tests/golden/ref_kats.json, extracted by tests/golden/extract_ref_kats.py) and by
// first-principles big-integer golden vectors (tests/golden/*.json from tests/golden/gen_golden.py).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <cmath>
#include <functional>
#include <thread>
#include <vector>
#include "constants_gen.h"

namespace oracle {

typedef unsigned __int128 u128;
constexpr int N = 12;

// ---- biginteger/mod.rs:108-141  (arithmetic::{adc, sbb, mac_with_carry})
static inline uint64_t adc(uint64_t a, uint64_t b, uint64_t& carry) {
    u128 t = (u128)a + b + carry;
    carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}
static inline uint64_t sbb(uint64_t a, uint64_t b, uint64_t& borrow) {
    u128 t = ((u128)1 << 64) + a - b - borrow;
    borrow = (t >> 64) == 0 ? 1 : 0;
    return (uint64_t)t;
}
static inline uint64_t mac_with_carry(uint64_t a, uint64_t b, uint64_t c, uint64_t& carry) {
    u128 t = (u128)a + (u128)b * c + carry;
    carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}

// ---- biginteger/macros.rs:4-275  (BigInteger768)
struct Big {
    uint64_t l[N];
    bool is_zero() const { for (int i = 0; i < N; i++) if (l[i]) return false; return true; }          // :120
    bool is_even() const { return (l[0] & 1) == 0; }
    bool operator==(const Big& o) const { return memcmp(l, o.l, sizeof l) == 0; }
    bool operator!=(const Big& o) const { return !(*this == o); }
    bool lt(const Big& o) const {                                                                        // Ord :224-237
        for (int i = N - 1; i >= 0; i--) { if (l[i] < o.l[i]) return true; if (l[i] > o.l[i]) return false; }
        return false;
    }
    bool add_nocarry(const Big& o) { uint64_t c = 0; for (int i = 0; i < N; i++) l[i] = adc(l[i], o.l[i], c); return c != 0; }   // :14
    bool sub_noborrow(const Big& o) { uint64_t b = 0; for (int i = 0; i < N; i++) l[i] = sbb(l[i], o.l[i], b); return b != 0; }  // :25
    void mul2() { uint64_t last = 0; for (int i = 0; i < N; i++) { uint64_t t = l[i] >> 63; l[i] = (l[i] << 1) | last; last = t; } }   // :36
    void div2() { uint64_t t = 0; for (int i = N - 1; i >= 0; i--) { uint64_t t2 = l[i] << 63; l[i] = (l[i] >> 1) | t; t = t2; } }      // :73
    void divn(uint32_t n) {                                                                              // :84-107
        if (n >= 64 * N) { memset(l, 0, sizeof l); return; }
        while (n >= 64) { uint64_t t = 0; for (int i = N - 1; i >= 0; i--) { uint64_t x = l[i]; l[i] = t; t = x; } n -= 64; }
        if (n > 0) { uint64_t t = 0; for (int i = N - 1; i >= 0; i--) { uint64_t t2 = l[i] << (64 - n); l[i] = (l[i] >> n) | t; t = t2; } }
    }
    static Big from_u64(uint64_t v) { Big b; memset(b.l, 0, sizeof b.l); b.l[0] = v; return b; }        // :266
};

// ---- parameter packs (fields/mnt4753/fq.rs:18-112, fields/mnt6753/fq.rs:17-111)
struct P4 {
    static constexpr uint64_t MODULUS[N] = GH_P4_P_64, R[N] = GH_P4_R_64, R2[N] = GH_P4_R2_64;
    static constexpr uint64_t INV = GH_P4_INV64;
    static constexpr uint64_t GENERATOR[N] = GH_P4_GEN17_M_64, ROOT_OF_UNITY[N] = GH_P4_ROOT_M_64;
    static constexpr int TWO_ADICITY = GH_P4_TWO_ADICITY;
};
struct P6 {
    static constexpr uint64_t MODULUS[N] = GH_P6_P_64, R[N] = GH_P6_R_64, R2[N] = GH_P6_R2_64;
    static constexpr uint64_t INV = GH_P6_INV64;
    static constexpr uint64_t GENERATOR[N] = GH_P6_GEN17_M_64, ROOT_OF_UNITY[N] = GH_P6_ROOT_M_64;
    static constexpr int TWO_ADICITY = GH_P6_TWO_ADICITY;
};
template <class P> static inline Big big_of(const uint64_t (&a)[N]) { Big b; for (int i = 0; i < N; i++) b.l[i] = a[i]; return b; }

// ---- fields/models/fp_768.rs  (Fp768<P>: Montgomery form, R = 2^768)
template <class P> struct Fp {
    Big v;
    static Fp zero() { Fp r; memset(r.v.l, 0, sizeof r.v.l); return r; }                                 // :287
    static Fp one() { Fp r; r.v = big_of<P>(P::R); return r; }                                           // :312-314
    static Big modulus() { return big_of<P>(P::MODULUS); }
    bool is_zero() const { return v.is_zero(); }                                                         // :291
    bool is_one() const { return v == big_of<P>(P::R); }
    bool operator==(const Fp& o) const { return v == o.v; }
    bool is_valid() const { return v.lt(modulus()); }                                                    // :39-41
    void reduce() { if (!is_valid()) v.sub_noborrow(modulus()); }                                        // :44-48
    Fp& add_assign(const Fp& o) { v.add_nocarry(o.v); reduce(); return *this; }                          // :929-937
    Fp& sub_assign(const Fp& o) { if (v.lt(o.v)) v.add_nocarry(modulus()); v.sub_noborrow(o.v); return *this; }  // :939-949
    Fp& double_in_place() { v.mul2(); reduce(); return *this; }                                          // :303-309
    Fp neg() const { if (is_zero()) return *this; Fp r; r.v = modulus(); r.v.sub_noborrow(v); return r; }  // :870-883
    // mont_reduce :50-281 -- 12 rounds k = r[i]*INV, r += k*p*2^(64 i), with the carry2 chain
    void mont_reduce(uint64_t r[2 * N]) {
        uint64_t carry2 = 0;
        for (int i = 0; i < N; i++) {
            uint64_t k = r[i] * P::INV, carry = 0;
            mac_with_carry(r[i], k, P::MODULUS[0], carry);
            for (int j = 1; j < N; j++) r[i + j] = mac_with_carry(r[i + j], k, P::MODULUS[j], carry);
            r[i + N] = adc(r[i + N], carry2, carry);
            carry2 = carry;
        }
        for (int i = 0; i < N; i++) v.l[i] = r[N + i];
        reduce();
    }
    // mul_assign :1009-1185 -- schoolbook rows with mac_with_carry, then mont_reduce
    Fp& mul_assign(const Fp& o) {
        uint64_t r[2 * N];
        memset(r, 0, sizeof r);
        for (int i = 0; i < N; i++) {
            uint64_t carry = 0;
            for (int j = 0; j < N; j++) r[i + j] = mac_with_carry(r[i + j], v.l[i], o.v.l[j], carry);
            r[i + N] = carry;
        }
        mont_reduce(r);
        return *this;
    }
    // square_in_place :339-548 -- off-diagonal products once, doubled, plus the diagonal
    Fp& square_in_place() {
        uint64_t r[2 * N];
        memset(r, 0, sizeof r);
        for (int i = 0; i < N - 1; i++) {
            uint64_t carry = 0;
            for (int j = i + 1; j < N; j++) r[i + j] = mac_with_carry(r[i + j], v.l[i], v.l[j], carry);
            r[i + N] = carry;
        }
        r[2 * N - 1] = r[2 * N - 2] >> 63;
        for (int i = 2 * N - 2; i >= 2; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 63);
        r[1] = r[1] << 1;
        uint64_t carry = 0;
        for (int i = 0; i < N; i++) {
            r[2 * i] = mac_with_carry(r[2 * i], v.l[i], v.l[i], carry);
            r[2 * i + 1] = adc(r[2 * i + 1], 0, carry);
        }
        mont_reduce(r);
        return *this;
    }
    Fp mul(const Fp& o) const { Fp r = *this; r.mul_assign(o); return r; }
    Fp square() const { Fp r = *this; r.square_in_place(); return r; }
    Fp add(const Fp& o) const { Fp r = *this; r.add_assign(o); return r; }
    Fp sub(const Fp& o) const { Fp r = *this; r.sub_assign(o); return r; }
    Fp dbl() const { Fp r = *this; r.double_in_place(); return r; }
    // inverse :551-605 -- binary extended Euclid (Guajardo et al., Alg. 16), b starts at R2
    bool inverse(Fp& out) const {
        if (is_zero()) return false;
        Big one = Big::from_u64(1), u = v, w = modulus();
        Fp b, c = zero();
        b.v = big_of<P>(P::R2);
        while (u != one && w != one) {
            while (u.is_even()) {
                u.div2();
                if (b.v.is_even()) b.v.div2(); else { b.v.add_nocarry(modulus()); b.v.div2(); }
            }
            while (w.is_even()) {
                w.div2();
                if (c.v.is_even()) c.v.div2(); else { c.v.add_nocarry(modulus()); c.v.div2(); }
            }
            if (w.lt(u)) { u.sub_noborrow(w); b.sub_assign(c); } else { w.sub_noborrow(u); c.sub_assign(b); }
        }
        out = (u == one) ? b : c;
        return true;
    }
    // from_repr / into_repr :627-667
    static bool from_repr(const Big& r, Fp& out) {
        Fp t; t.v = r;
        if (!t.is_valid()) return false;
        Fp r2; r2.v = big_of<P>(P::R2);
        t.mul_assign(r2);
        out = t;
        return true;
    }
    Big into_repr() const {
        uint64_t r[2 * N];
        memset(r, 0, sizeof r);
        for (int i = 0; i < N; i++) r[i] = v.l[i];
        Fp t; t.mont_reduce(r);
        return t.v;
    }
    // Field::pow, fields/mod.rs:136-157 -- MSB-first square and multiply over u64 limbs of the exponent
    Fp pow(const uint64_t* e, int nlimbs) const {
        Fp res = one();
        bool found_one = false;
        for (int i = nlimbs * 64 - 1; i >= 0; i--) {
            bool bit = (e[i / 64] >> (i % 64)) & 1;
            if (!found_one) { if (bit) found_one = true; else continue; }
            res.square_in_place();
            if (bit) res.mul_assign(*this);
        }
        return res;
    }
    static Fp multiplicative_generator() { Fp r; r.v = big_of<P>(P::GENERATOR); return r; }
    static Fp root_of_unity() { Fp r; r.v = big_of<P>(P::ROOT_OF_UNITY); return r; }
};

// small-constant multiple used for the tower non-residues (the reference multiplies by the
// Montgomery constant NONRESIDUE: fields/mnt4753/fq2.rs:19, fields/mnt6753/fq3.rs:18 -- same value)
template <class P> static inline Fp<P> fp_from_small(uint64_t k) {
    Fp<P> r; Fp<P>::from_repr(Big::from_u64(k), r); return r;
}

// ---- fields/models/fp2.rs  (Fp2 = Fp[X]/(X^2 - NR))
template <class P, int NR> struct Fp2 {
    typedef Fp<P> B;
    B c0, c1;
    static constexpr int DEG = 2;
    static Fp2 zero() { return Fp2{B::zero(), B::zero()}; }
    static Fp2 one() { return Fp2{B::one(), B::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool is_one() const { return c0.is_one() && c1.is_zero(); }
    bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
    static B nr() { static const B v = fp_from_small<P>(NR); return v; }
    Fp2 add(const Fp2& o) const { return Fp2{c0.add(o.c0), c1.add(o.c1)}; }                              // :373-378
    Fp2 sub(const Fp2& o) const { return Fp2{c0.sub(o.c0), c1.sub(o.c1)}; }                              // :380-385
    Fp2 dbl() const { return Fp2{c0.dbl(), c1.dbl()}; }
    Fp2 neg() const { return Fp2{c0.neg(), c1.neg()}; }                                                  // :312-318
    Fp2 mul(const Fp2& o) const {                                                                        // :389-400 Karatsuba
        B v0 = c0.mul(o.c0), v1 = c1.mul(o.c1);
        Fp2 r;
        r.c1 = c0.add(c1).mul(o.c0.add(o.c1)).sub(v0).sub(v1);
        r.c0 = v0.add(nr().mul(v1));
        return r;
    }
    Fp2 square() const {                                                                                 // :128-144
        B v0 = c0.sub(c1), v3 = c0.sub(nr().mul(c1)), v2 = c0.mul(c1);
        v0 = v0.mul(v3).add(v2);
        Fp2 r;
        r.c1 = v2.dbl();
        r.c0 = v0.add(nr().mul(v2));
        return r;
    }
    bool inverse(Fp2& out) const {                                                                       // :146-166
        if (is_zero()) return false;
        B v0 = c0.square(), v1 = c1.square();
        v0 = v0.sub(nr().mul(v1));
        B vi; v0.inverse(vi);
        out = Fp2{c0.mul(vi), c1.mul(vi).neg()};
        return true;
    }
};

// ---- fields/models/fp3.rs  (Fp3 = Fp[X]/(X^3 - NR))
template <class P, int NR> struct Fp3 {
    typedef Fp<P> B;
    B c0, c1, c2;
    static constexpr int DEG = 3;
    static Fp3 zero() { return Fp3{B::zero(), B::zero(), B::zero()}; }
    static Fp3 one() { return Fp3{B::one(), B::zero(), B::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
    bool is_one() const { return c0.is_one() && c1.is_zero() && c2.is_zero(); }
    bool operator==(const Fp3& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
    static B nr() { static const B v = fp_from_small<P>(NR); return v; }
    Fp3 add(const Fp3& o) const { return Fp3{c0.add(o.c0), c1.add(o.c1), c2.add(o.c2)}; }              // :435-442
    Fp3 sub(const Fp3& o) const { return Fp3{c0.sub(o.c0), c1.sub(o.c1), c2.sub(o.c2)}; }              // :444-451
    Fp3 dbl() const { return Fp3{c0.dbl(), c1.dbl(), c2.dbl()}; }
    Fp3 neg() const { return Fp3{c0.neg(), c1.neg(), c2.neg()}; }                                        // :373
    Fp3 mul(const Fp3& o) const {                                                                        // :453-477
        const B &a = o.c0, &b = o.c1, &c = o.c2, &d = c0, &e = c1, &f = c2;
        B ad = d.mul(a), be = e.mul(b), cf = f.mul(c);
        B x = e.add(f).mul(b.add(c)).sub(be).sub(cf);
        B y = d.add(e).mul(a.add(b)).sub(ad).sub(be);
        B z = d.add(f).mul(a.add(c)).sub(ad).add(be).sub(cf);
        return Fp3{ad.add(nr().mul(x)), y.add(nr().mul(cf)), z};
    }
    Fp3 square() const {                                                                                 // :165-185 CH-SQR2
        const B &a = c0, &b = c1, &c = c2;
        B s0 = a.square(), ab = a.mul(b), s1 = ab.dbl(), s2 = a.sub(b).add(c).square();
        B bc = b.mul(c), s3 = bc.dbl(), s4 = c.square();
        return Fp3{s0.add(nr().mul(s3)), s1.add(nr().mul(s4)), s1.add(s2).add(s3).sub(s0).sub(s4)};
    }
    bool inverse(Fp3& out) const {                                                                       // :187-219
        if (is_zero()) return false;
        B t0 = c0.square(), t1 = c1.square(), t2 = c2.square();
        B t3 = c0.mul(c1), t4 = c0.mul(c2), t5 = c1.mul(c2);
        B n5 = nr().mul(t5);
        B s0 = t0.sub(n5), s1 = nr().mul(t2).sub(t3), s2 = t1.sub(t4);
        B a1 = c2.mul(s1), a2 = c1.mul(s2);
        B a3 = nr().mul(a1.add(a2));
        B t6; c0.mul(s0).add(a3).inverse(t6);
        out = Fp3{t6.mul(s0), t6.mul(s1), t6.mul(s2)};
        return true;
    }
};

// uniform wrapper so curve code can be written once: F::T with add/sub/dbl/neg/mul/square/inverse
template <class P> struct Fp1 : Fp<P> {
    static constexpr int DEG = 1;
    Fp1() {}
    Fp1(const Fp<P>& f) : Fp<P>(f) {}
};

// ---- curves: models/short_weierstrass_projective.rs + curves/mnt{4,6}753/{g1,g2}.rs
struct Mnt4G1 {
    typedef Fp<P4> F;
    static constexpr int DEG = 1;
    static F mul_by_a(const F& z) { static const F a = fp_from_small<P4>(2); return a.mul(z); }           // COEFF_A = 2, g1.rs:20
};
struct Mnt6G1 {
    typedef Fp<P6> F;
    static constexpr int DEG = 1;
    static F mul_by_a(const F& z) { static const F a = fp_from_small<P6>(11); return a.mul(z); }          // COEFF_A = 11, g1.rs:20
};
struct Mnt4G2 {
    typedef Fp2<P4, 13> F;
    static constexpr int DEG = 2;
    static F mul_by_a(const F& z) {                                                                        // g2.rs:113-118
        static const Fp<P4> k = fp_from_small<P4>(26);  // MUL_BY_A_C0 = MUL_BY_A_C1 = NONRESIDUE * COEFF_A
        return F{k.mul(z.c0), k.mul(z.c1)};
    }
};
struct Mnt6G2 {
    typedef Fp3<P6, 11> F;
    static constexpr int DEG = 3;
    static F mul_by_a(const F& z) {                                                                        // g2.rs:149-155
        static const Fp<P6> k0 = fp_from_small<P6>(121), k2 = fp_from_small<P6>(11);
        return F{k0.mul(z.c1), k0.mul(z.c2), k2.mul(z.c0)};
    }
};

template <class C> struct Affine {
    typename C::F x, y;
    bool infinity;
    static Affine zero() { return Affine{C::F::zero(), C::F::one(), true}; }                              // swp.rs:130-132
    bool is_zero() const { return infinity; }
    Affine neg() const { return infinity ? *this : Affine{x, y.neg(), false}; }
};

template <class C> struct Projective {
    typedef typename C::F F;
    F x, y, z;
    static Projective zero() { return Projective{F::zero(), F::one(), F::zero()}; }                       // :372-378
    bool is_zero() const { return z.is_zero(); }                                                          // :388-390
    bool is_normalized() const { return is_zero() || z.is_one(); }                                        // :398-400
    static Projective from_affine(const Affine<C>& p) {                                                   // :651-659
        return p.is_zero() ? zero() : Projective{p.x, p.y, F::one()};
    }
    bool eq(const Projective& o) const {                                                                  // PartialEq :298-317
        if (is_zero()) return o.is_zero();
        if (o.is_zero()) return false;
        return x.mul(o.z) == o.x.mul(z) && y.mul(o.z) == o.y.mul(z);
    }
    Projective& double_in_place() {                                                                       // :444-479 dbl-2007-bl
        if (is_zero()) return *this;
        F xx = x.square(), zz = z.square();
        F w = C::mul_by_a(zz).add(xx.add(xx.dbl()));
        F s = y.mul(z).dbl();
        F sss = s.square().mul(s);
        F r = y.mul(s);
        F rr = r.square();
        F b = x.add(r).square().sub(xx).sub(rr);
        F h = w.square().sub(b.add(b));
        x = h.mul(s);
        y = w.mul(b.sub(h)).sub(rr.add(rr));
        z = sss;
        return *this;
    }
    void add_assign_mixed(const Affine<C>& o) {                                                           // :481-519 madd-1998-cmo
        if (o.is_zero()) return;
        if (is_zero()) { x = o.x; y = o.y; z = F::one(); return; }
        F v = o.x.mul(z), u = o.y.mul(z);
        if (u == y && v == x) { double_in_place(); return; }
        u = u.sub(y);
        F uu = u.square();
        v = v.sub(x);
        F vv = v.square(), vvv = v.mul(vv), r = vv.mul(x);
        F a = uu.mul(z).sub(vvv).sub(r.dbl());
        F nx = v.mul(a), ny = u.mul(r.sub(a)).sub(vvv.mul(y)), nz = vvv.mul(z);
        x = nx; y = ny; z = nz;
    }
    void add_assign(const Projective& o) {                                                                // :574-617 add-1998-cmo-2
        if (is_zero()) { *this = o; return; }
        if (o.is_zero()) return;
        if (eq(o)) { double_in_place(); return; }
        F y1z2 = y.mul(o.z), x1z2 = x.mul(o.z), z1z2 = z.mul(o.z);
        F u = z.mul(o.y).sub(y1z2), uu = u.square();
        F v = z.mul(o.x).sub(x1z2), vv = v.square(), vvv = v.mul(vv);
        F r = vv.mul(x1z2);
        F a = uu.mul(z1z2).sub(vvv.add(r).add(r));
        x = v.mul(a);
        y = r.sub(a).mul(u).sub(vvv.mul(y1z2));
        z = vvv.mul(z1z2);
    }
    // mul_assign :521-539 -- double-and-add over the bits of a canonical scalar, MSB first
    Projective mul_bits(const Big& k) const {
        Projective res = zero();
        bool found_one = false;
        for (int i = 64 * N - 1; i >= 0; i--) {
            bool bit = (k.l[i / 64] >> (i % 64)) & 1;
            if (found_one) res.double_in_place(); else found_one = bit;
            if (bit) res.add_assign(*this);
        }
        return res;
    }
    Affine<C> into_affine() const {                                                                       // :663-678
        if (is_zero()) return Affine<C>::zero();
        if (z.is_one()) return Affine<C>{x, y, false};
        F zi; z.inverse(zi);
        return Affine<C>{x.mul(zi), y.mul(zi), false};
    }
    // batch_normalization :402-442 (Montgomery's trick)
    static void batch_normalization(std::vector<Projective>& v) {
        std::vector<F> prod;
        prod.reserve(v.size());
        F tmp = F::one();
        for (auto& g : v) if (!g.is_normalized()) { tmp = tmp.mul(g.z); prod.push_back(tmp); }
        if (prod.empty()) return;
        F ti; tmp.inverse(ti); tmp = ti;
        size_t k = prod.size();
        for (size_t idx = v.size(); idx-- > 0;) {
            Projective& g = v[idx];
            if (g.is_normalized()) continue;
            k--;
            F s = k > 0 ? prod[k - 1] : F::one();
            F newtmp = tmp.mul(g.z);
            g.z = tmp.mul(s);
            tmp = newtmp;
        }
        for (auto& g : v) if (!g.is_normalized()) { g.x = g.x.mul(g.z); g.y = g.y.mul(g.z); g.z = F::one(); }
    }
};

// ---- a tiny fork-join helper standing in for rayon (fft/multicore.rs:7-34 Worker, rayon par_iter)
static inline void parallel_for(size_t count, int threads, const std::function<void(size_t)>& fn);

}  // namespace oracle

#include <functional>
namespace oracle {
static inline void parallel_for(size_t count, int threads, const std::function<void(size_t)>& fn) {
    if (threads <= 1 || count <= 1) { for (size_t i = 0; i < count; i++) fn(i); return; }
    std::atomic<size_t> next(0);
    std::vector<std::thread> pool;
    int nt = (int)std::min<size_t>((size_t)threads, count);
    for (int t = 0; t < nt; t++) pool.emplace_back([&] { for (;;) { size_t i = next.fetch_add(1); if (i >= count) break; fn(i); } });
    for (auto& th : pool) th.join();
}
static inline uint32_t log2_floor(size_t num) { uint32_t p = 0; while (((size_t)1 << (p + 1)) <= num) p++; return p; }  // multicore.rs:36-46

// ---- msm/variable_base.rs:10-83  VariableBaseMSM::msm_inner (literal restatement)
template <class C, class ScalarP>
Projective<C> msm_inner(const Affine<C>* bases, size_t n_bases, const Big* scalars, size_t n_scalars, int threads) {
    typedef Projective<C> G;
    size_t c;
    if (n_scalars < 32) c = 3;
    else c = (size_t)std::ceil(2.0 / 3.0 * std::log2((double)(uint32_t)n_scalars) + 2.0);                  // :14-18 (as u32!)
    const size_t num_bits = 753;                                                                           // MODULUS_BITS :20-21
    const Big fr_one = Fp<ScalarP>::one().into_repr();                                                     // :22
    const G zero = G::zero();
    std::vector<size_t> window_starts;
    for (size_t w = 0; w < num_bits; w += c) window_starts.push_back(w);                                   // :25
    const size_t n = std::min(n_bases, n_scalars);                                                         // zip :36
    std::vector<G> window_sums(window_starts.size(), zero);
    parallel_for(window_starts.size(), threads, [&](size_t wi) {                                           // :30-31 into_par_iter
        const size_t w_start = window_starts[wi];
        G res = zero;
        std::vector<G> buckets(((size_t)1 << c) - 1, zero);                                                // :35
        for (size_t i = 0; i < n; i++) {
            if (scalars[i].is_zero()) continue;                                                            // filter :36
            if (scalars[i] == fr_one) {
                if (w_start == 0) res.add_assign_mixed(bases[i]);                                          // :37-41
            } else {
                Big s = scalars[i];
                s.divn((uint32_t)w_start);                                                                 // :45-47
                uint64_t d = s.l[0] % ((uint64_t)1 << c);                                                  // :50
                if (d != 0) buckets[d - 1].add_assign_mixed(bases[i]);                                     // :55-57
            }
        }
        G::batch_normalization(buckets);                                                                   // :60
        G running = zero;
        for (size_t b = buckets.size(); b-- > 0;) {                                                        // :62-66
            running.add_assign_mixed(buckets[b].into_affine());
            res.add_assign(running);
        }
        window_sums[wi] = res;
    });
    G lowest = window_sums[0];                                                                             // :73
    G total = zero;                                                                                        // :76-82
    for (size_t wi = window_sums.size(); wi-- > 1;) {
        total.add_assign(window_sums[wi]);
        for (size_t k = 0; k < c; k++) total.double_in_place();
    }
    total.add_assign(lowest);
    return total;
}

// ---- msm/fixed_base.rs:7-79  FixedBaseMSM (literal restatement)
static inline size_t fixed_base_window_size(size_t num_scalars) {                                          // get_mul_window_size :7-13
    if (num_scalars < 32) return 3;
    return (size_t)std::ceil(std::log((double)(uint32_t)num_scalars));                                     // ln, as u32
}
template <class C>
std::vector<std::vector<Projective<C>>> fixed_base_window_table(size_t scalar_size, size_t window, Projective<C> g) {   // :15-43
    typedef Projective<C> T;
    const size_t in_window = (size_t)1 << window;
    const size_t outerc = (scalar_size + window - 1) / window;
    const size_t last_in_window = (size_t)1 << (scalar_size - (outerc - 1) * window);
    std::vector<std::vector<T>> multiples_of_g(outerc, std::vector<T>(in_window, T::zero()));
    T g_outer = g;
    for (size_t outer = 0; outer < outerc; outer++) {
        T g_inner = T::zero();
        const size_t cur_in_window = outer == outerc - 1 ? last_in_window : in_window;
        for (size_t inner = 0; inner < cur_in_window; inner++) {
            multiples_of_g[outer][inner] = g_inner;
            g_inner.add_assign(g_outer);
        }
        for (size_t k = 0; k < window; k++) g_outer.double_in_place();
    }
    return multiples_of_g;
}
template <class C, class ScalarP>
Projective<C> fixed_base_windowed_mul(size_t outerc, size_t window, const std::vector<std::vector<Projective<C>>>& table,
                                      const Fp<ScalarP>& scalar) {                                          // :45-66
    const Big repr = scalar.into_repr();                        // to_bits() reversed = little-endian bit order
    Projective<C> res = table[0][0];
    for (size_t outer = 0; outer < outerc; outer++) {
        size_t inner = 0;
        for (size_t i = 0; i < window; i++) {
            const size_t bit = outer * window + i;
            if (bit < 753 && ((repr.l[bit / 64] >> (bit % 64)) & 1)) inner |= (size_t)1 << i;               // MODULUS_BITS = 753
        }
        res.add_assign(table[outer][inner]);
    }
    return res;
}
template <class C, class ScalarP>
std::vector<Projective<C>> fixed_base_msm(size_t scalar_size, size_t window, const std::vector<std::vector<Projective<C>>>& table,
                                          const Fp<ScalarP>* v, size_t n, int threads) {                   // :68-78
    const size_t outerc = (scalar_size + window - 1) / window;
    std::vector<Projective<C>> out(n, Projective<C>::zero());
    parallel_for(n, threads, [&](size_t i) { out[i] = fixed_base_windowed_mul<C, ScalarP>(outerc, window, table, v[i]); });
    return out;
}

// ---- fft/domain.rs
template <class P> struct Domain {
    typedef Fp<P> F;
    uint64_t size; uint32_t log_size_of_group;
    F size_as_field_element, size_inv, group_gen, group_gen_inv, generator_inv;
    // new :65-94
    static bool create(size_t num_coeffs, Domain& d) {
        uint64_t size = 1; uint32_t lg = 0;
        while (size < num_coeffs) { size <<= 1; lg++; }                                                    // next_power_of_two
        if ((int)lg >= P::TWO_ADICITY) return false;                                                       // :69-71
        F g = F::root_of_unity();
        for (int i = (int)lg; i < P::TWO_ADICITY; i++) g.square_in_place();                                // :76-79
        d.size = size; d.log_size_of_group = lg;
        F::from_repr(Big::from_u64(size), d.size_as_field_element);
        d.size_as_field_element.inverse(d.size_inv);
        d.group_gen = g;
        g.inverse(d.group_gen_inv);
        F::multiplicative_generator().inverse(d.generator_inv);
        return true;
    }
};

static inline uint32_t bitreverse(uint32_t n, uint32_t l) { uint32_t r = 0; for (uint32_t i = 0; i < l; i++) { r = (r << 1) | (n & 1); n >>= 1; } return r; }

// serial_fft :315-358
template <class P> void serial_fft(Fp<P>* a, uint32_t n, Fp<P> omega, uint32_t log_n) {
    typedef Fp<P> F;
    for (uint32_t k = 0; k < n; k++) { uint32_t rk = bitreverse(k, log_n); if (k < rk) std::swap(a[rk], a[k]); }
    uint32_t m = 1;
    for (uint32_t s = 0; s < log_n; s++) {
        uint64_t e = n / (2 * m);
        F w_m = omega.pow(&e, 1);                                                                          // :338
        for (uint32_t k = 0; k < n; k += 2 * m) {
            F w = F::one();
            for (uint32_t j = 0; j < m; j++) {
                F t = a[k + j + m]; t.mul_assign(w);                                                       // :343-351
                F tmp = a[k + j]; tmp.sub_assign(t);
                a[k + j + m] = tmp;
                a[k + j].add_assign(t);
                w.mul_assign(w_m);
            }
        }
        m *= 2;
    }
}

// parallel_fft :360-416
template <class P> void parallel_fft(Fp<P>* a, uint32_t log_n, Fp<P> omega, uint32_t log_cpus, int threads) {
    typedef Fp<P> F;
    const uint32_t num_cpus = 1u << log_cpus, log_new_n = log_n - log_cpus, new_n = 1u << log_new_n, n = 1u << log_n;
    std::vector<std::vector<F>> tmp(num_cpus, std::vector<F>(new_n, F::zero()));
    uint64_t e = num_cpus;
    F new_omega = omega.pow(&e, 1);
    parallel_for(num_cpus, threads, [&](size_t j) {
        uint64_t ej = j, es = (uint64_t)j << log_new_n;
        F omega_j = omega.pow(&ej, 1), omega_step = omega.pow(&es, 1);
        F elt = F::one();
        for (uint32_t i = 0; i < new_n; i++) {
            for (uint32_t s = 0; s < num_cpus; s++) {
                uint32_t idx = (i + (s << log_new_n)) % n;
                F t = a[idx]; t.mul_assign(elt);
                tmp[j][i].add_assign(t);
                elt.mul_assign(omega_step);
            }
            elt.mul_assign(omega_j);
        }
        serial_fft<P>(tmp[j].data(), new_n, new_omega, log_new_n);
    });
    const uint32_t mask = num_cpus - 1;
    for (uint32_t idx = 0; idx < n; idx++) a[idx] = tmp[idx & mask][idx >> log_cpus];                      // :402-415
}

// best_fft :305-313  (threads plays rayon::current_num_threads())
template <class P> void best_fft(Fp<P>* a, uint32_t log_n, Fp<P> omega, int threads) {
    uint32_t log_cpus = log2_floor((size_t)(threads < 1 ? 1 : threads));
    if (log_n <= log_cpus) serial_fft<P>(a, 1u << log_n, omega, log_n);
    else parallel_fft<P>(a, log_n, omega, log_cpus, threads);
}

// distribute_powers :140-152 (chunked like Worker::scope; the values do not depend on the chunking)
template <class P> void distribute_powers(Fp<P>* a, size_t n, Fp<P> g, int threads) {
    typedef Fp<P> F;
    size_t cpus = threads < 1 ? 1 : threads;
    size_t chunk = n < cpus ? 1 : n / cpus;
    size_t nchunks = (n + chunk - 1) / chunk;
    parallel_for(nchunks, threads, [&](size_t ci) {
        uint64_t e = ci * chunk;
        F u = g.pow(&e, 1);
        for (size_t i = ci * chunk; i < std::min(n, (ci + 1) * chunk); i++) { a[i].mul_assign(u); u.mul_assign(g); }
    });
}

// fft_in_place / ifft_in_place / coset_* :113-179; `a` already resized to domain size by the caller
// except for coset_fft, whose scaling applies to the unpadded prefix n_in (:163-165).
template <class P> void domain_transform(const Domain<P>& d, Fp<P>* a, size_t n_in, bool inverse, bool coset, int threads) {
    typedef Fp<P> F;
    const size_t n = d.size;
    if (coset && !inverse) distribute_powers<P>(a, std::min(n_in, n), F::multiplicative_generator(), threads);
    best_fft<P>(a, d.log_size_of_group, inverse ? d.group_gen_inv : d.group_gen, threads);
    if (inverse) {
        parallel_for((n + 4095) / 4096, threads, [&](size_t ci) { for (size_t i = ci * 4096; i < std::min(n, (ci + 1) * 4096); i++) a[i].mul_assign(d.size_inv); });  // :137
        if (coset) distribute_powers<P>(a, n, d.generator_inv, threads);                                   // :176-178
    }
}

// ---- proof-systems/src/groth16/r1cs_to_qap.rs:121-166  witness_map, from the evaluated rows on
// (a, b, c: domain-size vectors, already holding the constraint evaluations and the input rows).
template <class P>
void witness_map(std::vector<Fp<P>>& a, std::vector<Fp<P>>& b, std::vector<Fp<P>>& c, uint32_t log_n,
                 const Fp<P>& d1, const Fp<P>& d2, const Fp<P>& d3, std::vector<Fp<P>>& h, int threads) {
    typedef Fp<P> F;
    Domain<P> d;
    Domain<P>::create((size_t)1 << log_n, d);
    const size_t n = d.size;
    domain_transform<P>(d, a.data(), n, true, false, threads);                    // ifft_in_place(a)  :121
    domain_transform<P>(d, b.data(), n, true, false, threads);                    // ifft_in_place(b)  :122
    h.assign(n, F::zero());                                                       // :124
    for (size_t i = 0; i < n; i++) { F t = d2.mul(a[i]).add(d1.mul(b[i])); h[i].mul_assign(t); }   // :125-128 (zero *= ...)
    h[0].sub_assign(d3);                                                          // :129
    F d1d2 = d1.mul(d2);                                                          // :130
    h[0].sub_assign(d1d2);                                                        // :131
    h.push_back(d1d2);                                                            // :132
    domain_transform<P>(d, a.data(), n, false, true, threads);                    // coset_fft(a) :134
    domain_transform<P>(d, b.data(), n, false, true, threads);                    // coset_fft(b) :135
    std::vector<F> ab(a);
    for (size_t i = 0; i < n; i++) ab[i].mul_assign(b[i]);                        // :137
    domain_transform<P>(d, c.data(), n, true, false, threads);                    // :153
    domain_transform<P>(d, c.data(), n, false, true, threads);                    // :154
    for (size_t i = 0; i < n; i++) ab[i].sub_assign(c[i]);                        // :156-158
    uint64_t e = n;                                                               // divide_by_vanishing_poly_on_coset (domain.rs:245-256)
    F vi; F::multiplicative_generator().pow(&e, 1).sub(F::one()).inverse(vi);
    for (size_t i = 0; i < n; i++) ab[i].mul_assign(vi);                          // :160
    domain_transform<P>(d, ab.data(), n, true, true, threads);                    // coset_ifft :161
    for (size_t i = 0; i + 1 < n; i++) h[i].add_assign(ab[i]);                    // :163-166
}

// ---- algebra/src/fields/mod.rs:412-442  batch_inversion (Montgomery's trick; zero elements are skipped)
template <class P> void batch_inversion(Fp<P>* v, size_t n) {
    typedef Fp<P> F;
    std::vector<F> prod;
    prod.reserve(n);
    F tmp = F::one();
    for (size_t i = 0; i < n; i++) if (!v[i].is_zero()) { tmp.mul_assign(v[i]); prod.push_back(tmp); }       // :418-423
    F ti; tmp.inverse(ti); tmp = ti;                                                                        // :426
    size_t k = prod.size();
    for (size_t i = n; i-- > 0;) {                                                                          // :429-441, backwards
        if (v[i].is_zero()) continue;
        k--;
        const F s = k > 0 ? prod[k - 1] : F::one();
        const F newtmp = tmp.mul(v[i]);
        v[i] = tmp.mul(s);
        tmp = newtmp;
    }
}

// ---- algebra/src/fft/domain.rs:183-219  evaluate_all_lagrange_coefficients
template <class P> std::vector<Fp<P>> evaluate_all_lagrange_coefficients(const Domain<P>& d, const Fp<P>& tau) {
    typedef Fp<P> F;
    const size_t size = d.size;
    uint64_t e = size;
    const F t_size = tau.pow(&e, 1);                                                                        // :186
    const F one = F::one();
    std::vector<F> u(size, F::zero());
    if (t_size == one) {                                                                                    // :188-199
        F omega_i = one;
        for (size_t i = 0; i < size; i++) {
            if (omega_i == tau) { u[i] = one; break; }
            omega_i.mul_assign(d.group_gen);
        }
        return u;
    }
    F l = t_size.sub(one).mul(d.size_inv);                                                                  // :203
    F r = one;
    std::vector<F> ls(size, F::zero());
    for (size_t i = 0; i < size; i++) {                                                                     // :207-212
        u[i] = tau.sub(r);
        ls[i] = l;
        l.mul_assign(d.group_gen);
        r.mul_assign(d.group_gen);
    }
    batch_inversion<P>(u.data(), size);                                                                     // :214
    for (size_t i = 0; i < size; i++) u[i] = ls[i].mul(u[i]);                                               // :215-217
    return u;
}

// ---- proof-systems/src/gm17/r1cs_to_sap.rs:191-240  witness_map, from the evaluated rows on
// (a, c: domain-size vectors as built at :158-190 and :207-230).
template <class P>
void sap_witness_map(std::vector<Fp<P>>& a, std::vector<Fp<P>>& c, uint32_t log_n, const Fp<P>& d1, const Fp<P>& d2,
                     std::vector<Fp<P>>& h, int threads) {
    typedef Fp<P> F;
    Domain<P> d;
    Domain<P>::create((size_t)1 << log_n, d);
    const size_t n = d.size;
    domain_transform<P>(d, a.data(), n, true, false, threads);                    // ifft_in_place(a)  :191
    const F d1_double = d1.dbl();                                                 // :193
    h.assign(n, d1_double);                                                       // :194
    for (size_t i = 0; i < n; i++) h[i].mul_assign(a[i]);                         // :195
    h[0].sub_assign(d2);                                                          // :196
    const F d1d1 = d1.square();                                                   // :197
    h[0].sub_assign(d1d1);                                                        // :198
    h.push_back(d1d1);                                                            // :199
    domain_transform<P>(d, a.data(), n, false, true, threads);                    // coset_fft(a) :201
    std::vector<F> aa(a);
    for (size_t i = 0; i < n; i++) aa[i].mul_assign(a[i]);                        // :203
    domain_transform<P>(d, c.data(), n, true, false, threads);                    // :232
    domain_transform<P>(d, c.data(), n, false, true, threads);                    // :233
    for (size_t i = 0; i < n; i++) aa[i].sub_assign(c[i]);                        // :235
    uint64_t e = n;
    F vi; F::multiplicative_generator().pow(&e, 1).sub(F::one()).inverse(vi);     // divide_by_vanishing_poly_on_coset :237
    for (size_t i = 0; i < n; i++) aa[i].mul_assign(vi);
    domain_transform<P>(d, aa.data(), n, true, true, threads);                    // coset_ifft :238
    for (size_t i = 0; i + 1 < n; i++) h[i].add_assign(aa[i]);                    // :240-243
}

}  // namespace oracle

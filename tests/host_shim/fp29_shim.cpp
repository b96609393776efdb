// Host build (g++) of the device arithmetic headers, exported for ctypes so that
// tests/test_fp29_host.py can check the exact code the HIP kernels run against Python integers.
// Test infrastructure only; not part of the product library.
#include <string.h>
#include "../../ginger-lib_amd/csrc/ec29.h"
#include "../../ginger-lib_amd/csrc/host_math.h"

using namespace gh;

template <class P> static void fp_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    Fp x = fp_unpack(a), y = fp_unpack(b), r;
    switch (op) {
        case 0: r = fp_mul<P>(x, y); break;
        case 1: r = fp_sqr<P>(x); break;
        case 2: r = fp_add<P>(x, y); break;
        case 3: r = fp_sub<P>(x, y); break;
        case 4: r = fp_neg<P>(x); break;
        case 5: r = fp_dbl<P>(x); break;
        case 6: r = fp_from_abi<P>(a); break;
        case 7: fp_to_abi<P>(out, x); return;
        case 8: r = fp_mul_small<P, 11>(x); break;
        case 9: r = fp_mul_small<P, 13>(x); break;
        case 10: r = fp_mul_small<P, 26>(x); break;
        case 11: r = fp_mul_small<P, 121>(x); break;
        case 12: r = fp_inv<P>(x); break;
        case 13: r = fp_inv_plain<P>(x); break;
        case 14: r = fp_mul_small_rt<P>(x, b[0]); break;     // multiplier (1 .. 15) in the first word of b
        default: r = fp_zero();
    }
    fp_pack(out, r);
}

// ABI layout for tower elements: DEG x 24 words, Montgomery radix 2^768.
template <class C> static void ec_op(int op, const uint32_t* p, const uint32_t* q, uint32_t* out) {
    typedef typename C::F F;
    const int W = 24 * F::DEG;
    Proj<C> a{F::from_abi(p), F::from_abi(p + W), F::from_abi(p + 2 * W)}, r;
    switch (op) {
        case 0: {  // madd: q affine (x, y)
            Aff<C> b{F::from_abi(q), F::from_abi(q + W)};
            r = proj_madd<C>(a, b);
            break;
        }
        case 1: {
            Proj<C> b{F::from_abi(q), F::from_abi(q + W), F::from_abi(q + 2 * W)};
            r = proj_add<C>(a, b);
            break;
        }
        case 2: r = proj_dbl<C>(a); break;
        case 3: {  // field mul of the x coordinates (tower check)
            r = a;
            r.x = F::mul(a.x, F::from_abi(q));
            break;
        }
        case 4: r = a; r.x = F::sqr(a.x); break;
        default: r = proj_zero<C>();
    }
    F::to_abi(out, r.x);
    F::to_abi(out + W, r.y);
    F::to_abi(out + 2 * W, r.z);
}

template <class P> static void h64_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    typedef HF1<P> F;
    H64 x, y, r;
    memcpy(x.l, a, 96); memcpy(y.l, b, 96);
    switch (op) {
        case 0: r = F::mul(x, y); break;
        case 2: r = F::add(x, y); break;
        case 3: r = F::sub(x, y); break;
        case 4: r = F::neg(x); break;
        case 5: r = F::dbl(x); break;
        case 8: r = F::mul_small(x, 11); break;
        default: r = F::zero();
    }
    memcpy(out, r.l, 96);
}
// G1 doubling / addition on the fast host field (ABI Montgomery limbs in and out)
template <class HC> static void h64_ec(int op, const uint64_t* p, const uint64_t* q, uint64_t* out) {
    Proj<HC> a, b, r;
    memcpy(&a, p, sizeof a); memcpy(&b, q, sizeof b);
    r = op == 0 ? proj_add<HC>(a, b) : proj_dbl<HC>(a);
    memcpy(out, &r, sizeof r);
}

template <class P> static void fp_mul2_op(const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* out) {
    fp_pack(out, fp_mul2<P>(fp_unpack(a), fp_unpack(b), fp_unpack(c), fp_unpack(d)));
}
template <class P> static void fp_mul2s_op(const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* out) {
    fp_pack(out, fp_mul2s<P>(fp_unpack(a), fp_unpack(b), fp_unpack(c), fp_unpack(d)));
}
// XYZZ mixed addition (ec29.h): p = (X, Y, ZZ, ZZZ) and q = (x, y) in ABI Montgomery words; out = the sum's four coordinates
// followed by its projective image (xyzz_to_proj); returns the `same` flag
template <class C> static int xyzz_op(const uint32_t* p, const uint32_t* q, uint32_t* out) {
    typedef typename C::F F;
    const int W = 24 * F::DEG;
    Xyzz<C> a{F::from_abi(p), F::from_abi(p + W), F::from_abi(p + 2 * W), F::from_abi(p + 3 * W)};
    Aff<C> b{F::from_abi(q), F::from_abi(q + W)};
    bool same = false;
    const Xyzz<C> r = xyzz_madd<C>(a, b, same);
    F::to_abi(out, r.x); F::to_abi(out + W, r.y); F::to_abi(out + 2 * W, r.zz); F::to_abi(out + 3 * W, r.zzz);
    const Proj<C> h = xyzz_to_proj<C>(r);
    F::to_abi(out + 4 * W, h.x); F::to_abi(out + 5 * W, h.y); F::to_abi(out + 6 * W, h.z);
    return same ? 1 : 0;
}
template <class P> static void fp_mul3_op(const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, const uint32_t* e,
                                           const uint32_t* f, uint32_t* out) {
    fp_pack(out, fp_mul3<P>(fp_unpack(a), fp_unpack(b), fp_unpack(c), fp_unpack(d), fp_unpack(e), fp_unpack(f)));
}
extern "C" {
void t_fp_mul3(int field, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, const uint32_t* e,
               const uint32_t* f, uint32_t* out) {
    if (field == 4) fp_mul3_op<P4>(a, b, c, d, e, f, out); else fp_mul3_op<P6>(a, b, c, d, e, f, out);
}
void t_fp_mul2s(int field, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* out) {
    if (field == 4) fp_mul2s_op<P4>(a, b, c, d, out); else fp_mul2s_op<P6>(a, b, c, d, out);
}
int t_xyzz_madd(int curve, const uint32_t* p, const uint32_t* q, uint32_t* out) {
    switch (curve) {
        case 0: return xyzz_op<Mnt4G1>(p, q, out);
        case 1: return xyzz_op<Mnt4G2>(p, q, out);
        case 2: return xyzz_op<Mnt6G1>(p, q, out);
        case 3: return xyzz_op<Mnt6G2>(p, q, out);
    }
    return -1;
}
void t_fp_mul2(int field, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* out) {
    if (field == 4) fp_mul2_op<P4>(a, b, c, d, out); else fp_mul2_op<P6>(a, b, c, d, out);
}
void t_h64_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    if (field == 4) h64_op<P4>(op, a, b, out); else h64_op<P6>(op, a, b, out);
}
void t_h64_ec(int field, int op, const uint64_t* p, const uint64_t* q, uint64_t* out) {
    if (field == 4) h64_ec<HostMnt4G1>(op, p, q, out); else h64_ec<HostMnt6G1>(op, p, q, out);
}
// curve ids as in ginger_hip.h (0..3): add / double on the host fold fields, incl. the G2 towers
void t_h64_ec_curve(int curve, int op, const uint64_t* p, const uint64_t* q, uint64_t* out) {
    switch (curve) {
        case 0: h64_ec<HostMnt4G1>(op, p, q, out); break;
        case 1: h64_ec<HostMnt4G2>(op, p, q, out); break;
        case 2: h64_ec<HostMnt6G1>(op, p, q, out); break;
        case 3: h64_ec<HostMnt6G2>(op, p, q, out); break;
    }
}
void t_fp_op(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    if (field == 4) fp_op<P4>(op, a, b, out); else fp_op<P6>(op, a, b, out);
}
// curve: 0 mnt4753_g1, 1 mnt4753_g2, 2 mnt6753_g1, 3 mnt6753_g2
void t_ec_op(int curve, int op, const uint32_t* p, const uint32_t* q, uint32_t* out) {
    switch (curve) {
        case 0: ec_op<Mnt4G1>(op, p, q, out); break;
        case 1: ec_op<Mnt4G2>(op, p, q, out); break;
        case 2: ec_op<Mnt6G1>(op, p, q, out); break;
        case 3: ec_op<Mnt6G2>(op, p, q, out); break;
    }
}
}

// oracle_capi.cpp -- extern "C" surface of the CPU oracle (see oracle.hpp; TEST INFRASTRUCTURE ONLY).
// Data formats are the same as include/ginger_hip.h so that tests can feed identical buffers to
// both sides: Montgomery (R = 2^768) 12-u64 field elements, canonical 12-u64 scalars.
#include "oracle.hpp"

using namespace oracle;

namespace {

template <class F> struct Ser;
template <class P> struct Ser<Fp<P>> {
    static void rd(Fp<P>& f, const uint64_t* p) { memcpy(f.v.l, p, 96); }
    static void wr(uint64_t* p, const Fp<P>& f) { memcpy(p, f.v.l, 96); }
};
template <class P, int NR> struct Ser<Fp2<P, NR>> {
    static void rd(Fp2<P, NR>& f, const uint64_t* p) { memcpy(f.c0.v.l, p, 96); memcpy(f.c1.v.l, p + 12, 96); }
    static void wr(uint64_t* p, const Fp2<P, NR>& f) { memcpy(p, f.c0.v.l, 96); memcpy(p + 12, f.c1.v.l, 96); }
};
template <class P, int NR> struct Ser<Fp3<P, NR>> {
    static void rd(Fp3<P, NR>& f, const uint64_t* p) { memcpy(f.c0.v.l, p, 96); memcpy(f.c1.v.l, p + 12, 96); memcpy(f.c2.v.l, p + 24, 96); }
    static void wr(uint64_t* p, const Fp3<P, NR>& f) { memcpy(p, f.c0.v.l, 96); memcpy(p + 12, f.c1.v.l, 96); memcpy(p + 24, f.c2.v.l, 96); }
};

template <class C> Projective<C> rd_proj(const uint64_t* p) {
    typedef typename C::F F;
    Projective<C> g;
    Ser<F>::rd(g.x, p); Ser<F>::rd(g.y, p + 12 * C::DEG); Ser<F>::rd(g.z, p + 24 * C::DEG);
    return g;
}
template <class C> void wr_proj(uint64_t* p, const Projective<C>& g) {
    typedef typename C::F F;
    Ser<F>::wr(p, g.x); Ser<F>::wr(p + 12 * C::DEG, g.y); Ser<F>::wr(p + 24 * C::DEG, g.z);
}
template <class C> Affine<C> rd_aff(const uint64_t* p, bool inf) {
    typedef typename C::F F;
    if (inf) return Affine<C>::zero();
    Affine<C> a; a.infinity = false;
    Ser<F>::rd(a.x, p); Ser<F>::rd(a.y, p + 12 * C::DEG);
    return a;
}
template <class C> void wr_aff(uint64_t* p, uint8_t* inf, const Affine<C>& a) {
    typedef typename C::F F;
    Ser<F>::wr(p, a.x); Ser<F>::wr(p + 12 * C::DEG, a.y);
    *inf = a.infinity ? 1 : 0;
}

template <class C, class SP>
int msm_c(const uint64_t* bases, const uint8_t* infinity, size_t n_bases, const uint64_t* scalars, size_t n_scalars,
          uint64_t* out_xyz, int threads) {
    std::vector<Affine<C>> b(n_bases);
    for (size_t i = 0; i < n_bases; i++) b[i] = rd_aff<C>(bases + i * 24 * C::DEG, infinity && infinity[i]);
    std::vector<Big> s(n_scalars);
    for (size_t i = 0; i < n_scalars; i++) memcpy(s[i].l, scalars + i * 12, 96);
    Projective<C> r = msm_inner<C, SP>(b.data(), n_bases, s.data(), n_scalars, threads);
    wr_proj<C>(out_xyz, r);
    return 0;
}

// FixedBaseMSM::multi_scalar_mul over the window table of g (fixed_base.rs): scalars are Montgomery field elements
// (T::ScalarField), out = n projective points; window == 0: get_mul_window_size(n).  Returns the window used.
template <class C, class SP>
int fixed_c(const uint64_t* g_xyz, size_t scalar_size, size_t window, const uint64_t* scalars, size_t n, uint64_t* out_xyz, int threads) {
    if (window == 0) window = fixed_base_window_size(n);
    Projective<C> g = rd_proj<C>(g_xyz);
    auto table = fixed_base_window_table<C>(scalar_size, window, g);
    auto res = fixed_base_msm<C, SP>(scalar_size, window, table, reinterpret_cast<const Fp<SP>*>(scalars), n, threads);
    for (size_t i = 0; i < n; i++) wr_proj<C>(out_xyz + i * 36 * C::DEG, res[i]);
    return (int)window;
}

// op: 0 add (proj+proj) 1 double 2 mixed add (q affine, q_inf flag in `flag`) 3 scalar mul by canonical k (in q)
//     4 into_affine (out = x||y, returns infinity flag) 5 projective equality (returns 0/1)
template <class C> int ec_c(int op, const uint64_t* p, const uint64_t* q, int flag, uint64_t* out) {
    Projective<C> a = rd_proj<C>(p);
    switch (op) {
        case 0: { Projective<C> b = rd_proj<C>(q); a.add_assign(b); wr_proj<C>(out, a); return 0; }
        case 1: a.double_in_place(); wr_proj<C>(out, a); return 0;
        case 2: { Affine<C> b = rd_aff<C>(q, flag != 0); a.add_assign_mixed(b); wr_proj<C>(out, a); return 0; }
        case 3: { Big k; memcpy(k.l, q, 96); wr_proj<C>(out, a.mul_bits(k)); return 0; }
        case 4: { uint8_t inf; wr_aff<C>(out, &inf, a.into_affine()); return inf; }
        case 5: { Projective<C> b = rd_proj<C>(q); return a.eq(b) ? 1 : 0; }
    }
    return -1;
}

// op: 0 mul 1 square 2 add 3 sub 4 neg 5 double 6 inverse (returns 0 if not invertible)
template <class F> int ext_c(int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    F x, y, r;
    Ser<F>::rd(x, a);
    if (b) Ser<F>::rd(y, b);
    switch (op) {
        case 0: r = x.mul(y); break;
        case 1: r = x.square(); break;
        case 2: r = x.add(y); break;
        case 3: r = x.sub(y); break;
        case 4: r = x.neg(); break;
        case 5: r = x.dbl(); break;
        case 6: if (!x.inverse(r)) return 0; break;
        default: return -1;
    }
    Ser<F>::wr(out, r);
    return 1;
}

template <class P> int fft_c(uint64_t* data, size_t n_in, uint32_t log_n, uint32_t flags, int threads) {
    Domain<P> d;
    if (!Domain<P>::create((size_t)1 << log_n, d) || d.log_size_of_group != log_n) return -2;
    // data holds 2^log_n elements; the caller has already resized (zero padded / truncated)
    domain_transform<P>(d, reinterpret_cast<Fp<P>*>(data), n_in, flags & 1, flags & 2, threads);
    return 0;
}

}  // namespace

extern "C" {

// field: 4 = p4 (MNT4 Fq / MNT6 Fr), 6 = p6 (MNT6 Fq / MNT4 Fr).  Raw Montgomery limbs in/out.
// op: 0 mul 1 square 2 add 3 sub 4 neg 5 double 6 inverse 7 from_repr (canonical -> Montgomery) 8 into_repr
int oracle_fp_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    if (op <= 6) return field == 4 ? ext_c<Fp<P4>>(op, a, b, out) : ext_c<Fp<P6>>(op, a, b, out);
    Big x; memcpy(x.l, a, 96);
    if (op == 7) {
        if (field == 4) { Fp<P4> r; if (!Fp<P4>::from_repr(x, r)) return 0; memcpy(out, r.v.l, 96); }
        else { Fp<P6> r; if (!Fp<P6>::from_repr(x, r)) return 0; memcpy(out, r.v.l, 96); }
        return 1;
    }
    if (op == 8) {
        Big r;
        if (field == 4) { Fp<P4> f; f.v = x; r = f.into_repr(); } else { Fp<P6> f; f.v = x; r = f.into_repr(); }
        memcpy(out, r.l, 96);
        return 1;
    }
    return -1;
}
// tower: 2 = Fq2 over p4 (MNT4 G2), 3 = Fq3 over p6 (MNT6 G2)
int oracle_ext_op(int tower, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    return tower == 2 ? ext_c<Fp2<P4, 13>>(op, a, b, out) : ext_c<Fp3<P6, 11>>(op, a, b, out);
}
// curve ids as in ginger_hip.h: 0 mnt4753_g1, 1 mnt4753_g2, 2 mnt6753_g1, 3 mnt6753_g2
int oracle_ec_op(int curve, int op, const uint64_t* p, const uint64_t* q, int flag, uint64_t* out) {
    switch (curve) {
        case 0: return ec_c<Mnt4G1>(op, p, q, flag, out);
        case 1: return ec_c<Mnt4G2>(op, p, q, flag, out);
        case 2: return ec_c<Mnt6G1>(op, p, q, flag, out);
        case 3: return ec_c<Mnt6G2>(op, p, q, flag, out);
    }
    return -1;
}
int oracle_msm(int curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases, const uint64_t* scalars,
               size_t n_scalars, uint64_t* out_xyz, int threads) {
    switch (curve) {  // scalar field of MNT4 is p6 and vice versa
        case 0: return msm_c<Mnt4G1, P6>(bases, infinity, n_bases, scalars, n_scalars, out_xyz, threads);
        case 1: return msm_c<Mnt4G2, P6>(bases, infinity, n_bases, scalars, n_scalars, out_xyz, threads);
        case 2: return msm_c<Mnt6G1, P4>(bases, infinity, n_bases, scalars, n_scalars, out_xyz, threads);
        case 3: return msm_c<Mnt6G2, P4>(bases, infinity, n_bases, scalars, n_scalars, out_xyz, threads);
    }
    return -1;
}
int oracle_fixed_base_msm(int curve, const uint64_t* g_xyz, size_t scalar_size, size_t window, const uint64_t* scalars, size_t n,
                          uint64_t* out_xyz, int threads) {
    switch (curve) {
        case 0: return fixed_c<Mnt4G1, P6>(g_xyz, scalar_size, window, scalars, n, out_xyz, threads);
        case 1: return fixed_c<Mnt4G2, P6>(g_xyz, scalar_size, window, scalars, n, out_xyz, threads);
        case 2: return fixed_c<Mnt6G1, P4>(g_xyz, scalar_size, window, scalars, n, out_xyz, threads);
        case 3: return fixed_c<Mnt6G2, P4>(g_xyz, scalar_size, window, scalars, n, out_xyz, threads);
    }
    return -1;
}
// field ids as in ginger_hip.h: 0 = MNT4-753 Fr (p6), 1 = MNT6-753 Fr (p4).  In place on 2^log_n elements.
// flags: 1 inverse, 2 coset.  n_in = number of leading elements that were real input (coset_fft scaling).
int oracle_fft(int field, uint64_t* data, size_t n_in, uint32_t log_n, uint32_t flags, int threads) {
    return field == 0 ? fft_c<P6>(data, n_in, log_n, flags, threads) : fft_c<P4>(data, n_in, log_n, flags, threads);
}
// serial_fft / parallel_fft individually (fft/test.rs:45-72 compares them)
int oracle_fft_variant(int field, uint64_t* data, uint32_t log_n, int parallel_log_cpus) {
    if (field == 0) {
        Domain<P6> d; if (!Domain<P6>::create((size_t)1 << log_n, d)) return -2;
        Fp<P6>* a = reinterpret_cast<Fp<P6>*>(data);
        if (parallel_log_cpus < 0) serial_fft<P6>(a, 1u << log_n, d.group_gen, log_n);
        else parallel_fft<P6>(a, log_n, d.group_gen, (uint32_t)parallel_log_cpus, 1 << parallel_log_cpus);
    } else {
        Domain<P4> d; if (!Domain<P4>::create((size_t)1 << log_n, d)) return -2;
        Fp<P4>* a = reinterpret_cast<Fp<P4>*>(data);
        if (parallel_log_cpus < 0) serial_fft<P4>(a, 1u << log_n, d.group_gen, log_n);
        else parallel_fft<P4>(a, log_n, d.group_gen, (uint32_t)parallel_log_cpus, 1 << parallel_log_cpus);
    }
    return 0;
}
// domain constants: out = size_inv || group_gen || group_gen_inv || generator_inv (4 x 12 u64); returns 0 if None
int oracle_domain(int field, size_t num_coeffs, uint64_t* out, uint32_t* log_n) {
    if (field == 0) {
        Domain<P6> d; if (!Domain<P6>::create(num_coeffs, d)) return 0;
        memcpy(out, d.size_inv.v.l, 96); memcpy(out + 12, d.group_gen.v.l, 96); memcpy(out + 24, d.group_gen_inv.v.l, 96); memcpy(out + 36, d.generator_inv.v.l, 96);
        *log_n = d.log_size_of_group;
    } else {
        Domain<P4> d; if (!Domain<P4>::create(num_coeffs, d)) return 0;
        memcpy(out, d.size_inv.v.l, 96); memcpy(out + 12, d.group_gen.v.l, 96); memcpy(out + 24, d.group_gen_inv.v.l, 96); memcpy(out + 36, d.generator_inv.v.l, 96);
        *log_n = d.log_size_of_group;
    }
    return 1;
}
// pointwise product (mul_polynomials_in_evaluation_domain, domain.rs:289-302) and scaling (:245-256)
int oracle_vec_mul(int field, uint64_t* a, const uint64_t* b, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (field == 0) { Fp<P6> x, y; memcpy(x.v.l, a + 12 * i, 96); memcpy(y.v.l, b + 12 * i, 96); x.mul_assign(y); memcpy(a + 12 * i, x.v.l, 96); }
        else { Fp<P4> x, y; memcpy(x.v.l, a + 12 * i, 96); memcpy(y.v.l, b + 12 * i, 96); x.mul_assign(y); memcpy(a + 12 * i, x.v.l, 96); }
    }
    return 0;
}
// i = (g^size - 1)^-1 : evaluate_vanishing_polynomial(g).inverse()  (domain.rs:229-231, :246-249)
int oracle_vanishing_inv_on_coset(int field, uint32_t log_n, uint64_t* out) {
    uint64_t e = (uint64_t)1 << log_n;
    if (field == 0) { Fp<P6> g = Fp<P6>::multiplicative_generator().pow(&e, 1).sub(Fp<P6>::one()), r; g.inverse(r); memcpy(out, r.v.l, 96); }
    else { Fp<P4> g = Fp<P4>::multiplicative_generator().pow(&e, 1).sub(Fp<P4>::one()), r; g.inverse(r); memcpy(out, r.v.l, 96); }
    return 0;
}
// witness_map (r1cs_to_qap.rs:121-166) from evaluated rows; h_out receives 2^log_n + 1 elements
int oracle_witness_map(int field, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint32_t log_n,
                       const uint64_t* d1, const uint64_t* d2, const uint64_t* d3, uint64_t* h_out, int threads) {
    const size_t n = (size_t)1 << log_n;
    if (field == 0) {
        std::vector<Fp<P6>> va(n), vb(n), vc(n), h;
        memcpy(va.data(), a, n * 96); memcpy(vb.data(), b, n * 96); memcpy(vc.data(), c, n * 96);
        Fp<P6> f1, f2, f3; memcpy(f1.v.l, d1, 96); memcpy(f2.v.l, d2, 96); memcpy(f3.v.l, d3, 96);
        witness_map<P6>(va, vb, vc, log_n, f1, f2, f3, h, threads);
        memcpy(h_out, h.data(), (n + 1) * 96);
    } else {
        std::vector<Fp<P4>> va(n), vb(n), vc(n), h;
        memcpy(va.data(), a, n * 96); memcpy(vb.data(), b, n * 96); memcpy(vc.data(), c, n * 96);
        Fp<P4> f1, f2, f3; memcpy(f1.v.l, d1, 96); memcpy(f2.v.l, d2, 96); memcpy(f3.v.l, d3, 96);
        witness_map<P4>(va, vb, vc, log_n, f1, f2, f3, h, threads);
        memcpy(h_out, h.data(), (n + 1) * 96);
    }
    return 0;
}
// batch_inversion (fields/mod.rs:412-442) in place on n Montgomery elements
int oracle_batch_inversion(int field, uint64_t* a, size_t n) {
    if (field == 0) batch_inversion<P6>(reinterpret_cast<Fp<P6>*>(a), n);
    else batch_inversion<P4>(reinterpret_cast<Fp<P4>*>(a), n);
    return 0;
}
// evaluate_all_lagrange_coefficients (domain.rs:183-219): out receives 2^log_n elements
int oracle_lagrange(int field, uint32_t log_n, const uint64_t* tau, uint64_t* out) {
    const size_t n = (size_t)1 << log_n;
    if (field == 0) {
        Domain<P6> d; if (!Domain<P6>::create(n, d)) return -2;
        Fp<P6> t; memcpy(t.v.l, tau, 96);
        auto u = evaluate_all_lagrange_coefficients<P6>(d, t);
        memcpy(out, u.data(), n * 96);
    } else {
        Domain<P4> d; if (!Domain<P4>::create(n, d)) return -2;
        Fp<P4> t; memcpy(t.v.l, tau, 96);
        auto u = evaluate_all_lagrange_coefficients<P4>(d, t);
        memcpy(out, u.data(), n * 96);
    }
    return 0;
}
// R1CStoSAP::witness_map (gm17/r1cs_to_sap.rs:191-240) from evaluated rows; h_out receives 2^log_n + 1 elements
int oracle_sap_witness_map(int field, const uint64_t* a, const uint64_t* c, uint32_t log_n, const uint64_t* d1, const uint64_t* d2,
                           uint64_t* h_out, int threads) {
    const size_t n = (size_t)1 << log_n;
    if (field == 0) {
        std::vector<Fp<P6>> va(n), vc(n), h;
        memcpy(va.data(), a, n * 96); memcpy(vc.data(), c, n * 96);
        Fp<P6> f1, f2; memcpy(f1.v.l, d1, 96); memcpy(f2.v.l, d2, 96);
        sap_witness_map<P6>(va, vc, log_n, f1, f2, h, threads);
        memcpy(h_out, h.data(), (n + 1) * 96);
    } else {
        std::vector<Fp<P4>> va(n), vc(n), h;
        memcpy(va.data(), a, n * 96); memcpy(vc.data(), c, n * 96);
        Fp<P4> f1, f2; memcpy(f1.v.l, d1, 96); memcpy(f2.v.l, d2, 96);
        sap_witness_map<P4>(va, vc, log_n, f1, f2, h, threads);
        memcpy(h_out, h.data(), (n + 1) * 96);
    }
    return 0;
}
}

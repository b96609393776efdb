// ntt.hip -- evaluation-domain tables and launch sequence of the transforms (ntt_kernels.h).
#include "runtime.h"
#include "ntt_kernels.h"
#include "host_math.h"
#include "asm_kernels.h"

namespace gh_rt {
using namespace gh;

template <class P> struct FieldConsts;
template <> struct FieldConsts<P6> {  // MNT4-753 Fr
    static constexpr int two_adicity = GH_P6_TWO_ADICITY;
    static const uint64_t* root_m() { static const uint64_t v[12] = GH_P6_ROOT_M_64; return v; }
    static const uint64_t* gen_m() { static const uint64_t v[12] = GH_P6_GEN17_M_64; return v; }
};
template <> struct FieldConsts<P4> {  // MNT6-753 Fr
    static constexpr int two_adicity = GH_P4_TWO_ADICITY;
    static const uint64_t* root_m() { static const uint64_t v[12] = GH_P4_ROOT_M_64; return v; }
    static const uint64_t* gen_m() { static const uint64_t v[12] = GH_P4_GEN17_M_64; return v; }
};

template <class P> int build_pow_table(Fp* tab, int log_n, const Fp& first, Fp base) {
    HIPCHK(hipMemcpyAsync(tab, &first, sizeof(Fp), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    for (int s = 0; s < log_n; s++) {
        uint32_t half = 1u << s;
        hipLaunchKernelGGL((pow_table_step_kernel<P>), dim3((half + 255) / 256), dim3(256), 0, g.stream, tab, half, base);
        base = fp_sqr<P>(base);
    }
    HIPCHK(hipGetLastError());
    return GH_OK;
}

// Tables are built into locals and committed to the Domain only after every allocation and launch has
// succeeded: a failed build (e.g. out of memory next to a 100 GB shift table) leaves no half-initialised
// entry behind for the next call to trip over.
template <class P> int build_table_checked(Fp** slot, int log_n, const Fp& first, const Fp& base) {
    const size_t N = (size_t)1 << log_n;
    Fp* t = nullptr;
    HIPCHK(hipMalloc((void**)&t, N * sizeof(Fp)));
    int rc = build_pow_table<P>(t, log_n, first, base);
    if (!rc && hipStreamSynchronize(g.stream) != hipSuccess) { g_err = "building a domain table failed"; rc = GH_E_HIP; }
    if (rc) { hipFree(t); return rc; }
    *slot = t;
    return GH_OK;
}

template <class P> int get_domain(int fidx, int log_n, bool need_coset, bool need_coset_inv, Domain** out) {
    auto it = g.domains[fidx].find(log_n);
    const size_t N = (size_t)1 << log_n;
    if (it == g.domains[fidx].end()) {
        Domain d;
        d.log_n = log_n;
        // group_gen = ROOT_OF_UNITY^(2^(s - log_n))   (domain.rs:76-79)
        Fp w = fp_from_abi<P>(reinterpret_cast<const uint32_t*>(FieldConsts<P>::root_m()));
        for (int i = log_n; i < FieldConsts<P>::two_adicity; i++) w = fp_sqr<P>(w);
        // size_inv = (N as field element)^-1, internal form
        Fp n_int = fp_one<P>();
        for (int i = 0; i < log_n; i++) n_int = fp_dbl<P>(n_int);
        d.size_inv = host_fp_inv<P>(n_int);
        hipError_t e = hipMalloc((void**)&d.scratch, N * 96);
        if (e != hipSuccess) { g_err = std::string("domain scratch: ") + hipGetErrorString(e); return e == hipErrorOutOfMemory ? GH_E_NOMEM : GH_E_HIP; }
        int rc = build_table_checked<P>(&d.tw, log_n, fp_one<P>(), w);
        if (rc) { hipFree(d.scratch); return rc; }
        if (hipMalloc((void**)&d.d_size_inv, 128) != hipSuccess ||
            hipMemcpy(d.d_size_inv, &d.size_inv, sizeof(Fp), hipMemcpyHostToDevice) != hipSuccess) {
            g_err = "domain constants: device allocation failed";
            hipFree(d.scratch); hipFree(d.tw); if (d.d_size_inv) hipFree(d.d_size_inv);
            return GH_E_NOMEM;
        }
        it = g.domains[fidx].emplace(log_n, d).first;
    }
    Domain& d = it->second;
    if (need_coset && !d.coset) {
        Fp gen = fp_from_abi<P>(reinterpret_cast<const uint32_t*>(FieldConsts<P>::gen_m()));
        int rc = build_table_checked<P>(&d.coset, log_n, fp_one<P>(), gen);
        if (rc) return rc;
    }
    if (need_coset_inv && !d.coset_inv) {
        Fp gen = fp_from_abi<P>(reinterpret_cast<const uint32_t*>(FieldConsts<P>::gen_m()));
        Fp gi = host_fp_inv<P>(gen);
        int rc = build_table_checked<P>(&d.coset_inv, log_n, d.size_inv, gi);
        if (rc) return rc;
    }
    *out = &d;
    return GH_OK;
}

// sync = false: the caller (witness_map) issues several transforms and pointwise kernels back to back on the library
// stream and waits once at the end
template <class P> int fft_run(int fidx, void* d_data, uint32_t log_n, uint32_t flags, bool sync = true) {
    if ((int)log_n >= FieldConsts<P>::two_adicity) {
        g_err = "domain exceeds the field's 2-adicity";
        return GH_E_UNSUPPORTED;
    }
    g.last_fft_ms = 0;
    if (log_n == 0) return GH_OK;  // size-1 domain: identity (size_inv = 1, g^0 = 1)
    const bool inverse = flags & GH_FFT_INVERSE, coset = flags & GH_FFT_COSET;
    Domain* d;
    int rc = get_domain<P>(fidx, (int)log_n, coset && !inverse, coset && inverse, &d);
    if (rc) return rc;
    const int P_ = ((int)log_n + NTT_MAX_LOGR - 1) / NTT_MAX_LOGR;
    int ks[8];
    {
        int base = (int)log_n / P_, rem = (int)log_n % P_;
        for (int s = 0; s < P_; s++) ks[s] = base + (s < rem ? 1 : 0);
    }
    static bool attr_set[2] = {false, false};
    if (!attr_set[fidx]) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_kernel<P>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, NL * 4 * NTT_MAX_TILE));
        attr_set[fidx] = true;
    }
    // Ping-pong: with an even number of passes the caller's vector and the domain's scratch alternate; with an odd number (2^24:
    // three) a second scratch vector lets the last pass write into the caller's vector instead of a 3.2 GB copy at the end.
    if (P_ >= 3 && (P_ & 1) && !d->scratch2 && !d->scratch2_failed) {
        if (hipMalloc((void**)&d->scratch2, ((size_t)96) << log_n) != hipSuccess) {
            (void)hipGetLastError();
            d->scratch2 = nullptr;
            d->scratch2_failed = true;
        }
    }
    const bool three = P_ >= 3 && (P_ & 1) && d->scratch2;
    uint32_t* bufs[2] = {(uint32_t*)d_data, d->scratch};
    int cur = 0, log_ns = 0;
    static const bool ntt_asm = gh_asm::ntt_enabled();
    HIPCHK(hipEventRecord(g.ev[4], g.stream));
    for (int s = 0; s < P_; s++) {
        NttPassArgs A;
        A.in = bufs[cur];
        A.out = bufs[cur ^ 1];
        if (three) {          // data -> scratch -> scratch2 -> data, then the usual alternation data <-> scratch
            if (s == 0) { A.in = (uint32_t*)d_data; A.out = d->scratch; }
            else if (s == 1) { A.in = d->scratch; A.out = d->scratch2; }
            else if (s == 2) { A.in = d->scratch2; A.out = (uint32_t*)d_data; }
        }
        A.tw = d->tw;
        A.pre = (s == 0 && coset && !inverse) ? d->coset : nullptr;
        A.post = nullptr;
        A.has_post_scalar = 0;
        A.post_scalar = d->size_inv;
        if (s == P_ - 1 && inverse) {
            if (coset) A.post = d->coset_inv; else A.has_post_scalar = 1;
        }
        A.log_n = (int)log_n;
        A.k = ks[s];
        A.log_ns = log_ns;
        int log_c = (int)log_n - ks[s];
        if (log_c > NTT_LOG_TILE - ks[s]) log_c = NTT_LOG_TILE - ks[s];
        A.log_c = log_c;
        A.inverse = inverse ? 1 : 0;
        const int E = 1 << (ks[s] + log_c);
        int threads = E / 2;
        if (threads < 64) threads = 64;
        const unsigned grid = 1u << ((int)log_n - ks[s] - log_c);
        if (ntt_asm && gh_asm::ntt_supported((int)log_n, ks[s])) {
            // the assembly pass (asmgen/ntt_pass.py): same indices and factors, a wave per 256 elements
            gh_asm::NttAsmArgs q;
            q.in = A.in; q.out = A.out; q.tw = A.tw; q.pre = A.pre;
            q.post = A.post ? (const void*)A.post : (A.has_post_scalar ? (const void*)d->d_size_inv : nullptr);
            q.post_stride = A.post ? 104u : 0u;
            q.log_n = log_n; q.log_ns = (uint32_t)log_ns; q.inverse = (uint32_t)A.inverse;
            q.n_waves = 1u << (log_n - 8); q.pad = 0;
            if ((rc = gh_asm::ntt_pass_launch(std::is_same<P, P6>::value ? 6 : 4, ks[s], q, g.stream))) return rc;
        } else {
            hipLaunchKernelGGL((ntt_pass_kernel<P>), dim3(grid), dim3(threads), (size_t)NL * 4 * E, g.stream, A);
        }
        if (three && s < 3) cur = s == 2 ? 0 : 1;      // after pass 2 the result is in the caller's vector
        else cur ^= 1;
        log_ns += ks[s];
    }
    HIPCHK(hipGetLastError());
    if (cur == 1) HIPCHK(hipMemcpyAsync(d_data, d->scratch, ((size_t)96) << log_n, hipMemcpyDeviceToDevice, g.stream));
    HIPCHK(hipEventRecord(g.ev[5], g.stream));
    if (sync) {
        HIPCHK(hipStreamSynchronize(g.stream));
        HIPCHK(hipEventElapsedTime(&g.last_fft_ms, g.ev[4], g.ev[5]));
    }
    return GH_OK;
}

template <class P> int vec_op(int op, void* d_a, const void* d_b, const uint64_t* scalar12, size_t n, bool sync = true) {
    if (n == 0) return GH_OK;
    Fp s = fp_zero();
    if (op == 2) s = fp_from_abi<P>(reinterpret_cast<const uint32_t*>(scalar12));
    dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    if (op == 0) hipLaunchKernelGGL((vec_op_kernel<P, 0>), grid, blk, 0, g.stream, (uint32_t*)d_a, (const uint32_t*)d_b, s, n);
    else if (op == 1) hipLaunchKernelGGL((vec_op_kernel<P, 1>), grid, blk, 0, g.stream, (uint32_t*)d_a, (const uint32_t*)d_b, s, n);
    else hipLaunchKernelGGL((vec_op_kernel<P, 2>), grid, blk, 0, g.stream, (uint32_t*)d_a, (const uint32_t*)d_b, s, n);
    HIPCHK(hipGetLastError());
    if (sync) HIPCHK(hipStreamSynchronize(g.stream));
    return GH_OK;
}


// R1CStoQAP::witness_map, transform part (proof-systems/src/groth16/r1cs_to_qap.rs:121-166)
template <class P> int witness_map_t(int fidx, void* d_a, void* d_b, void* d_c, uint32_t log_n, const uint64_t* d1,
                                     const uint64_t* d2, const uint64_t* d3, void* d_h) {
    const size_t N = (size_t)1 << log_n;
    int rc;
    // a, b -> coefficients -> evaluations on the coset (:121-122, :134-135)
    if ((rc = fft_run<P>(fidx, d_a, log_n, GH_FFT_INVERSE, false))) return rc;
    if ((rc = fft_run<P>(fidx, d_b, log_n, GH_FFT_INVERSE, false))) return rc;
    if ((rc = fft_run<P>(fidx, d_a, log_n, GH_FFT_COSET, false))) return rc;
    if ((rc = fft_run<P>(fidx, d_b, log_n, GH_FFT_COSET, false))) return rc;
    if ((rc = vec_op<P>(0, d_a, d_b, nullptr, N, false))) return rc;                        // ab = a .* b (:137)
    if ((rc = fft_run<P>(fidx, d_c, log_n, GH_FFT_INVERSE, false))) return rc;              // :153
    if ((rc = fft_run<P>(fidx, d_c, log_n, GH_FFT_COSET, false))) return rc;                // :154
    if ((rc = vec_op<P>(1, d_a, d_c, nullptr, N, false))) return rc;                        // ab -= c (:156-158)
    // divide_by_vanishing_poly_on_coset: multiply by (g^N - 1)^-1  (domain.rs:245-256, :229-231)
    Fp gen = fp_from_abi<P>(reinterpret_cast<const uint32_t*>(FieldConsts<P>::gen_m()));
    Fp gn = gen;
    for (uint32_t i = 0; i < log_n; i++) gn = fp_sqr<P>(gn);
    Fp vinv = host_fp_inv<P>(fp_sub<P>(gn, fp_one<P>()));
    uint64_t vinv_abi[12];
    fp_to_abi<P>(reinterpret_cast<uint32_t*>(vinv_abi), vinv);
    if ((rc = vec_op<P>(2, d_a, nullptr, vinv_abi, N, false))) return rc;
    if ((rc = fft_run<P>(fidx, d_a, log_n, GH_FFT_INVERSE | GH_FFT_COSET, false))) return rc;   // :161
    // h (:124-132, :163-166)
    Fp f1 = fp_from_abi<P>(reinterpret_cast<const uint32_t*>(d1)), f2 = fp_from_abi<P>(reinterpret_cast<const uint32_t*>(d2));
    Fp f3 = fp_from_abi<P>(reinterpret_cast<const uint32_t*>(d3));
    Fp d1d2 = fp_mul<P>(f1, f2);
    Fp h0 = fp_neg<P>(fp_add<P>(f3, d1d2));
    uint32_t w0[24], w1[24];
    fp_to_abi<P>(w0, h0);
    fp_to_abi<P>(w1, d1d2);
    hipLaunchKernelGGL((witness_finish_kernel<P>), dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, g.stream,
                       (const uint32_t*)d_a, (uint32_t*)d_h, N, fp_unpack(w0), fp_unpack(w1));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(g.stream));
    return GH_OK;
}

// R1CStoSAP::witness_map, transform part (proof-systems/src/gm17/r1cs_to_sap.rs:194-240): d_a, d_c hold the 2^log_n
// evaluations built by the caller (:158-192, :207-230); both are overwritten; d_h receives 2^log_n + 1 coefficients.
template <class P> int sap_witness_map_t(int fidx, void* d_a, void* d_c, uint32_t log_n, const uint64_t* d1, const uint64_t* d2, void* d_h) {
    const size_t N = (size_t)1 << log_n;
    int rc;
    if ((rc = fft_run<P>(fidx, d_a, log_n, GH_FFT_INVERSE, false))) return rc;                       // :191
    // h = 2 d1 * a  (coefficients) (:193-195)
    HIPCHK(hipMemcpyAsync(d_h, d_a, N * 96, hipMemcpyDeviceToDevice, g.stream));
    Fp f1 = fp_from_abi<P>(reinterpret_cast<const uint32_t*>(d1)), f2 = fp_from_abi<P>(reinterpret_cast<const uint32_t*>(d2));
    uint64_t d1_double[12];
    fp_to_abi<P>(reinterpret_cast<uint32_t*>(d1_double), fp_dbl<P>(f1));
    if ((rc = vec_op<P>(2, d_h, nullptr, d1_double, N, false))) return rc;
    if ((rc = fft_run<P>(fidx, d_a, log_n, GH_FFT_COSET, false))) return rc;                         // :201
    if ((rc = vec_op<P>(0, d_a, d_a, nullptr, N, false))) return rc;                                 // aa = a .* a (:203)
    if ((rc = fft_run<P>(fidx, d_c, log_n, GH_FFT_INVERSE, false))) return rc;                       // :232
    if ((rc = fft_run<P>(fidx, d_c, log_n, GH_FFT_COSET, false))) return rc;                         // :233
    if ((rc = vec_op<P>(1, d_a, d_c, nullptr, N, false))) return rc;                                 // aa -= c (:235)
    Fp gen = fp_from_abi<P>(reinterpret_cast<const uint32_t*>(FieldConsts<P>::gen_m()));
    Fp gn = gen;
    for (uint32_t i = 0; i < log_n; i++) gn = fp_sqr<P>(gn);
    Fp vinv = host_fp_inv<P>(fp_sub<P>(gn, fp_one<P>()));
    uint64_t vinv_abi[12];
    fp_to_abi<P>(reinterpret_cast<uint32_t*>(vinv_abi), vinv);
    if ((rc = vec_op<P>(2, d_a, nullptr, vinv_abi, N, false))) return rc;                            // :237
    if ((rc = fft_run<P>(fidx, d_a, log_n, GH_FFT_INVERSE | GH_FFT_COSET, false))) return rc;        // :238
    Fp d1d1 = fp_sqr<P>(f1);
    Fp h0 = fp_neg<P>(fp_add<P>(f2, d1d1));                                                          // :196-198
    uint32_t w0[24], w1[24];
    fp_to_abi<P>(w0, h0);
    fp_to_abi<P>(w1, d1d1);
    hipLaunchKernelGGL((sap_finish_kernel<P>), dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, g.stream,
                       (const uint32_t*)d_a, (uint32_t*)d_h, N, fp_unpack(w0), fp_unpack(w1));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(g.stream));
    return GH_OK;
}

template <class P> int batch_inverse_t(void* d_a, size_t n) {
    if (n == 0) return GH_OK;
    Fp* tmp;
    int rc = pool_get("inv_tmp", n * sizeof(Fp), (void**)&tmp);
    if (rc) return rc;
    const size_t threads = (n + INV_RUN - 1) / INV_RUN;
    hipLaunchKernelGGL((batch_inverse_kernel<P>), dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, g.stream, (uint32_t*)d_a, tmp, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(g.stream));
    return GH_OK;
}

template <class P> int lagrange_t(int fidx, uint32_t log_n, const uint64_t* tau12, void* d_out) {
    if ((int)log_n >= FieldConsts<P>::two_adicity) { g_err = "domain exceeds the field's 2-adicity"; return GH_E_UNSUPPORTED; }
    const size_t N = (size_t)1 << log_n;
    const Fp tau = fp_from_abi<P>(reinterpret_cast<const uint32_t*>(tau12));
    Fp t_size = tau;
    for (uint32_t i = 0; i < log_n; i++) t_size = fp_sqr<P>(t_size);
    const dim3 grid((unsigned)((N + 255) / 256)), blk(256);
    Fp one_tab = fp_one<P>();
    const Fp* tw = nullptr;
    Fp* tw1 = nullptr;
    if (log_n == 0) {                      // size-1 domain: w^0 = 1, no table
        HIPCHK(hipMalloc((void**)&tw1, sizeof(Fp)));
        HIPCHK(hipMemcpy(tw1, &one_tab, sizeof(Fp), hipMemcpyHostToDevice));
        tw = tw1;
    } else {
        Domain* d;
        int rc = get_domain<P>(fidx, (int)log_n, false, false, &d);
        if (rc) return rc;
        tw = d->tw;
    }
    int rc = GH_OK;
    if (fp_eq(t_size, fp_one<P>())) {
        hipLaunchKernelGGL((lagrange_kernel<P, 0>), grid, blk, 0, g.stream, (uint32_t*)d_out, tw, N, tau, fp_zero());
    } else {
        Fp n_int = fp_one<P>();
        for (uint32_t i = 0; i < log_n; i++) n_int = fp_dbl<P>(n_int);
        const Fp l0 = fp_mul<P>(fp_sub<P>(t_size, fp_one<P>()), host_fp_inv<P>(n_int));
        hipLaunchKernelGGL((lagrange_kernel<P, 1>), grid, blk, 0, g.stream, (uint32_t*)d_out, tw, N, tau, l0);
        rc = batch_inverse_t<P>(d_out, N);
        if (!rc) hipLaunchKernelGGL((lagrange_kernel<P, 2>), grid, blk, 0, g.stream, (uint32_t*)d_out, tw, N, tau, l0);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    if (tw1) (void)hipFree(tw1);
    if (!rc && e != hipSuccess) { g_err = std::string("lagrange coefficients: ") + hipGetErrorString(e); rc = GH_E_HIP; }
    return rc;
}

int sap_witness_map(gh_field_t field, void* d_a, void* d_c, uint32_t log_n, const uint64_t* d1, const uint64_t* d2, void* d_h) {
    if (field == GH_MNT4753_FR) return sap_witness_map_t<P6>(0, d_a, d_c, log_n, d1, d2, d_h);
    if (field == GH_MNT6753_FR) return sap_witness_map_t<P4>(1, d_a, d_c, log_n, d1, d2, d_h);
    g_err = "unknown field id";
    return GH_E_BAD_ARG;
}
int batch_inverse(gh_field_t field, void* d_a, size_t n) {
    if (field == GH_MNT4753_FR) return batch_inverse_t<P6>(d_a, n);
    if (field == GH_MNT6753_FR) return batch_inverse_t<P4>(d_a, n);
    g_err = "unknown field id";
    return GH_E_BAD_ARG;
}
int lagrange_coefficients(gh_field_t field, uint32_t log_n, const uint64_t* tau12, void* d_out) {
    if (field == GH_MNT4753_FR) return lagrange_t<P6>(0, log_n, tau12, d_out);
    if (field == GH_MNT6753_FR) return lagrange_t<P4>(1, log_n, tau12, d_out);
    g_err = "unknown field id";
    return GH_E_BAD_ARG;
}

int witness_map(gh_field_t field, void* d_a, void* d_b, void* d_c, uint32_t log_n, const uint64_t* d1,
                const uint64_t* d2, const uint64_t* d3, void* d_h) {
    if (field == GH_MNT4753_FR) return witness_map_t<P6>(0, d_a, d_b, d_c, log_n, d1, d2, d3, d_h);
    if (field == GH_MNT6753_FR) return witness_map_t<P4>(1, d_a, d_b, d_c, log_n, d1, d2, d3, d_h);
    g_err = "unknown field id";
    return GH_E_BAD_ARG;
}

int fft_run(gh_field_t field, void* d_data, uint32_t log_n, uint32_t flags) {
    if (field == GH_MNT4753_FR) return fft_run<P6>(0, d_data, log_n, flags);
    if (field == GH_MNT6753_FR) return fft_run<P4>(1, d_data, log_n, flags);
    g_err = "unknown field id";
    return GH_E_BAD_ARG;
}
int vec_op(gh_field_t field, int op, void* d_a, const void* d_b, const uint64_t* s, size_t n) {
    if (field == GH_MNT4753_FR) return vec_op<P6>(op, d_a, d_b, s, n);
    if (field == GH_MNT6753_FR) return vec_op<P4>(op, d_a, d_b, s, n);
    g_err = "unknown field id";
    return GH_E_BAD_ARG;
}

}  // namespace gh_rt

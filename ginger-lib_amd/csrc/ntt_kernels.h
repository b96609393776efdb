// ntt_kernels.h -- radix-2 evaluation-domain transforms (fft / ifft / coset variants of
// algebra/src/fft/domain.rs:113-179) as multi-pass Stockham NTT on gfx950.
//
// Mathematical contract (domain.rs, natural order in / out):
//   fft(a)[k]  = sum_j a[j] w^(jk),  w = group_gen (domain.rs:76-79);  ifft uses w^-1 and size_inv;
//   coset_fft scales a[i] by g^i (g = 17) first, coset_ifft scales by g^-i last (:140-179).
// The reference computes this with bit-reversal + log n in-place DIT stages whose twiddles are a
// running product (2 Fp-mul per butterfly, serial_fft :315-358).  Results are canonical field
// elements, so only the values matter; the schedule here is chosen for the GPU:
//
//   N = 2^n is factored into P = ceil(n/8) passes of radix R_s = 2^k_s (k_s <= 8).  Pass s is a
//   Stockham (self-sorting, out-of-place) step: block b owns C = 512 / R consecutive column indices
//   j in [bC, bC+C) and all R rows t; it reads x[j + t N/R] (rows are C*96-byte contiguous runs),
//   multiplies by the inter-pass twiddle w_(Ns R)^((j mod Ns) t) (pass 0: by the coset factor g^i
//   instead, if any), runs the R-point DFT as k radix-2 DIF stages through LDS (one butterfly per
//   thread per stage, element-major SoA in LDS so lanes hit distinct banks), and writes row
//   bitrev(t) to y[(j - j mod Ns) R + (j mod Ns) + bitrev(t) Ns], fused with the final scaling
//   (size_inv, or size_inv * g^-i) in the last pass.
//   Fp-mul per element per pass: 1 (twiddle) + (k-1)/2 (the last DIF stage has unit twiddles).
//
// HBM data stays in the ABI layout (12 u64, Montgomery 2^768).  The transform is linear, so raw ABI
// limbs are used directly as internal-form residues (they represent 2^14 x); twiddle tables are
// in internal Montgomery form, hence products come out in the same scaling and no conversion
// multiplications are needed on either side.
#pragma once
#include <hip/hip_runtime.h>
#include "fp29.h"

namespace gh {

// Elements per block tile.  The kernel needs ~256 VGPRs, i.e. 2 waves per SIMD = 512 threads per
// CU.  One 512-thread block per CU (1024-element tile) leaves nothing to run while that block
// sits at a barrier or in its load / store phase; two 256-thread blocks (512-element tiles,
// 52 KiB of LDS each) overlap each other.
constexpr int NTT_LOG_TILE = 9;
constexpr int NTT_MAX_TILE = 1 << NTT_LOG_TILE;
constexpr int NTT_MAX_LOGR = 8;

#ifndef GH_LD_ST_FP
#define GH_LD_ST_FP
__device__ __forceinline__ Fp ld_fp(const Fp* p) {
    Fp r;
    const uint2* q = reinterpret_cast<const uint2*>(p);
#pragma unroll
    for (int i = 0; i < NL / 2; i++) { uint2 v = q[i]; r.l[2 * i] = v.x; r.l[2 * i + 1] = v.y; }
    return r;
}
__device__ __forceinline__ void st_fp(Fp* p, const Fp& a) {
    uint2* q = reinterpret_cast<uint2*>(p);
#pragma unroll
    for (int i = 0; i < NL / 2; i++) q[i] = make_uint2(a.l[2 * i], a.l[2 * i + 1]);
}
#endif
// ABI element (24 words) <-> registers, 16-byte accesses
__device__ __forceinline__ Fp ld_abi_raw(const uint32_t* p) {
    uint32_t w[24];
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 6; i++) { uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
    return fp_unpack(w);
}
__device__ __forceinline__ void st_abi_raw(uint32_t* p, const Fp& a) {
    uint32_t w[24];
    fp_pack(w, a);
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < 6; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

struct NttPassArgs {
    const uint32_t* in;   // N x 24 words
    uint32_t* out;        // N x 24 words
    const Fp* tw;         // w^i, i < N, internal Montgomery form
    const Fp* pre;        // optional per-input-index factor (coset g^i), pass 0 only
    const Fp* post;       // optional per-output-index factor (size_inv * g^-i), last pass only
    Fp post_scalar;       // used when post == nullptr && has_post_scalar
    int has_post_scalar;
    int log_n, k, log_ns, log_c, inverse;
};

__device__ __forceinline__ uint32_t bitrev_k(uint32_t t, int k) { return __brev(t) >> (32 - k); }

template <class P>
__global__ void __launch_bounds__(NTT_MAX_TILE / 2, 2) ntt_pass_kernel(NttPassArgs A) {
    extern __shared__ uint32_t lds[];  // [NL][E]
    const int k = A.k, log_c = A.log_c;
    const int R = 1 << k, C = 1 << log_c, E = R << log_c;
    const uint32_t N = 1u << A.log_n, nmask = N - 1;
    const uint32_t stride = N >> k;
    const uint32_t ns_mask = (1u << A.log_ns) - 1;
    const int tau = threadIdx.x;
    if (tau >= (E >> 1) && E > 1) {
        // idle lanes of a small tile still have to reach the barriers below
    }
    const bool active = tau < (E >> 1);
    const int c = tau & (C - 1);
    const int q = tau >> log_c;
    const uint32_t j = blockIdx.x * C + c;
    const uint32_t kk = j & ns_mask;
    Fp a, b;
    int ta = q, tb = q + (R >> 1);
    if (active) {
        const uint32_t ia = j + (uint32_t)ta * stride, ib = j + (uint32_t)tb * stride;
        a = ld_abi_raw(A.in + (size_t)ia * 24);
        b = ld_abi_raw(A.in + (size_t)ib * 24);
        if (A.pre) {
            a = fp_mul<P>(a, ld_fp(A.pre + ia));
            b = fp_mul<P>(b, ld_fp(A.pre + ib));
        }
        if (A.log_ns > 0) {
            const int sh = A.log_n - A.log_ns - k;
            uint32_t ea = (kk * (uint32_t)ta) << sh, eb = (kk * (uint32_t)tb) << sh;
            if (A.inverse) { ea = (N - ea) & nmask; eb = (N - eb) & nmask; }
            a = fp_mul<P>(a, ld_fp(A.tw + ea));
            b = fp_mul<P>(b, ld_fp(A.tw + eb));
        }
    }
    for (int m = 0; m < k; m++) {
        const int h = R >> (m + 1);
        if (m > 0) {
            const int grp = q / h, i = q & (h - 1);
            ta = grp * 2 * h + i;
            tb = ta + h;
            if (active) {
                const int ea = (ta << log_c) + c, eb = (tb << log_c) + c;
#pragma unroll
                for (int w = 0; w < NL; w++) { a.l[w] = lds[w * E + ea]; b.l[w] = lds[w * E + eb]; }
            }
        }
        // s = a + b is written out before the twiddle product of d = a - b is started, and scheduling
        // fences keep hipcc from interleaving the two: the live set stays inside the 256-register budget.
        Fp s, d;
        const bool to_lds = m < k - 1;
        const int ea_w = (ta << log_c) + c, eb_w = (tb << log_c) + c;
        if (active) {
            s = fp_add<P>(a, b);
            if (to_lds) {
#pragma unroll
                for (int w = 0; w < NL; w++) lds[w * E + ea_w] = s.l[w];
            }
            __builtin_amdgcn_sched_barrier(0);
            d = fp_sub<P>(a, b);
            __builtin_amdgcn_sched_barrier(0);
            if (h > 1) {
                uint32_t e = (uint32_t)(q & (h - 1)) << (A.log_n - (k - m));  // w_(2h)^i = w_N^(i N / 2h)
                if (A.inverse) e = (N - e) & nmask;
                d = fp_mul<P>(d, ld_fp(A.tw + e));
            }
            if (to_lds) {
#pragma unroll
                for (int w = 0; w < NL; w++) lds[w * E + eb_w] = d.l[w];
            }
        }
        if (to_lds) {
            __syncthreads();
        } else if (active) {
            const uint32_t jbase = ((j - kk) << k) + kk;
            const uint32_t oa = jbase + (bitrev_k((uint32_t)ta, k) << A.log_ns);
            const uint32_t ob = jbase + (bitrev_k((uint32_t)tb, k) << A.log_ns);
            if (A.post) {
                s = fp_mul<P>(s, ld_fp(A.post + oa));
                d = fp_mul<P>(d, ld_fp(A.post + ob));
            } else if (A.has_post_scalar) {
                s = fp_mul<P>(s, A.post_scalar);
                d = fp_mul<P>(d, A.post_scalar);
            }
            st_abi_raw(A.out + (size_t)oa * 24, s);
            st_abi_raw(A.out + (size_t)ob * 24, d);
        }
    }
}

// N == 1: the transform is the identity up to the scalings (domain.rs: serial_fft with n = 1 is a no-op).
template <class P>
__global__ void ntt_size1_kernel(const uint32_t* in, uint32_t* out, Fp post_scalar, int has_post_scalar) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        Fp a = ld_abi_raw(in);
        if (has_post_scalar) a = fp_mul<P>(a, post_scalar);
        st_abi_raw(out, a);
    }
}

// out[i + half] = out[i] * factor for i < half  (doubling construction of a power table)
template <class P>
__global__ void __launch_bounds__(256) pow_table_step_kernel(Fp* tab, uint32_t half, Fp factor) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < half) st_fp(tab + half + i, fp_mul<P>(ld_fp(tab + i), factor));
}

// pointwise kernels on ABI-layout vectors.  op 0: a*=b, 1: a-=b, 2: a*=scalar
// a*b of two ABI-Montgomery values through the internal multiplier carries 2^(768+768-754) = 2^782;
// one more product with 2^740 (CIN) brings it back to 2^768.
template <class P, int OP>
__global__ void __launch_bounds__(256) vec_op_kernel(uint32_t* a, const uint32_t* b, Fp scalar_int, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp x = ld_abi_raw(a + i * 24);
    if (OP == 0) {
        Fp y = ld_abi_raw(b + i * 24);
        x = fp_mul<P>(fp_mul<P>(x, y), fp_const<P>(P::CIN));
    } else if (OP == 1) {
        Fp y = ld_abi_raw(b + i * 24);
        x = fp_sub<P>(x, y);
    } else {
        x = fp_mul<P>(x, scalar_int);  // scalar in internal form: raw * s
    }
    st_abi_raw(a + i * 24, x);
}

// last step of witness_map (r1cs_to_qap.rs:124-132, :163-166): h[0 .. N] from ab[0 .. N)
//   h[0] = ab[0] + h0_add;  h[i] = ab[i] (0 < i < N-1);  h[N-1] = 0;  h[N] = h_last
// ab / h in ABI layout; h0_add, h_last are ABI-Montgomery values unpacked to raw limbs (plain
// modular add works on raw residues).
template <class P>
__global__ void __launch_bounds__(256) witness_finish_kernel(const uint32_t* ab, uint32_t* h, size_t N, Fp h0_add, Fp h_last) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    Fp v;
    // (N == 1: h[..N-1] is empty in the reference, so h[0] = -d3 - d1 d2 -- the i == 0 term applies to the zero)
    if (i == N) v = h_last;
    else {
        v = i == N - 1 ? fp_zero() : ld_abi_raw(ab + i * 24);
        if (i == 0) v = fp_add<P>(v, h0_add);
    }
    st_abi_raw(h + i * 24, v);
}

// last step of the GM17 / SAP witness map (proof-systems/src/gm17/r1cs_to_sap.rs:194-240): hbase holds 2 d1 * ifft(a),
//   h[i] = hbase[i] + aa[i] (i < N - 1);  h[N-1] = hbase[N-1];  h[0] += h0_add (= -d2 - d1^2);  h[N] = d1^2
template <class P>
__global__ void __launch_bounds__(256) sap_finish_kernel(const uint32_t* aa, uint32_t* h, size_t N, Fp h0_add, Fp h_last) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    Fp v;
    if (i == N) v = h_last;
    else {
        v = ld_abi_raw(h + i * 24);
        if (i < N - 1) v = fp_add<P>(v, ld_abi_raw(aa + i * 24));
        if (i == 0) v = fp_add<P>(v, h0_add);
    }
    st_abi_raw(h + i * 24, v);
}

// batch_inversion (algebra/src/fields/mod.rs:412-442): a[i] <- a[i]^-1, zeros are left alone.  One thread per run of
// INV_RUN elements: Montgomery's trick inside the run (running products parked in `tmp`), one safegcd inversion per thread.
constexpr int INV_RUN = 32;
template <class P>
__global__ void __launch_bounds__(64) batch_inverse_kernel(uint32_t* a, Fp* tmp, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i0 = t * INV_RUN;
    if (i0 >= n) return;
    const int cnt = (int)(n - i0 < (size_t)INV_RUN ? n - i0 : (size_t)INV_RUN);
    Fp run = fp_one<P>();
    for (int j = 0; j < cnt; j++) {
        const Fp x = fp_from_abi<P>(a + (i0 + j) * 24);
        if (!fp_is_zero(x)) run = fp_mul<P>(run, x);
        st_fp(tmp + i0 + j, run);                          // product of the non-zero elements up to and including j
    }
    Fp inv = fp_inv<P>(run);
    for (int j = cnt - 1; j >= 0; j--) {
        const Fp x = fp_from_abi<P>(a + (i0 + j) * 24);
        if (fp_is_zero(x)) continue;
        const Fp before = j > 0 ? ld_fp(tmp + i0 + j - 1) : fp_one<P>();
        fp_to_abi<P>(a + (i0 + j) * 24, fp_mul<P>(inv, before));
        inv = fp_mul<P>(inv, x);
    }
}

// evaluate_all_lagrange_coefficients (algebra/src/fft/domain.rs:183-219), tau outside the domain:
//   step 1: out[i] = tau - w^i            (then batch_inverse_kernel)
//   step 2: out[i] = out[i] * l0 * w^i    with l0 = (tau^N - 1) / N
// tau in the domain (tau^N = 1): out[i] = (w^i == tau) ? 1 : 0.   tw: w^i in internal form; tau, l0 internal.
template <class P, int STEP>
__global__ void __launch_bounds__(256) lagrange_kernel(uint32_t* out, const Fp* __restrict__ tw, size_t N, Fp tau, Fp l0) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const Fp w = ld_fp(tw + i);
    if (STEP == 0) {            // indicator
        if (fp_eq(w, tau)) fp_to_abi<P>(out + i * 24, fp_one<P>());
        else st_abi_raw(out + i * 24, fp_zero());
    } else if (STEP == 1) {
        fp_to_abi<P>(out + i * 24, fp_sub<P>(tau, w));
    } else {
        const Fp x = fp_from_abi<P>(out + i * 24);
        fp_to_abi<P>(out + i * 24, fp_mul<P>(x, fp_mul<P>(l0, w)));
    }
}

// ABI Montgomery -> internal Montgomery for a vector of n elements (table seeds etc.)
template <class P>
__global__ void __launch_bounds__(256) abi_to_internal_kernel(const uint32_t* in, Fp* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) st_fp(out + i, fp_from_abi<P>(in + i * 24));
}

}  // namespace gh

// msm_kernels.h -- variable-base multi-scalar multiplication (Pippenger bucket method) on gfx950.
//
// Reference algorithm: algebra/src/msm/variable_base.rs:10-83 -- per c-bit window a *serial* loop
// over all pairs doing buckets[digit-1].add_assign_mixed(base) (:36-59), batch-normalise, running
// sum (:60-66), then a Horner fold of the windows (:73-82); parallelism = number of windows.
// The group sum is order-independent (only the affine image of the result is canonical),
// so the device uses its own schedule:
//
//   1. msm_digits_kernel     one thread per scalar: signed c-bit digits d in [-2^(c-1), 2^(c-1)]
//                            (halves the bucket count), histogram of |d| per (window, bucket).
//   2. exclusive scan of the histogram (msm_scan_*), bucket order by descending size
//      (msm_size_*: counting sort on the bucket size so the 64 lanes of a wave get equal work).
//   3. msm_scatter_kernel    counting-sort scatter: for every bucket the list of (pair index | sign).
//   4. msm_accumulate_kernel one thread per bucket walks its list: gather the base (internal
//                            layout, 208 B for G1), conditional negate, projective mixed add
//                            (ec29.h proj_madd, 11 Fp-mul).  ~94 % of all work (as in the reference).
//      chunk mode + msm_heavy_combine_kernel: buckets longer than the heavy threshold are cut into
//                            chunks that the same kernel sums like ordinary buckets, then combined
//                            (skewed real-world witnesses -- many equal small scalars -- and
//                            sparsely populated top windows).
//   5. msm_wave_reduce_kernel  sum_b b * B_b per window without the reference's per-window inversion
//                            and without doublings: "wave programs" whose every step is one
//                            projective addition from a single inlined call site (serial sums per
//                            lane, then tree / suffix scan / tree across the 64 lanes through LDS:
//                            the wavefront-wide bucket reduction); two launches.
//   6. window fold           ~750 dependent doublings: latency-bound, done on the host
//                            (msm_impl.h fold_windows) with the reduction's powers of two merged in.
#pragma once
#include <hip/hip_runtime.h>
#include "ec29.h"
#include "aff_kernels.h"

namespace gh {

constexpr int MSM_REDUCE_L = 16;         // buckets folded serially per lane in reduce level 1
constexpr int MSM_MAX_HEAVY_THRESHOLD = 1024;  // upper bound of the run-time heavy threshold
constexpr int MSM_SIZE_BINS = MSM_MAX_HEAVY_THRESHOLD + 2;

// ---------------------------------------------------------------- generic point load / store
template <class C> __device__ __forceinline__ Aff<C> ld_aff(const Aff<C>* p) {
    Aff<C> r;
    const uint2* q = reinterpret_cast<const uint2*>(p);
    uint2* d = reinterpret_cast<uint2*>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(Aff<C>) / 8); i++) d[i] = q[i];
    return r;
}
template <class C> __device__ __forceinline__ Proj<C> ld_proj(const Proj<C>* p) {
    Proj<C> r;
    const uint2* q = reinterpret_cast<const uint2*>(p);
    uint2* d = reinterpret_cast<uint2*>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(Proj<C>) / 8); i++) d[i] = q[i];
    return r;
}
template <class C> __device__ __forceinline__ void st_proj(Proj<C>* p, const Proj<C>& v) {
    uint2* q = reinterpret_cast<uint2*>(p);
    const uint2* s = reinterpret_cast<const uint2*>(&v);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(Proj<C>) / 8); i++) q[i] = s[i];
}

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

// ---------------------------------------------------------------- bases: ABI -> internal layout
// in: n x (2 * DEG * 24) words, x || y, either Montgomery 2^768 (the in-memory form, fp_768.rs:24-30) or --
// canonical != 0 -- plain integers < p (what ToBytes writes: into_repr(), fp_768.rs:784-789);
// out: n x Aff<C> (internal 2^754): one product by 2^740 resp. 2^1508 per coefficient.
template <class C>
__global__ void __launch_bounds__(256) msm_convert_bases_kernel(const uint32_t* in, Aff<C>* out, size_t n, int canonical) {
    typedef typename C::F F;
    typedef typename C::PF PF;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* p = in + i * (size_t)(48 * F::DEG);
    const Fp k = canonical ? fp_const<PF>(PF::R2I) : fp_const<PF>(PF::CIN);
    Aff<C> a;
#pragma unroll
    for (int d = 0; d < F::DEG; d++) {
        F::comp(a.x, d) = fp_mul<PF>(fp_unpack(p + 24 * d), k);
        F::comp(a.y, d) = fp_mul<PF>(fp_unpack(p + 24 * (F::DEG + d)), k);
    }
    uint2* q = reinterpret_cast<uint2*>(out + i);
    const uint2* s = reinterpret_cast<const uint2*>(&a);
#pragma unroll
    for (int w = 0; w < (int)(sizeof(Aff<C>) / 8); w++) q[w] = s[w];
}

// ---------------------------------------------------------------- 0. precomputed shift tables
// Bases are static per proving key (groth16/mod.rs:158-170), and 288 GB of HBM hold a lot of them:
// for a resident key the library can store, next to P_i, the points 2^(c w) P_i for every window w
// (table row w; W rows of n affine points).  Window w of scalar i then adds table[w][i] instead of
// P_i, all windows share ONE set of 2^(c-1) buckets, and the per-window reduction and the Horner
// fold over windows (variable_base.rs:60-82) collapse into a single bucket reduction.  With the
// bucket count decoupled from the window count, c can grow (c = 21 at n = 2^20: 36 additions per
// pair instead of 48).
//
// One thread per base: c doublings per row (out-of-line dbl-2007-bl), projective rows parked in the
// table / a scratch array, then ONE field inversion per base (Montgomery's trick over its W - 1 Z
// coordinates, as batch_normalization does: short_weierstrass_projective.rs:402-442) turns them
// into affine rows.
template <class P> __device__ __attribute__((noinline)) Fp dev_fp_inv(const Fp& a) {   // a^(p-2)
    uint32_t e[NL];
    int32_t bw = -2;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int32_t x = (int32_t)P::P[i] + bw;
        e[i] = (uint32_t)x & LM;
        bw = x >> 31;
    }
    Fp r = fp_one<P>();
#pragma unroll
    for (int i = NL - 1; i >= 0; i--) {
        const uint32_t ei = e[i];
#pragma nounroll
        for (int b = LB - 1; b >= 0; b--) {
            r = fp_sqr_call<P>(r);
            if ((ei >> b) & 1u) r = fp_mul_call<P>(r, a);
        }
    }
    return r;
}
template <class F> struct DevInv;
template <class P, bool I> struct DevInv<F1<P, I>> {
    static __device__ __forceinline__ Fp inv(const Fp& a) { return dev_fp_inv<P>(a); }
};
template <class P, int NR, bool I> struct DevInv<F2<P, NR, I>> {   // (a0 - a1 X) / (a0^2 - NR a1^2)   (fp2.rs inverse)
    static __device__ __forceinline__ Fp2T inv(const Fp2T& a) {
        Fp n = fp_sub<P>(fp_sqr_call<P>(a.c0), fp_mul_small<P, NR>(fp_sqr_call<P>(a.c1)));
        Fp ni = dev_fp_inv<P>(n);
        return Fp2T{fp_mul_call<P>(a.c0, ni), fp_neg<P>(fp_mul_call<P>(a.c1, ni))};
    }
};
template <class P, int NR, bool I> struct DevInv<F3<P, NR, I>> {   // norm-based inverse (fp3.rs inverse)
    static __device__ __forceinline__ Fp3T inv(const Fp3T& a) {
        Fp t0 = fp_sqr_call<P>(a.c0), t1 = fp_sqr_call<P>(a.c1), t2 = fp_sqr_call<P>(a.c2);
        Fp t3 = fp_mul_call<P>(a.c0, a.c1), t4 = fp_mul_call<P>(a.c0, a.c2), t5 = fp_mul_call<P>(a.c1, a.c2);
        Fp c0 = fp_sub<P>(t0, fp_mul_small<P, NR>(t5));
        Fp c1 = fp_sub<P>(fp_mul_small<P, NR>(t2), t3);
        Fp c2 = fp_sub<P>(t1, t4);
        Fp n = fp_add<P>(fp_mul_call<P>(a.c0, c0),
                         fp_mul_small<P, NR>(fp_add<P>(fp_mul_call<P>(a.c2, c1), fp_mul_call<P>(a.c1, c2))));
        Fp ni = dev_fp_inv<P>(n);
        return Fp3T{fp_mul_call<P>(c0, ni), fp_mul_call<P>(c1, ni), fp_mul_call<P>(c2, ni)};
    }
};

template <class T> __device__ __forceinline__ T ld_words(const T* p) {
    T r;
    const uint2* q = reinterpret_cast<const uint2*>(p);
    uint2* d = reinterpret_cast<uint2*>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 8); i++) d[i] = q[i];
    return r;
}
template <class T> __device__ __forceinline__ void st_words(T* p, const T& v) {
    uint2* q = reinterpret_cast<uint2*>(p);
    const uint2* s = reinterpret_cast<const uint2*>(&v);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 8); i++) q[i] = s[i];
}

// table: W rows of n points (row 0 = the bases themselves, already written by the caller);
// this launch covers bases [i0, i0 + cnt).  zs / zp: (W - 1) x slab field elements of scratch
// (Z_w and the running products Z_1 .. Z_w).  bad[0] is set when a doubling chain reaches infinity
// (a base of 2-power order: the caller then keeps the plain per-window path for this key).
template <class C>
__global__ void __launch_bounds__(64)
msm_precompute_kernel(Aff<C>* __restrict__ table, const uint8_t* __restrict__ infinity, size_t n, size_t i0, size_t cnt,
                      size_t slab, int c, int W, typename C::FC::T* __restrict__ zs, typename C::FC::T* __restrict__ zp,
                      uint32_t* __restrict__ bad) {
    typedef typename C::FC F;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cnt) return;
    const size_t i = i0 + t;
    if (infinity != nullptr && infinity[i] != 0) {   // never read (its digits are dropped); keep the rows defined
        const Aff<C> b = ld_words(table + i);
        for (int w = 1; w < W; w++) st_words(table + (size_t)w * n + i, b);
        return;
    }
    const Aff<C> b = ld_words(table + i);
    Proj<C> p{b.x, b.y, F::one()};
    typename F::T run = F::one();
    for (int w = 1; w < W; w++) {
        for (int d = 0; d < c; d++) p = proj_dbl_call<C>(p);
        if (F::is_zero(p.z)) { atomicOr(bad, 1u); return; }
        st_words(table + (size_t)w * n + i, Aff<C>{p.x, p.y});
        st_words(zs + (size_t)(w - 1) * slab + t, p.z);
        run = F::mul(run, p.z);
        st_words(zp + (size_t)(w - 1) * slab + t, run);
    }
    typename F::T inv = DevInv<F>::inv(run);   // 1 / (Z_1 ... Z_(W-1))
    for (int w = W - 1; w >= 1; w--) {
        typename F::T zi = inv;
        if (w > 1) zi = F::mul(inv, ld_words(zp + (size_t)(w - 2) * slab + t));   // 1 / Z_w
        inv = F::mul(inv, ld_words(zs + (size_t)(w - 1) * slab + t));
        Aff<C> q = ld_words(table + (size_t)w * n + i);
        q.x = F::mul(q.x, zi);
        q.y = F::mul(q.y, zi);
        st_words(table + (size_t)w * n + i, q);
    }
}

// The same table with the doublings in JACOBIAN coordinates (round 3): dbl-2007-bl is 1 M + 8 S + a ZZ^2 against the 5 M + 6 S of
// the homogeneous doubling above; for the prime-field curves (G1) the products are inlined as well, where the kernel above
// calls them out of line -- 2^20 G1 bases, c = 21: 0.85 -> 0.40 s.  Row w holds (X, Y) of 2^(c w) P until the backward
// sweep turns them into x = X / Z^2, y = Y / Z^3 with ONE inversion per base over the row Z's (Montgomery's trick, as above).
template <class C, class F>
__global__ void __launch_bounds__(F::DEG == 1 ? 256 : 64)
msm_precompute_jac_kernel(Aff<C>* __restrict__ table, const uint8_t* __restrict__ infinity, size_t n, size_t i0, size_t cnt,
                          size_t slab, int c, int W, typename F::T* __restrict__ zs, typename F::T* __restrict__ zp,
                          uint32_t* __restrict__ bad) {
    typedef typename F::T T;          // F: the inlined prime field for G1, the out-of-line towers for G2 (same formulas)
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cnt) return;
    const size_t i = i0 + t;
    const Aff<C> b = ld_words(table + i);
    if (infinity != nullptr && infinity[i] != 0) {
        for (int w = 1; w < W; w++) st_words(table + (size_t)w * n + i, b);
        return;
    }
    T X = b.x, Y = b.y, Z = F::one(), run = F::one();
    for (int w = 1; w < W; w++) {
#pragma nounroll
        for (int d = 0; d < c; d++) {
            const T xx = F::sqr(X), yy = F::sqr(Y), zz = F::sqr(Z);
            const T yyyy = F::sqr(yy);
            const T s = F::dbl(F::sub(F::sub(F::sqr(F::add(X, yy)), xx), yyyy));
            const T m = F::add(F::add(F::dbl(xx), xx), C::mul_by_a(F::sqr(zz)));
            const T z3 = F::sub(F::sub(F::sqr(F::add(Y, Z)), yy), zz);
            X = F::sub(F::sqr(m), F::dbl(s));
            Y = F::sub(F::mul(m, F::sub(s, X)), F::dbl(F::dbl(F::dbl(yyyy))));
            Z = z3;
        }
        if (F::is_zero(Z)) { atomicOr(bad, 1u); return; }
        st_words(table + (size_t)w * n + i, Aff<C>{X, Y});
        st_words(zs + (size_t)(w - 1) * slab + t, Z);
        run = F::mul(run, Z);
        st_words(zp + (size_t)(w - 1) * slab + t, run);
    }
    T inv = DevInv<F>::inv(run);   // 1 / (Z_1 ... Z_(W-1))
    for (int w = W - 1; w >= 1; w--) {
        T zi = inv;
        if (w > 1) zi = F::mul(inv, ld_words(zp + (size_t)(w - 2) * slab + t));   // 1 / Z_w
        inv = F::mul(inv, ld_words(zs + (size_t)(w - 1) * slab + t));
        Aff<C> q = ld_words(table + (size_t)w * n + i);
        const T zi2 = F::sqr(zi);
        q.x = F::mul(q.x, zi2);
        q.y = F::mul(q.y, F::mul(zi2, zi));
        st_words(table + (size_t)w * n + i, q);
    }
}

// Wave-aggregated counter increment: returns the old value of base[key] as if every active lane
// had done atomicAdd(base + key, 1).  Up to `iters` distinct keys are combined into one atomic
// each (leader election by ballot); the rest fall back to per-lane atomics.  Random digits pay a
// few ballots; skewed digits (the top window holds only 0/1/2, witnesses are full of 0/1) no
// longer serialise a million atomics on one address.  Must be called by all 64 lanes of the wave.
static __device__ __forceinline__ uint32_t wave_agg_inc(uint32_t* base, uint32_t key, bool active, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(active);
    uint32_t result = 0;
    bool done = !active;
    for (int it = 0; it < iters && todo; it++) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lkey = __shfl(key, leader);
        const bool mine = !done && key == lkey;
        const unsigned long long same = __ballot(mine);
        uint32_t b = 0;
        if (lane == leader) b = atomicAdd(base + lkey, (uint32_t)__popcll(same));
        b = __shfl(b, leader);
        if (mine) { result = b + (uint32_t)__popcll(same & ((1ull << lane) - 1ull)); done = true; }
        todo &= ~same;
    }
    if (!done) result = atomicAdd(base + key, 1u);
    return result;
}

// ---------------------------------------------------------------- 1. digits + histogram
// scalars: n x 24 words canonical (< r).  digits[w * n + i] = signed digit (0 = no contribution).
// counts[w * win_stride + |d| - slot_shift] += 1; win_stride = nb = 2^(c-1) + 1 (slot 0 unused, slot_shift 0) gives every
// window its own bucket set; win_stride = 0 files all windows into ONE bucket set (precomputed
// shift tables, section 0 below: window w then reads its bases from table row w).  That merged set uses slot_shift 1:
// slot = |d| - 1, exactly 2^(c-1) slots -- a power of two, so the bucket reduction's segments tile it without a
// remainder (one straggling wave program on an already occupied SIMD doubled the whole launch); the weight of a slot
// is then slot + 1 and the fold adds the plain sum of all buckets once (msm_impl.h fold_merged).
//
// (a) sign folding: s > r/2 is replaced by r - s with the base negated (s P = (r - s)(-P)), so the
//     magnitude is below 2^752 and bit 752 never needs a window.
// (b) signed recoding of the c-bit windows: v = bits(s, wc, c) + carry; if v > 2^(c-1):
//     d = v - 2^c, carry = 1  (reference digit rule, unsigned: variable_base.rs:43-50).
// (c) the top window.  num_windows = floor(752 / c) + 1.  If c does not divide 752 the last window
//     holds the 752 mod c leftover bits plus the carry and never carries out.  If c divides 752
//     (c = 16: 47 * 16) the last window would hold nothing but the carry -- one bucket with n/2
//     entries.  Instead (top_unsigned) window W-2, the top real one, is taken UNSIGNED:
//     v <= 0.885 * 2^c + 1 is filed under slot v of window W-2 if v <= 2^(c-1), else under slot
//     v - 2^(c-1) of window W-1, which thereby is "region b" of window W-2 (same window weight,
//     slot offset 2^(c-1): the host adds 2^(c-1) * sum(region b), msm_impl.h fold_windows).
struct MsmModulus { uint32_t w[24]; };

// ---- equal bases.  A proving key holds the SAME point for every variable with the same polynomial (the `Benchmark` circuit's
// closing constraint puts L_last(t) into B for every recorded variable: half of b_query is one point), and
// sum s_i P = (sum s_i) P: the scalars of equal (or opposite) bases are added up front and the MSM runs on the distinct bases only
// (nothing in the reference does this; VariableBaseMSM::multi_scalar_mul's result is the same group element).  Without it every
// pair of equal bases that meet in a bucket is a doubling / a cancellation (swp.rs:492) -- millions per MSM on such a key.
// msm_base_hash_kernel: 128 bits over the abscissa's limbs per base (0, 0 for infinity); the host groups equal hashes
// (msm_impl.h dedup_bases); msm_dup_verify_kernel compares every member with its group's first base limb for limb (a hash
// collision drops out of the group) and records the sign; msm_merge_scalars_kernel does the sums at MSM time.
template <class C>
static __global__ void __launch_bounds__(256)
msm_base_hash_kernel(const Aff<C>* __restrict__ pts, const uint8_t* __restrict__ inf, size_t n, uint64_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t h1 = 0x9E3779B97F4A7C15ull, h2 = 0xC2B2AE3D27D4EB4Full;
    if (inf && inf[i]) { out[2 * i] = 0; out[2 * i + 1] = 0; return; }
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&pts[i].x);
    constexpr int NW = (int)(sizeof(pts[0].x) / 4);
    for (int k = 0; k < NW; k++) {
        const uint64_t x = w[k];
        h1 = ((h1 ^ x) * 0xFF51AFD7ED558CCDull); h1 ^= h1 >> 29;
        h2 = (h2 + x) * 0x94D049BB133111EBull; h2 = (h2 << 27) | (h2 >> 37);
    }
    if (h1 == 0 && h2 == 0) h1 = 1;
    out[2 * i] = h1; out[2 * i + 1] = h2;
}

// members[j]: base index (bit 31 clear); group g = members[starts[g] .. starts[g + 1]), its first entry the canonical base.
// Sets bit 31 where the member is the NEGATIVE of the canonical base; flags[j] = 1 where it is neither (not the same point).
template <class C>
static __global__ void __launch_bounds__(256)
msm_dup_verify_kernel(const Aff<C>* __restrict__ pts, const uint32_t* __restrict__ starts, uint32_t n_groups, uint32_t* __restrict__ members,
                      uint8_t* __restrict__ flags) {
    const uint32_t g = blockIdx.x;
    if (g >= n_groups) return;
    const uint32_t lo = starts[g], hi = starts[g + 1];
    const uint32_t canon = members[lo] & 0x7FFFFFFFu;
    constexpr int NW = (int)(sizeof(pts[0].x) / 4);
    const uint32_t* cx = reinterpret_cast<const uint32_t*>(&pts[canon].x);
    const uint32_t* cy = reinterpret_cast<const uint32_t*>(&pts[canon].y);
    for (uint32_t j = lo + 1 + threadIdx.x; j < hi; j += blockDim.x) {
        const uint32_t m = members[j] & 0x7FFFFFFFu;
        const uint32_t* mx = reinterpret_cast<const uint32_t*>(&pts[m].x);
        const uint32_t* my = reinterpret_cast<const uint32_t*>(&pts[m].y);
        bool same_x = true, same_y = true;
        for (int k = 0; k < NW; k++) { same_x &= mx[k] == cx[k]; same_y &= my[k] == cy[k]; }
        // equal x on the curve: y equal or opposite (y = 0 would be both: a point of order two never enters a key)
        flags[j] = same_x ? 0 : 1;
        members[j] = m | ((same_x && !same_y) ? 0x80000000u : 0u);
    }
}

// out = in with, per group, the canonical base's scalar replaced by the signed sum of the group's scalars mod r and the other
// members' scalars zeroed.  One block per group; scalars are canonical integers below r (12 x u64 as 24 words).
static __device__ __forceinline__ void scalar_add_mod(uint32_t* a, const uint32_t* b, const MsmModulus& r) {
    uint32_t cy = 0;
    uint32_t t[24];
#pragma unroll
    for (int k = 0; k < 24; k++) { const uint64_t x = (uint64_t)a[k] + b[k] + cy; a[k] = (uint32_t)x; cy = (uint32_t)(x >> 32); }
    uint32_t bw = 0;
#pragma unroll
    for (int k = 0; k < 24; k++) { const uint64_t x = (uint64_t)a[k] - r.w[k] - bw; t[k] = (uint32_t)x; bw = (uint32_t)(x >> 32) & 1u; }
    if (cy || !bw) {
#pragma unroll
        for (int k = 0; k < 24; k++) a[k] = t[k];
    }
}
// Two steps, so that one huge group (half a proving key's b_query can be ONE point) is not one block's work: the groups are cut
// into chunks of at most MSM_DUP_CHUNK members (chunks[3 k] = first member, [3 k + 1] = end, [3 k + 2] = group); step 1 sums a
// chunk's scalars into partial[k] and zeroes its non-canonical members in `out`, step 2 adds a group's partial sums into its
// canonical base's scalar.
constexpr uint32_t MSM_DUP_CHUNK = 4096;
static __global__ void __launch_bounds__(256)
msm_merge_scalars_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n, const uint32_t* __restrict__ starts,
                         const uint32_t* __restrict__ members, const uint32_t* __restrict__ chunks, uint32_t n_chunks,
                         uint32_t* __restrict__ partial, MsmModulus r) {
    __shared__ uint32_t part[256][25];
    const uint32_t ck = blockIdx.x;
    if (ck >= n_chunks) return;
    const uint32_t lo = chunks[3 * ck], hi = chunks[3 * ck + 1], first = starts[chunks[3 * ck + 2]];
    uint32_t acc[24];
#pragma unroll
    for (int k = 0; k < 24; k++) acc[k] = 0;
    for (uint32_t j = lo + threadIdx.x; j < hi; j += blockDim.x) {
        const uint32_t e = members[j], m = e & 0x7FFFFFFFu;
        if (m >= n) continue;                                      // the call uses fewer scalars than the key has bases
        uint32_t s[24];
        const uint4* q = reinterpret_cast<const uint4*>(in + (size_t)m * 24);
#pragma unroll
        for (int k = 0; k < 6; k++) { const uint4 v = q[k]; s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w; }
        if (e >> 31) {                                             // - s = r - s (0 stays 0)
            uint32_t any = 0, bw = 0;
#pragma unroll
            for (int k = 0; k < 24; k++) any |= s[k];
            if (any) {
#pragma unroll
                for (int k = 0; k < 24; k++) { const uint64_t x = (uint64_t)r.w[k] - s[k] - bw; s[k] = (uint32_t)x; bw = (uint32_t)(x >> 32) & 1u; }
            }
        }
        scalar_add_mod(acc, s, r);
        if (j != first) {
            uint4* o = reinterpret_cast<uint4*>(out + (size_t)m * 24);
#pragma unroll
            for (int k = 0; k < 6; k++) o[k] = make_uint4(0, 0, 0, 0);
        }
    }
#pragma unroll
    for (int k = 0; k < 24; k++) part[threadIdx.x][k] = acc[k];
    __syncthreads();
    for (uint32_t step = 128; step > 0; step >>= 1) {
        if (threadIdx.x < step) {
            uint32_t a[24], b[24];
#pragma unroll
            for (int k = 0; k < 24; k++) { a[k] = part[threadIdx.x][k]; b[k] = part[threadIdx.x + step][k]; }
            scalar_add_mod(a, b, r);
#pragma unroll
            for (int k = 0; k < 24; k++) part[threadIdx.x][k] = a[k];
        }
        __syncthreads();
    }
    if (threadIdx.x < 24) partial[(size_t)ck * 24 + threadIdx.x] = part[0][threadIdx.x];
}
// gchunk[g] .. gchunk[g + 1]: the chunks of group g (consecutive)
static __global__ void __launch_bounds__(64)
msm_merge_groups_kernel(uint32_t* __restrict__ out, size_t n, const uint32_t* __restrict__ starts, const uint32_t* __restrict__ members,
                        const uint32_t* __restrict__ gchunk, uint32_t n_groups, const uint32_t* __restrict__ partial, MsmModulus r) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const uint32_t canon = members[starts[g]] & 0x7FFFFFFFu;
    if (canon >= n) return;                                        // members ascend: none of the group is in this call
    uint32_t acc[24];
#pragma unroll
    for (int k = 0; k < 24; k++) acc[k] = 0;
    for (uint32_t ck = gchunk[g]; ck < gchunk[g + 1]; ck++) {
        uint32_t s[24];
#pragma unroll
        for (int k = 0; k < 24; k++) s[k] = partial[(size_t)ck * 24 + k];
        scalar_add_mod(acc, s, r);
    }
#pragma unroll
    for (int k = 0; k < 24; k++) out[(size_t)canon * 24 + k] = acc[k];
}

static __global__ void __launch_bounds__(256)
msm_digits_kernel(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ infinity, size_t n, int c,
                  int num_windows, uint32_t win_stride, int top_unsigned, MsmModulus r,
                  int32_t* __restrict__ digits, uint32_t* __restrict__ counts, int agg_iters, uint32_t slot_shift,
                  uint32_t sets /* bucket sets: window w files into set w % sets (per-window path: sets = num_windows) */) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    uint32_t s[25];
#pragma unroll
    for (int k = 0; k < 25; k++) s[k] = 0;
    if (valid) {
        const uint4* q = reinterpret_cast<const uint4*>(scalars + i * 24);
#pragma unroll
        for (int k = 0; k < 6; k++) { uint4 v = q[k]; s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w; }
    }
    // (a) t = r - s;  negate if t < s
    uint32_t t[24];
    uint32_t bw = 0;
    bool lt = false;   // t < s, decided by the most significant differing word
#pragma unroll
    for (int k = 0; k < 24; k++) {
        uint64_t x = (uint64_t)r.w[k] - s[k] - bw;
        t[k] = (uint32_t)x;
        bw = (uint32_t)(x >> 32) & 1u;
        if (t[k] != s[k]) lt = t[k] < s[k];
    }
    const bool sneg = lt && bw == 0;   // bw != 0 would mean s > r (non-canonical input): leave as is
    if (sneg) {
#pragma unroll
        for (int k = 0; k < 24; k++) s[k] = t[k];
    }
    const bool skip = !valid || (infinity != nullptr && infinity[i] != 0);
    const uint32_t half = 1u << (c - 1), full_mask = (1u << c) - 1;
    uint32_t carry = 0;
    int32_t d_next = 0;   // region-b digit of the top_unsigned scheme, emitted for window W-1
    for (int w = 0; w < num_windows; w++) {
        const int bit = w * c;
        uint32_t v = 0;
        if (bit < 768) {
            const int wi = bit >> 5, sh = bit & 31;
            uint64_t two = (uint64_t)s[wi] | ((uint64_t)s[wi + 1] << 32);
            v = (uint32_t)(two >> sh) & full_mask;
        }
        v += carry;
        int32_t d;
        if (top_unsigned && w == num_windows - 2) {
            if (v > half) { d = 0; d_next = (int32_t)(v - half); } else { d = (int32_t)v; }
            carry = 0;
        } else if (top_unsigned && w == num_windows - 1) {
            d = d_next;
        } else if (v > half) { d = (int32_t)v - (int32_t)(1u << c); carry = 1; } else { d = (int32_t)v; carry = 0; }
        if (sneg) d = -d;
        if (skip) d = 0;
        if (valid) digits[(size_t)w * n + i] = d;
        const uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
        if (counts) wave_agg_inc(counts + (size_t)((uint32_t)w % sets) * win_stride, mag - slot_shift, d != 0, agg_iters);   // (uniform branch)
    }
}

// ---------------------------------------------------------------- 2a. bucket lists without global atomics (large inputs)
// The histogram + scatter pair above issues two device-scope atomics per list entry on a counter array that is
// far larger than LDS (2 x 37.7 M at 2^20 pairs: 3.4 ms, 12 ms at 2^22).  For large inputs the lists are built by a
// two-level counting sort whose atomics all stay in LDS:
//   A  the bucket range is cut into NB <= 1024 (2048 beyond 2^23 buckets) bins of 2^bin_shift buckets; every block takes a tile of consecutive
//      entries e = w n + i, counts them per bin in LDS (msm_part_hist_kernel -> block_hist[bin][block]), an exclusive
//      scan over that array gives every (bin, block) its slice of the partitioned array, and the same tile is read
//      again to write (bucket, list value) pairs there (msm_part_scatter_kernel);
//   B  one block per bin: per-bucket counts in LDS, block scan -> counts[] / starts[], second sweep -> sorted[].
// Result: the same counts / starts / sorted arrays as the atomic path (order inside a bucket is arbitrary in both).
constexpr int MSM_PART_THREADS = 256;
constexpr int MSM_PART_MAX_BINS = 2048;   // (1024 unless the buckets need more: msm_impl.h, launch_sort)
struct MsmPartArgs {
    const int32_t* digits;
    size_t entries;        // W * n
    size_t n;
    uint32_t win_stride, row_stride, slot_shift;
    uint32_t sets;         // window w: bucket set w % sets, table row w / sets (per-window path: sets = W; full table: 1)
    uint32_t bin_shift, n_bins, tile, n_blocks;
};
// entry e -> (bucket, value); returns false for a zero digit
static __device__ __forceinline__ bool msm_part_entry(const MsmPartArgs& a, size_t e, uint32_t w, size_t w_base, uint32_t& bucket, uint32_t& value) {
    const int32_t d = a.digits[e];
    if (d == 0) return false;
    const uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
    const uint32_t i = (uint32_t)(e - w_base);
    bucket = (w % a.sets) * a.win_stride + (mag - a.slot_shift);
    value = (i + (w / a.sets) * a.row_stride) | (d < 0 ? 0x80000000u : 0u);
    return true;
}
static __global__ void __launch_bounds__(MSM_PART_THREADS) msm_part_hist_kernel(MsmPartArgs a, uint32_t* __restrict__ block_hist) {
    __shared__ uint32_t h[MSM_PART_MAX_BINS];
    for (uint32_t i = threadIdx.x; i < a.n_bins; i += MSM_PART_THREADS) h[i] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * a.tile;
    const uint32_t w0 = (uint32_t)(base / a.n);               // a tile (<= n entries) spans at most two windows
    const size_t split = (size_t)(w0 + 1) * a.n;
    for (uint32_t j = threadIdx.x; j < a.tile; j += MSM_PART_THREADS) {
        const size_t e = base + j;
        if (e >= a.entries) break;
        const uint32_t w = e >= split ? w0 + 1 : w0;
        uint32_t b, v;
        if (msm_part_entry(a, e, w, (size_t)w * a.n, b, v)) atomicAdd(&h[b >> a.bin_shift], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < a.n_bins; i += MSM_PART_THREADS) block_hist[(size_t)i * a.n_blocks + blockIdx.x] = h[i];
}
static __global__ void __launch_bounds__(MSM_PART_THREADS) msm_part_scatter_kernel(MsmPartArgs a, const uint32_t* __restrict__ block_off,
                                                                                  uint2* __restrict__ part) {
    __shared__ uint32_t cur[MSM_PART_MAX_BINS];
    for (uint32_t i = threadIdx.x; i < a.n_bins; i += MSM_PART_THREADS) cur[i] = block_off[(size_t)i * a.n_blocks + blockIdx.x];
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * a.tile;
    const uint32_t w0 = (uint32_t)(base / a.n);
    const size_t split = (size_t)(w0 + 1) * a.n;
    for (uint32_t j = threadIdx.x; j < a.tile; j += MSM_PART_THREADS) {
        const size_t e = base + j;
        if (e >= a.entries) break;
        const uint32_t w = e >= split ? w0 + 1 : w0;
        uint32_t b, v;
        if (msm_part_entry(a, e, w, (size_t)w * a.n, b, v)) {
            const uint32_t pos = atomicAdd(&cur[b >> a.bin_shift], 1u);
            part[pos] = make_uint2(b, v);
        }
    }
}
// one block per bin; dynamic LDS: 2^bin_shift counters.  block_off[bin * n_blocks] is where the bin's pairs start
// (block_off has n_bins * n_blocks + 1 elements: the last one is the total).
constexpr int MSM_BIN_THREADS = 1024;
constexpr int MSM_BIN_UNROLL = 8;
static __global__ void __launch_bounds__(MSM_BIN_THREADS) msm_bin_sort_kernel(const uint2* __restrict__ part, const uint32_t* __restrict__ block_off,
                                                                             uint32_t n_blocks, uint32_t bin_shift, uint32_t total,
                                                                             uint32_t* __restrict__ counts, uint32_t* __restrict__ starts,
                                                                             uint32_t* __restrict__ sorted, uint32_t ordered) {
    extern __shared__ uint32_t cnt[];                 // 2^bin_shift
    __shared__ uint32_t wsum[MSM_BIN_THREADS / 64];
    const uint32_t bin = blockIdx.x, size = 1u << bin_shift, b0 = bin << bin_shift;
    const uint32_t lo = block_off[(size_t)bin * n_blocks], hi = block_off[(size_t)(bin + 1) * n_blocks];
    for (uint32_t i = threadIdx.x; i < size; i += MSM_BIN_THREADS) cnt[i] = 0;
    __syncthreads();
    // both sweeps read MSM_BIN_UNROLL pairs per thread before the first LDS atomic: a bin of 2^19 pairs (2^24 pairs per window) is
    // 512 dependent load -> atomic -> store rounds per thread otherwise (44 ms for the 1280 bins there; round 3)
    for (uint32_t k = lo + threadIdx.x; k < hi; k += MSM_BIN_THREADS * MSM_BIN_UNROLL) {
        uint32_t bx[MSM_BIN_UNROLL];
#pragma unroll
        for (int u = 0; u < MSM_BIN_UNROLL; u++) {
            const uint32_t kk = k + (uint32_t)u * MSM_BIN_THREADS;
            bx[u] = kk < hi ? part[kk].x : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < MSM_BIN_UNROLL; u++) if (bx[u] != 0xFFFFFFFFu) atomicAdd(&cnt[bx[u] - b0], 1u);
    }
    __syncthreads();
    // exclusive scan of cnt: each thread owns `per` consecutive counters
    const uint32_t per = size / MSM_BIN_THREADS > 0 ? size / MSM_BIN_THREADS : 1;
    const uint32_t first = threadIdx.x * per;
    uint32_t mine = 0;
    if (first < size) for (uint32_t k = 0; k < per; k++) mine += cnt[first + k];
    uint32_t incl = mine;                              // wave-inclusive scan, then across the 16 waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int k = 0; k < wave; k++) wbase += wsum[k];
    uint32_t run = lo + wbase + incl - mine;
    if (first < size) {
        for (uint32_t k = 0; k < per; k++) {
            const uint32_t c = cnt[first + k], b = b0 + first + k;
            if (b < total) { counts[b] = c; starts[b] = run; }
            cnt[first + k] = run;                     // becomes the bucket's cursor
            run += c;
        }
    }
    __syncthreads();
    // Second sweep, in the order of the partitioned array: a bin's pairs arrive tile by tile, i.e. by ascending entry number
    // e = w n + i, and 1024 consecutive pairs of a bin are about one window's worth -- so with a barrier after every 1024 pairs
    // each bucket's list comes out ordered by window (merged buckets) / by base index (per-window buckets).  The accumulation
    // walks its lists in step, so ordered lists keep the gathers of the waves in flight closer together: 1-2 % on the
    // accumulation and on the pipelined batch at 2^20 .. 2^24, and the sweep itself is no slower with the barriers
    // (profiles/r03_sort_order_ab.txt, GH_SORT_ORDERED=0 = without them).  The loads of eight steps are issued together.
    for (uint32_t base = lo; base < hi; base += MSM_BIN_THREADS * MSM_BIN_UNROLL) {      // uniform trip count: barriers inside
        uint2 pv[MSM_BIN_UNROLL];
#pragma unroll
        for (int u = 0; u < MSM_BIN_UNROLL; u++) {
            const uint32_t kk = base + (uint32_t)u * MSM_BIN_THREADS + threadIdx.x;
            pv[u] = kk < hi ? part[kk] : make_uint2(0xFFFFFFFFu, 0u);
        }
#pragma unroll
        for (int u = 0; u < MSM_BIN_UNROLL; u++) {
            if (pv[u].x != 0xFFFFFFFFu) sorted[atomicAdd(&cnt[pv[u].x - b0], 1u)] = pv[u].y;
            if (ordered) __syncthreads();       // (GH_SORT_ORDERED=0: the A/B switch)
        }
    }
}

// ---------------------------------------------------------------- 2b. bucket order by descending size
// size bin = min(count, heavy_thr + 1); bins are laid out so that larger sizes come first, i.e.
// order[0 .. n_heavy) are the heavy buckets (count > heavy_thr).  Bucket sizes cluster around the
// mean, so the bins are aggregated in LDS per block before touching the global counters.
static __device__ __forceinline__ uint32_t msm_size_bin(uint32_t cnt, uint32_t heavy_thr) {
    uint32_t bin = cnt > heavy_thr ? heavy_thr + 1 : cnt;
    return heavy_thr + 1 - bin;  // reversed: heavy -> 0, then sizes heavy_thr .. 0
}
static __global__ void __launch_bounds__(256)
msm_size_hist_kernel(const uint32_t* counts, size_t total, uint32_t heavy_thr, uint32_t* size_hist, uint32_t* max_count) {
    __shared__ uint32_t h[MSM_SIZE_BINS];
    __shared__ uint32_t bmax;
    for (int i = threadIdx.x; i < MSM_SIZE_BINS; i += 256) h[i] = 0;
    if (threadIdx.x == 0) bmax = 0;
    __syncthreads();
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < total) {
        const uint32_t c = counts[g];
        atomicAdd(&h[msm_size_bin(c, heavy_thr)], 1u);
        atomicMax(&bmax, c);      // the largest bucket (the affine rounds size their depth by it)
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MSM_SIZE_BINS; i += 256) if (h[i]) atomicAdd(&size_hist[i], h[i]);
    if (threadIdx.x == 0 && bmax) atomicMax(max_count, bmax);
}
static __global__ void __launch_bounds__(256)
msm_size_scatter_kernel(const uint32_t* counts, size_t total, uint32_t heavy_thr, uint32_t* size_cursor, uint32_t* order) {
    __shared__ uint32_t h[MSM_SIZE_BINS];
    for (int i = threadIdx.x; i < MSM_SIZE_BINS; i += 256) h[i] = 0;
    __syncthreads();
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bin = 0, rank = 0;
    if (g < total) { bin = msm_size_bin(counts[g], heavy_thr); rank = atomicAdd(&h[bin], 1u); }
    __syncthreads();
    for (int i = threadIdx.x; i < MSM_SIZE_BINS; i += 256) { uint32_t c = h[i]; if (c) h[i] = atomicAdd(&size_cursor[i], c); }
    __syncthreads();
    if (g < total) order[h[bin] + rank] = (uint32_t)g;
}

// plan[0] = n_heavy, plan[1] = total number of chunks, plan[2] = list entries; chunk_start[h] for h in [0, n_heavy]
static __global__ void msm_heavy_plan_kernel(const uint32_t* size_hist, const uint32_t* counts, const uint32_t* order,
                                             const uint32_t* starts, uint32_t total, uint32_t chunk, uint32_t* chunk_start,
                                             uint32_t* plan) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t n_heavy = size_hist[0];
    uint32_t run = 0;
    for (uint32_t h = 0; h < n_heavy; h++) {
        chunk_start[h] = run;
        run += (counts[order[h]] + chunk - 1) / chunk;
    }
    chunk_start[n_heavy] = run;
    plan[0] = n_heavy;
    plan[1] = run;
    plan[2] = starts[total - 1] + counts[total - 1];   // list entries = additions the accumulation will issue
    plan[3] = 0;
}

// ---------------------------------------------------------------- 3. scatter
static __global__ void __launch_bounds__(256)
msm_scatter_kernel(const int32_t* __restrict__ digits, size_t n, int num_windows, uint32_t win_stride,
                   uint32_t row_stride /* 0, or the table's row length (merged windows) */,
                   uint32_t* __restrict__ cursor /* = copy of starts */, uint32_t* __restrict__ sorted, int agg_iters,
                   uint32_t slot_shift, uint32_t sets) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int w = blockIdx.y;
    const int32_t d = i < n ? digits[(size_t)w * n + i] : 0;
    const uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
    const uint32_t pos = wave_agg_inc(cursor + (size_t)((uint32_t)w % sets) * win_stride, mag - slot_shift, d != 0, agg_iters);
    if (d != 0) sorted[pos] = ((uint32_t)i + ((uint32_t)w / sets) * row_stride) | (d < 0 ? 0x80000000u : 0u);
}

// ---------------------------------------------------------------- 4. bucket accumulation
// order[] lists bucket ids by descending size: [0, n_heavy) are heavy (msm_heavy_*_kernel),
// the rest is walked here one bucket per thread; empty buckets store infinity.
//
// The loop body holds exactly ONE mixed addition and no function call, so the kernel's register
// budget is its own.  The reference's `P == Q -> double` branch (swp.rs:492-495) is reached when a
// bucket's running sum equals the incoming base (duplicate bases); instead of a doubling formula
// the thread then takes a three-step detour through a fixed "salt" point S (S = G or 2G, whichever
// has x != q.x, so q != +-S):  acc <- ((q + S) + q) - S = 2q, each step a generic mixed addition.
// P + (-P) needs no branch: the formula yields Z = 0 and the next addition restarts from infinity.
//
// WAVES = minimum waves per SIMD the register allocation must allow (1: up to 512 VGPR+AGPR,
// 2: up to 256); selected at run time (GH_ACC_WAVES) for A/B measurements.
// AFFIN = true: the input is the output list of the affine rounds (aff_kernels.h): bucket g owns the records
// [starts[g], starts[g] + counts[g]) of `bases` directly (no index list, no signs) and infinity markers are skipped.
template <class C, int WAVES, bool AFFIN = false>
__global__ void __launch_bounds__(256, WAVES)
msm_accumulate_kernel(const Aff<C>* __restrict__ bases, const uint32_t* __restrict__ sorted,
                      const uint32_t* __restrict__ starts, const uint32_t* __restrict__ counts,
                      const uint32_t* __restrict__ order, uint32_t total,
                      const Aff<C>* __restrict__ salts, Proj<C>* __restrict__ buckets,
                      const uint32_t* __restrict__ chunk_start, uint32_t n_heavy, uint32_t n_chunks, uint32_t chunk,
                      Proj<C>* __restrict__ partials, uint32_t g_first = 0, uint32_t list_base = 0) {
    typedef typename C::F F;
    // task list: [0, n_chunks) chunks of the heavy buckets (the longest tasks, scheduled first),
    //            then the buckets order[n_heavy ..] by descending size
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chunks + (total - n_heavy)) return;
    uint32_t beg, cnt;
    Proj<C>* dst;
    if constexpr (AFFIN) {       // bucket g_first + t of a chunk of buckets; its points sit at starts[g] - list_base of the chunk's list
        const uint32_t g = g_first + t;
        beg = starts[g] - list_base; cnt = counts[g];
        dst = buckets + g;
    } else if (t >= n_chunks) {         // one whole bucket per thread
        const uint32_t g = order[n_heavy + (t - n_chunks)];
        beg = starts[g]; cnt = counts[g];
        dst = buckets + g;
    } else {                     // chunk t: a slice of `chunk` entries of heavy bucket order[h] -> partials[t]
        uint32_t lo = 0, hi = n_heavy;   // largest h with chunk_start[h] <= t
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (chunk_start[mid] <= t) lo = mid; else hi = mid; }
        const uint32_t g = order[lo], j = t - chunk_start[lo];
        beg = starts[g] + j * chunk;
        cnt = counts[g] - j * chunk;
        if (cnt > chunk) cnt = chunk;
        dst = partials + t;
    }
    // Y1 is needed at the start (u = y2 Z1 - Y1) and at the very end (Y3 = ... - vvv Y1) of an
    // addition; in between it is parked in LDS (G1 only, 26 KiB per block, word-major so lanes
    // hit distinct banks) -- that is the difference between fitting the 256-register budget and
    // spilling to scratch.
    constexpr bool PARK = (F::DEG == 1) && WAVES >= 2;
    __shared__ uint32_t park[PARK ? NL : 1][PARK ? 256 : 1];
    Proj<C> acc = proj_zero<C>();
    uint32_t k = 0;
    int phase = 0, salt_id = 0;     // phase 0: list entry k; 1: +S; 2: entry k again; 3: -S
    uint32_t guard = 0;
    while (k < cnt && guard < 4 * cnt + 8) {
        guard++;
        Aff<C> q;
        if (phase == 1 || phase == 3) {
            q = ld_aff<C>(salts + salt_id);
            if (phase == 3) q.y = F::neg(q.y);
        } else {
            if constexpr (AFFIN) {   // T64 list (aff_kernels.h): element e in tile e / 64, slot e % 64
                static_assert(!AFFIN || F::DEG == 1, "affine-round lists: prime-field curves");
                const uint32_t e = beg + k;
                F::comp(q.x, 0) = t64_ld_x(bases, e >> 6, e & 63u);
                F::comp(q.y, 0) = t64_ld_y(bases, e >> 6, e & 63u);
                if (phase == 0 && F::comp(q.x, 0).l[0] == AFF_MARK) { k++; continue; }   // a cancelled pair: nothing to add
            } else {
                const uint32_t e = sorted[beg + k];
                q = ld_aff<C>(bases + (e & 0x7FFFFFFFu));
                if (e >> 31) q.y = F::neg(q.y);
            }
        }
        if (proj_is_zero<C>(acc)) {
            acc.x = q.x; acc.y = q.y; acc.z = F::one();
        } else {
            // madd-1998-cmo (swp.rs:497-517)
            typename F::T v = F::mul(q.x, acc.z);
            typename F::T u = F::mul(q.y, acc.z);
            if (phase == 0 && F::eq(u, acc.y) && F::eq(v, acc.x)) {   // acc == q: take the detour
                salt_id = F::eq(q.x, ld_aff<C>(salts).x) ? 1 : 0;
                phase = 1;
                continue;
            }
            if constexpr (WAVES == 1) {
                // one wave per SIMD (GH_ACC_WAVES=1; not the default): no fences, independent products next to each other.
                // 2^20 pairs: 25.6 ms against 22.4 ms for the fenced order at two waves.  Explicitly interleaved product
                // pairs (two accumulator chains alternating statement by statement: 2.8 us per product in isolation,
                // tools/microbench/lone_wave.hip) need more than 256 live registers here and lose the gain to
                // AGPR copies (1.9 K v_accvgpr moves per addition): 27.3 ms.
                u = F::sub(u, acc.y);
                v = F::sub(v, acc.x);
                typename F::T vv = F::sqr(v), uu = F::sqr(u);
                typename F::T r = F::mul(vv, acc.x), vvv = F::mul(v, vv);
                typename F::T t = F::mul(uu, acc.z), z3 = F::mul(vvv, acc.z);
                typename F::T a = F::sub(F::sub(t, vvv), F::dbl(r));
                typename F::T x3 = F::mul(v, a), m1 = F::mul(vvv, acc.y);
                acc.y = F::sub(F::mul(u, F::sub(r, a)), m1);
                acc.x = x3;
                acc.z = z3;
            } else {
            // operation order chosen to keep at most seven field elements live (register budget 256);
            // the scheduling fences make hipcc keep that order instead of hoisting products
#if defined(__HIP_DEVICE_COMPILE__)
#define GH_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define GH_FENCE()
#endif
            u = F::sub(u, acc.y);
            v = F::sub(v, acc.x);
            if constexpr (PARK) {
                const uint32_t* yw = reinterpret_cast<const uint32_t*>(&acc.y);
#pragma unroll
                for (int w = 0; w < NL; w++) park[w][threadIdx.x] = yw[w];
            }
            GH_FENCE();
            typename F::T vv = F::sqr(v);
            GH_FENCE();
            typename F::T r = F::mul(vv, acc.x);        // acc.x dead
            GH_FENCE();
            typename F::T vvv = F::mul(v, vv);          // vv dead
            GH_FENCE();
            typename F::T uu = F::sqr(u);
            GH_FENCE();
            typename F::T a = F::sub(F::sub(F::mul(uu, acc.z), vvv), F::dbl(r));   // uu dead
            GH_FENCE();
            acc.x = F::mul(v, a);                       // v dead
            GH_FENCE();
            typename F::T rma = F::sub(r, a);           // r, a dead
            GH_FENCE();
            typename F::T y1 = acc.y;
            if constexpr (PARK) {
                uint32_t* yw = reinterpret_cast<uint32_t*>(&y1);
#pragma unroll
                for (int w = 0; w < NL; w++) yw[w] = park[w][threadIdx.x];
            }
            // (the dual product with one reduction, F::mul_sub_mul, was measured SLOWER here: its two
            //  accumulators cost 360 B more spills per addition; 31.2 ms vs 28.8 ms at 2^20)
            acc.y = F::sub(F::mul(u, rma), F::mul(vvv, y1));
            GH_FENCE();
            acc.z = F::mul(vvv, acc.z);
#undef GH_FENCE
            }
        }
        if (phase == 0 || phase == 3) { k++; phase = 0; } else phase++;
    }
    st_proj<C>(dst, acc);
}

// ---------------------------------------------------------------- 4-asm. task table of the assembly accumulation kernel
// The task decode of msm_accumulate_kernel / msm_accumulate_xyzz_kernel (chunks of the heavy buckets first, then every other
// bucket by descending size) as a table, so that the assembly kernel (asmgen/g1_xyzz.py) starts from (beg, cnt, dst).
struct AccTaskRec {
    uint32_t beg, cnt;
    uint64_t dst;
};
template <class C>
__global__ void __launch_bounds__(256)
msm_acc_tasks_kernel(const uint32_t* __restrict__ starts, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ order,
                     uint32_t total, Proj<C>* __restrict__ buckets, const uint32_t* __restrict__ chunk_start, uint32_t n_heavy,
                     uint32_t n_chunks, uint32_t chunk, Proj<C>* __restrict__ partials, AccTaskRec* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chunks + (total - n_heavy)) return;
    AccTaskRec r;
    if (t >= n_chunks) {
        const uint32_t g = order[n_heavy + (t - n_chunks)];
        r.beg = starts[g]; r.cnt = counts[g];
        r.dst = (uint64_t)(uintptr_t)(buckets + g);
    } else {
        uint32_t lo = 0, hi = n_heavy;   // largest h with chunk_start[h] <= t
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (chunk_start[mid] <= t) lo = mid; else hi = mid; }
        const uint32_t g = order[lo], j = t - chunk_start[lo];
        r.beg = starts[g] + j * chunk;
        uint32_t cnt = counts[g] - j * chunk;
        r.cnt = cnt > chunk ? chunk : cnt;
        r.dst = (uint64_t)(uintptr_t)(partials + t);
    }
    out[t] = r;
}

// ---------------------------------------------------------------- 4a. G1 bucket accumulation on XYZZ accumulators
// The same task list, the same list walk and the same salt detour as msm_accumulate_kernel, with the running sum in
// extended Jacobian coordinates (ec29.h, Xyzz): madd-2008-s, 8 M + 2 S, and Y3 as one dual product on a single
// accumulator chain (fp_mul2s) -- 12 194 v_mad_u64_u32 per bucket update where madd-1998-cmo issues 14 226.
// Register plan (256 VGPRs, two waves per SIMD): ZZ and ZZZ stay in registers; X and Y of the running sum live in LDS
// (word-major, 2 x 26 KiB per block) and are read where the formula needs them (X: P and Q; Y: R and Y3), so the loop
// never holds more than six field elements next to a product's own m / r arrays.
// Output: the projective image (X ZZZ : Y ZZ : ZZ ZZZ) the bucket reduction expects.
template <class C, bool AFFIN = false>
__global__ void __launch_bounds__(256, 2)
msm_accumulate_xyzz_kernel(const Aff<C>* __restrict__ bases, const uint32_t* __restrict__ sorted,
                           const uint32_t* __restrict__ starts, const uint32_t* __restrict__ counts,
                           const uint32_t* __restrict__ order, uint32_t total,
                           const Aff<C>* __restrict__ salts, Proj<C>* __restrict__ buckets,
                           const uint32_t* __restrict__ chunk_start, uint32_t n_heavy, uint32_t n_chunks, uint32_t chunk,
                           Proj<C>* __restrict__ partials, uint32_t g_first = 0, uint32_t list_base = 0) {
    typedef typename C::PF P;
    static_assert(C::F::DEG == 1, "XYZZ accumulation kernel: prime-field curves");
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chunks + (total - n_heavy)) return;
    uint32_t beg, cnt;
    Proj<C>* dst;
    if constexpr (AFFIN) {
        const uint32_t g = g_first + t;
        beg = starts[g] - list_base; cnt = counts[g];
        dst = buckets + g;
    } else if (t >= n_chunks) {
        const uint32_t g = order[n_heavy + (t - n_chunks)];
        beg = starts[g]; cnt = counts[g];
        dst = buckets + g;
    } else {
        uint32_t lo = 0, hi = n_heavy;   // largest h with chunk_start[h] <= t
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (chunk_start[mid] <= t) lo = mid; else hi = mid; }
        const uint32_t g = order[lo], j = t - chunk_start[lo];
        beg = starts[g] + j * chunk;
        cnt = counts[g] - j * chunk;
        if (cnt > chunk) cnt = chunk;
        dst = partials + t;
    }
    __shared__ uint32_t park[2][NL][256];     // [0]: X, [1]: Y of this thread's running sum
    auto put = [&](int s, const Fp& v) {
#pragma unroll
        for (int w = 0; w < NL; w++) park[s][w][threadIdx.x] = v.l[w];
    };
    auto get = [&](int s) {
        Fp v;
#pragma unroll
        for (int w = 0; w < NL; w++) v.l[w] = park[s][w][threadIdx.x];
        return v;
    };
#define GH_FENCE() __builtin_amdgcn_sched_barrier(0)
    Fp zz = fp_zero(), zzz = fp_zero();
    // The parked ordinate is V = sigma Y, sigma = -1 while `sneg`: the dual product below then needs no negated operand --
    // with W = sigma R = (sigma q.y) ZZZ - V:  W (X3 - Q) + V PPP = -sigma Y3, so sigma flips with every update and is absorbed
    // by the sign the incoming point gets anyway (one 26-limb negation less per update).
    bool sneg = false;
    uint32_t k = 0;
    int phase = 0, salt_id = 0;     // phase 0: list entry k; 1: +S; 2: entry k again; 3: -S
    uint32_t guard = 0;
    while (k < cnt && guard < 4 * cnt + 8) {
        guard++;
        Fp qx, qy;
        if (phase == 1 || phase == 3) {
            const Aff<C> s = ld_aff<C>(salts + salt_id);
            qx = s.x; qy = ((phase == 3) != sneg) ? fp_neg<P>(s.y) : s.y;
        } else {
            if constexpr (AFFIN) {
                const uint32_t e = beg + k;
                qx = t64_ld_x(bases, e >> 6, e & 63u);
                qy = t64_ld_y(bases, e >> 6, e & 63u);
                if (phase == 0 && qx.l[0] == AFF_MARK) { k++; continue; }   // a cancelled pair: nothing to add
                if (sneg) qy = fp_neg<P>(qy);
            } else {
                const uint32_t e = sorted[beg + k];
                const Aff<C> b = ld_aff<C>(bases + (e & 0x7FFFFFFFu));
                qx = b.x; qy = (((e >> 31) != 0) != sneg) ? fp_neg<P>(b.y) : b.y;
            }
        }
        if (fp_is_zero(zz)) {
            put(0, qx); put(1, qy);
            zz = fp_one<P>(); zzz = zz;
        } else {
            Fp pp = fp_mul<P>(qx, zz);                              // U2
            GH_FENCE();
            Fp r = fp_mul<P>(qy, zzz);                              // S2
            GH_FENCE();
            pp = fp_sub<P>(pp, get(0));                             // P = U2 - X1
            r = fp_sub<P>(r, get(1));                               // W = sigma (S2 - Y1) = sigma R
            if (phase == 0 && fp_is_zero(pp) && fp_is_zero(r)) {    // acc == q: the detour (swp.rs:492-495 doubles here)
                salt_id = fp_eq(qx, ld_aff<C>(salts).x) ? 1 : 0;
                phase = 1;
                continue;
            }
            GH_FENCE();
            Fp p2 = fp_sqr<P>(pp);                                  // PP
            GH_FENCE();
            zz = fp_mul<P>(zz, p2);                                 // ZZ3
            GH_FENCE();
            pp = fp_mul<P>(pp, p2);                                 // PPP
            GH_FENCE();
            zzz = fp_mul<P>(zzz, pp);                               // ZZZ3
            GH_FENCE();
            p2 = fp_mul<P>(get(0), p2);                             // Q = X1 PP
            GH_FENCE();
            Fp x3 = fp_sub<P>(fp_sub<P>(fp_sqr<P>(r), pp), fp_dbl<P>(p2));
            put(0, x3);
            GH_FENCE();
            p2 = fp_sub<P>(x3, p2);                                 // X3 - Q
            GH_FENCE();
            put(1, fp_mul2s<P>(r, p2, get(1), pp));                 // W (X3 - Q) + V PPP = -sigma Y3: the new V, sigma flips
            sneg = !sneg;
            GH_FENCE();
        }
        if (phase == 0 || phase == 3) { k++; phase = 0; } else phase++;
    }
#undef GH_FENCE
    Proj<C> out = proj_zero<C>();
    if (!fp_is_zero(zz)) {
        out.x = fp_mul<P>(get(0), zzz);
        out.y = fp_mul<P>(sneg ? fp_neg<P>(get(1)) : get(1), zz);
        out.z = fp_mul<P>(zz, zzz);
    }
    st_proj<C>(dst, out);
}

// ---------------------------------------------------------------- 4b. G2 over Fq2: lane-pair formulation
// An Fq2 element (c0, c1) lives in TWO adjacent lanes: c0 in the even lane, c1 in the odd one, so a
// G2 point costs each lane the registers of a G1 point and the kernel needs neither out-of-line
// products nor scratch (the call-based Fq2 kernel moved ~16 KB of scratch per mixed addition
// and was bandwidth bound).  Products use the schoolbook split, two Fp products per lane:
//   even lane: c0 = a0 b0 + NR a1 b1        odd lane: c1 = a1 b0 + a0 b1
// with the partner's coefficients fetched by DPP quad permutes (26 moves per operand: swap, broadcast of either half).
// 22 Fp-product times per mixed addition and lane pair, against 31 on one lane -- but in registers.
// DUAL = true: one dual product with a single reduction per lane (2028 mads, 4 operands + 2 accumulators live:
// 512 registers, 1 wave / SIMD); DUAL = false: two plain products per lane (2704 mads, the register footprint
// of the G1 kernel: 256 registers, 2 waves / SIMD -- what the VALU needs to be kept busy).
template <class P, int NR, bool DUAL = true> struct F2S {
    typedef Fp T;
    static constexpr int DEG = 1;   // per-lane footprint
    static constexpr int LANES = 2;
    static constexpr int WAVES = DUAL ? 1 : 2;
    static __device__ __forceinline__ bool odd() { return (threadIdx.x & 1u) != 0; }
    static __device__ __forceinline__ T swap(const T& a) {
        T r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[i], 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        return r;                                                     // (__shfl_xor(x, 1) compiles to ds_bpermute_b32: LDS round trips)
    }
    static __device__ __forceinline__ T sel(bool c, const T& x, const T& y) {   // c ? x : y
        T r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = c ? x.l[i] : y.l[i];
        return r;
    }
    static __device__ __forceinline__ T zero() { return fp_zero(); }
    static __device__ __forceinline__ T one() { return odd() ? fp_zero() : fp_one<P>(); }
    static __device__ __forceinline__ T add(const T& a, const T& b) { return fp_add<P>(a, b); }
    static __device__ __forceinline__ T sub(const T& a, const T& b) { return fp_sub<P>(a, b); }
    static __device__ __forceinline__ T dbl(const T& a) { return fp_dbl<P>(a); }
    static __device__ __forceinline__ T neg(const T& a) { return fp_neg<P>(a); }
    // coefficient PART (0 / 1) of b in both lanes of the pair: one DPP move per limb (quad_perm [0,0,2,2] / [1,1,3,3])
    template <int PART> static __device__ __forceinline__ T bcast(const T& b) {
        T r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)b.l[i], PART ? 0xF5 : 0xA0, 0xF, 0xF, true);
        return r;
    }
    static __device__ __forceinline__ T mul(const T& a, const T& b) {
        const bool o = odd();
        // per lane:  even: a0 b0 + (NR a1) b1      odd: a1 b0 + a0 b1
        // = own a times b0, the partner's a (times NR on the even lane: fp_mul_small_rt with k = NR / 1) times b1
        const T ao = fp_mul_small_rt<P>(swap(a), o ? 1u : (uint32_t)NR), b0 = bcast<0>(b), b1 = bcast<1>(b);
        if constexpr (DUAL) {
            return fp_mul2<P>(a, b0, ao, b1);
        } else {
            return fp_add<P>(fp_mul<P>(a, b0), fp_mul<P>(ao, b1));
        }
    }
    static __device__ __forceinline__ T mul_sub_mul(const T& a, const T& b, const T& c, const T& d) { return sub(mul(a, b), mul(c, d)); }
    static __device__ __forceinline__ T sqr(const T& a) { return mul(a, a); }
    // (the dual product takes fully reduced operands only: (3 p^2 + R p) / R > 2p with a lazy one)
    static __device__ __forceinline__ T sub_lazy(const T& a, const T& b) { return fp_sub<P>(a, b); }
    // (a0 + a1 X)^-1 = (a0 - a1 X) / (a0^2 - NR a1^2)     (fp2.rs inverse): one Fp inversion per lane pair, run by both lanes
    static __device__ __forceinline__ T inv(const T& a) {
        const bool o = odd();
        const T s = fp_sqr<P>(a), t = swap(s);                       // own square, the partner's
        const T n = o ? fp_sub<P>(t, fp_mul_small<P, NR>(s)) : fp_sub<P>(s, fp_mul_small<P, NR>(t));
        const T r = fp_mul<P>(a, fp_inv<P>(n));
        return o ? fp_neg<P>(r) : r;
    }
    static __device__ __forceinline__ bool is_zero(const T& a) {
        const int z = fp_is_zero(a) ? 1 : 0;
        return (z & __shfl_xor(z, 1)) != 0;
    }
    static __device__ __forceinline__ bool eq(const T& a, const T& b) {
        const int z = fp_eq(a, b) ? 1 : 0;
        return (z & __shfl_xor(z, 1)) != 0;
    }
};

// Fq3 = Fp[X]/(X^3 - NR) over lane triples (lanes 3g, 3g+1, 3g+2 hold c0, c1, c2; lane 63 of a wave
// idles).  Schoolbook, three Fp products per lane:
//   c_j = sum_{m <= j} a_(j-m) b_m + NR sum_{m > j} a_(j-m+3) b_m
// TRIPLE = 0: three plain products in a rolled loop (4056 mads, 2 waves / SIMD); 1: ONE triple product with a single
// reduction (fp_mul3: 2704 mads) inlined at every site (1 wave / SIMD; hipcc does not get through it); 2: that triple
// product as one out-of-line function (GH_F3S_CALL_WAVES waves / SIMD) -- the default, see msm_impl.h.
// The tower product of a lane triple as ONE out-of-line device function (F3S mode 2): lane rotations, the two NR multiples,
// the operand selection and the triple product with its single reduction.  One body per kernel instead of one per
// product site, which is what lets hipcc get through the kernels at all (mode 1, the same code inlined at every site,
// had not compiled after an hour).  Two Fp arguments and an Fp result travel in VGPRs / on the stack.
#ifndef GH_F3S_CALL_WAVES
#define GH_F3S_CALL_WAVES 1
#endif
template <class P, int NR> __device__ __attribute__((noinline)) Fp f3s_mul_outlined(Fp a, Fp b) {
    const int lane = threadIdx.x & 63, j = lane % 3, base = lane - j;
    const int s1 = base + (j + 1) % 3, s2 = base + (j + 2) % 3;
    // c_j = a_j b_0 + [NR if j = 0] a_(j-1) b_1 + [NR if j < 2] a_(j-2) b_2     (indices mod 3): b_0, b_1, b_2 are the same
    // for the three lanes (broadcasts within the triple); the NR multiples by fp_mul_small_rt with k = NR / 1 per lane
    Fp an, ap, y1, y2, y3;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        an.l[i] = (uint32_t)__shfl((int)a.l[i], s1); ap.l[i] = (uint32_t)__shfl((int)a.l[i], s2);
        y1.l[i] = (uint32_t)__shfl((int)b.l[i], base); y2.l[i] = (uint32_t)__shfl((int)b.l[i], base + 1);
        y3.l[i] = (uint32_t)__shfl((int)b.l[i], base + 2);
    }
    const Fp x2 = fp_mul_small_rt<P>(ap, j == 0 ? (uint32_t)NR : 1u), x3 = fp_mul_small_rt<P>(an, j < 2 ? (uint32_t)NR : 1u);
    return fp_mul3<P>(a, y1, x2, y2, x3, y3);
}

template <class P, int NR, int TRIPLE = 0> struct F3S {
    typedef Fp T;
    static constexpr int DEG = 1;
    static constexpr int LANES = 3;
    static constexpr int WAVES = TRIPLE == 1 ? 1 : (TRIPLE == 2 ? GH_F3S_CALL_WAVES : 2);
    static __device__ __forceinline__ int comp() { return (int)((threadIdx.x & 63u) % 3u); }
    static __device__ __forceinline__ T rot(const T& a, int by) {   // coefficient held by lane (comp + by) mod 3 of this triple
        const int lane = threadIdx.x & 63, j = lane % 3, src = lane - j + (j + by) % 3;
        T r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = (uint32_t)__shfl((int)a.l[i], src);
        return r;
    }
    static __device__ __forceinline__ T sel3(int j, const T& x0, const T& x1, const T& x2) {
        T r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = j == 0 ? x0.l[i] : (j == 1 ? x1.l[i] : x2.l[i]);
        return r;
    }
    static __device__ __forceinline__ T zero() { return fp_zero(); }
    static __device__ __forceinline__ T one() { return comp() == 0 ? fp_one<P>() : fp_zero(); }
    static __device__ __forceinline__ T add(const T& a, const T& b) { return fp_add<P>(a, b); }
    static __device__ __forceinline__ T sub(const T& a, const T& b) { return fp_sub<P>(a, b); }
    static __device__ __forceinline__ T dbl(const T& a) { return fp_dbl<P>(a); }
    static __device__ __forceinline__ T neg(const T& a) { return fp_neg<P>(a); }
    // The three products of a lane run in a ROLLED loop (one fp_mul body per call site): with them
    // unrolled the kernel held 33 inlined products and hipcc needed more than half an hour for it.
    //   iteration m:  lane j takes a_((j - m) mod 3) * b_m, times NR when m > j (the wrapped terms)
    static __device__ __forceinline__ T mul(const T& a, const T& b) {
        if constexpr (TRIPLE == 2) return f3s_mul_outlined<P, NR>(a, b);
        if constexpr (TRIPLE == 1) {
            // c_j = a_j b_0 + [NR if j = 0] a_(j-1) b_1 + [NR if j < 2] a_(j-2) b_2     (indices mod 3)
            const int j = comp();
            const T an = rot(a, 1), ap = rot(a, 2), bn = rot(b, 1), bp = rot(b, 2);
            const T apn = fp_mul_small<P, NR>(ap), ann = fp_mul_small<P, NR>(an);
            T x2, x3;
#pragma unroll
            for (int i = 0; i < NL; i++) { x2.l[i] = j == 0 ? apn.l[i] : ap.l[i]; x3.l[i] = j < 2 ? ann.l[i] : an.l[i]; }
            return fp_mul3<P>(a, sel3(j, b, bp, bn), x2, sel3(j, bn, b, bp), x3, sel3(j, bp, bn, b));
        }
        const int lane = threadIdx.x & 63, j = lane % 3, base = lane - j;
        T acc = fp_zero();
#pragma nounroll
        for (int m = 0; m < 3; m++) {
            const int ja = j - m < 0 ? j - m + 3 : j - m;
            T x, y;
#pragma unroll
            for (int i = 0; i < NL; i++) {
                x.l[i] = (uint32_t)__shfl((int)a.l[i], base + ja);
                y.l[i] = (uint32_t)__shfl((int)b.l[i], base + m);
            }
            T t = fp_mul<P>(x, y);
            const T tn = fp_mul_small<P, NR>(t);
            const bool wrap = m > j;
#pragma unroll
            for (int i = 0; i < NL; i++) t.l[i] = wrap ? tn.l[i] : t.l[i];
            acc = fp_add<P>(acc, t);
        }
        return acc;
    }
    static __device__ __forceinline__ T sqr(const T& a) { return mul(a, a); }
    static __device__ __forceinline__ T mul_sub_mul(const T& a, const T& b, const T& c, const T& d) { return sub(mul(a, b), mul(c, d)); }
    static __device__ __forceinline__ T sub_lazy(const T& a, const T& b) { return fp_sub<P>(a, b); }   // triple / rolled products: reduced operands
    // norm-based inverse in Fp[X]/(X^3 - NR) (fp3.rs inverse), one Fp inversion per lane triple:
    //   c0 = a0^2 - NR a1 a2,  c1 = NR a2^2 - a0 a1,  c2 = a1^2 - a0 a2,  n = a0 c0 + NR (a2 c1 + a1 c2),  a^-1 = c / n
    // lane j holds a_j: it forms its own square and the product a_j a_(j+1), the rest travels by lane rotation.
    static __device__ __forceinline__ T inv(const T& a) {
        const int j = comp();
        const T a1 = rot(a, 1), a2 = rot(a, 2);                      // a_(j+1), a_(j+2)
        const T sq = fp_sqr<P>(a), pr = fp_mul<P>(a, a1);            // a_j^2, a_j a_(j+1):  pr = (a0 a1, a1 a2, a2 a0)
        const T sq1 = rot(sq, 1), sq2 = rot(sq, 2), pr1 = rot(pr, 1), pr2 = rot(pr, 2);
        // c_j:  j = 0: sq_0 - NR pr_1     j = 1: NR sq_2 - pr_0     j = 2: sq_1 - pr_2
        const T u = j == 0 ? sq : (j == 1 ? fp_mul_small<P, NR>(sq1) : sq2);      // lane 1: sq_2 = sq_(j+1); lane 2: sq_1 = sq_(j+2)
        const T v = j == 0 ? fp_mul_small<P, NR>(pr1) : (j == 1 ? pr2 : pr);      // lane 0: pr_1; lane 1: pr_0 = pr_(j+2); lane 2: pr_2 = own
        const T c = fp_sub<P>(u, v);
        // n = a0 c0 + NR (a2 c1 + a1 c2): lane j multiplies c_j by a_0, a_2, a_1
        const T q = fp_mul<P>(j == 0 ? a : (j == 1 ? a1 : a2), c);  // lane 1: a_2 = a_(j+1); lane 2: a_1 = a_(j+2)
        const T q1 = rot(q, 1), q2 = rot(q, 2);
        const T q0v = j == 0 ? q : (j == 1 ? q2 : q1);              // q_0 as seen from lane j
        const T qs = j == 0 ? fp_add<P>(q1, q2) : (j == 1 ? fp_add<P>(q, q1) : fp_add<P>(q2, q));   // q_1 + q_2
        const T n = fp_add<P>(q0v, fp_mul_small<P, NR>(qs));
        return fp_mul<P>(c, fp_inv<P>(n));
    }
    static __device__ __forceinline__ bool all3(bool z) {
        const int lane = threadIdx.x & 63, j = lane % 3, b = lane - j;
        const int v = z ? 1 : 0;
        return (__shfl(v, b) & __shfl(v, b + 1) & __shfl(v, b + 2)) != 0;
    }
    static __device__ __forceinline__ bool is_zero(const T& a) { return all3(fp_is_zero(a)); }
    static __device__ __forceinline__ bool eq(const T& a, const T& b) { return all3(fp_eq(a, b)); }
};

// Same task list and addition as msm_accumulate_kernel, LANES lanes per task (2: Fq2 pairs, 3: Fq3
// triples; a wave carries 64 / LANES tasks, the remaining lane of a triple wave idles).
#ifndef GH_SPLIT_WAVES
#define GH_SPLIT_WAVES 1   // measured on Fq2 (twice): 119 ms at 1 wave/SIMD (512 registers) vs 135 ms at 2 (1.5 KB of spills), 2^20 pairs
#endif
// AFFIN: as for msm_accumulate_kernel -- the input is the T64 output list of the affine rounds.
template <class C, class F, int LANES, bool AFFIN = false>
__global__ void __launch_bounds__(256, F::WAVES)
msm_accumulate_split_kernel(const Aff<C>* __restrict__ bases, const uint32_t* __restrict__ sorted,
                           const uint32_t* __restrict__ starts, const uint32_t* __restrict__ counts,
                           const uint32_t* __restrict__ order, uint32_t total,
                           const Aff<C>* __restrict__ salts, Proj<C>* __restrict__ buckets,
                           const uint32_t* __restrict__ chunk_start, uint32_t n_heavy, uint32_t n_chunks, uint32_t chunk,
                           Proj<C>* __restrict__ partials, uint32_t g_first = 0, uint32_t list_base = 0) {
    constexpr uint32_t TPW = 64 / LANES;                       // tasks per wave
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t t = wave * TPW + lane / LANES;
    const int comp = (int)(lane % LANES);
    const uint32_t ntasks = n_chunks + (total - n_heavy);
    const bool live = lane < TPW * LANES && t < ntasks;       // all lanes of a group agree
    uint32_t beg = 0, cnt = 0;
    Proj<C>* dst = buckets;
    if (live) {
        if constexpr (AFFIN) {
            const uint32_t g = g_first + t;
            beg = starts[g] - list_base; cnt = counts[g];
            dst = buckets + g;
        } else if (t >= n_chunks) {
            const uint32_t g = order[n_heavy + (t - n_chunks)];
            beg = starts[g]; cnt = counts[g];
            dst = buckets + g;
        } else {
            uint32_t lo = 0, hi = n_heavy;
            while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (chunk_start[mid] <= t) lo = mid; else hi = mid; }
            const uint32_t g = order[lo], j = t - chunk_start[lo];
            beg = starts[g] + j * chunk;
            cnt = counts[g] - j * chunk;
            if (cnt > chunk) cnt = chunk;
            dst = partials + t;
        }
    }
    __shared__ uint32_t park[NL][256];
    // this lane's coefficient: element e of an Aff / Proj is {c0, .., c_(LANES-1)} -> Fp index LANES e + comp
    auto ld_comp = [&](const void* base, int e) { return ld_fp(reinterpret_cast<const Fp*>(base) + LANES * e + comp); };
    Fp ax = fp_zero(), ay = F::one(), az = fp_zero();   // (0, 1, 0)
    uint32_t k = 0;
    int phase = 0, salt_id = 0;
    uint32_t guard = 0;
    // the loop condition is uniform within a lane group (its lanes share cnt, k, phase)
    while (k < cnt && guard < 4 * cnt + 8) {
        guard++;
        Fp qx, qy;
        if (phase == 1 || phase == 3) {
            qx = ld_comp(salts + salt_id, 0);
            qy = ld_comp(salts + salt_id, 1);
            if (phase == 3) qy = F::neg(qy);
        } else if constexpr (AFFIN) {
            const uint32_t e = beg + k;
            const size_t tile = e / TPW;
            const uint32_t slot = (e % TPW) * LANES + (uint32_t)comp;
            qx = t64_ld_x(bases, tile, slot);
            qy = t64_ld_y(bases, tile, slot);
            if (phase == 0 && k < cnt && qx.l[0] == AFF_MARK) { k++; continue; }   // a cancelled pair (every lane of the group sees the marker)
        } else {
            const uint32_t e = sorted[beg + k];
            const Aff<C>* b = bases + (e & 0x7FFFFFFFu);
            qx = ld_comp(b, 0);
            qy = ld_comp(b, 1);
            if (e >> 31) qy = F::neg(qy);
        }
        if (F::is_zero(az)) {
            ax = qx; ay = qy; az = F::one();
        } else {
            Fp v = F::mul(qx, az);
            Fp u = F::mul(qy, az);
            if (phase == 0 && F::eq(u, ay) && F::eq(v, ax)) {
                salt_id = F::eq(qx, ld_comp(salts, 0)) ? 1 : 0;
                phase = 1;
                continue;
            }
            u = F::sub(u, ay);
            v = F::sub(v, ax);
#pragma unroll
            for (int w = 0; w < NL; w++) park[w][threadIdx.x] = ay.l[w];
            __builtin_amdgcn_sched_barrier(0);
            Fp vv = F::sqr(v);
            __builtin_amdgcn_sched_barrier(0);
            Fp r = F::mul(vv, ax);
            __builtin_amdgcn_sched_barrier(0);
            Fp vvv = F::mul(v, vv);
            __builtin_amdgcn_sched_barrier(0);
            Fp uu = F::sqr(u);
            __builtin_amdgcn_sched_barrier(0);
            Fp a = F::sub(F::sub(F::mul(uu, az), vvv), F::dbl(r));
            __builtin_amdgcn_sched_barrier(0);
            ax = F::mul(v, a);
            __builtin_amdgcn_sched_barrier(0);
            Fp t1 = F::mul(u, F::sub(r, a));
            __builtin_amdgcn_sched_barrier(0);
            Fp y1;
#pragma unroll
            for (int w = 0; w < NL; w++) y1.l[w] = park[w][threadIdx.x];
            ay = F::sub(t1, F::mul(vvv, y1));
            __builtin_amdgcn_sched_barrier(0);
            az = F::mul(vvv, az);
        }
        if (phase == 0 || phase == 3) { k++; phase = 0; } else phase++;
    }
    if (live) {
        Fp* o = reinterpret_cast<Fp*>(dst);
        st_fp(o + 0 * LANES + comp, ax);
        st_fp(o + 1 * LANES + comp, ay);
        st_fp(o + 2 * LANES + comp, az);
    }
}

// wave-level sum of one projective point per lane through LDS; result valid in lane 0.
// sh must hold 64 Proj<C>.  All 64 lanes must call.
template <class C>
__device__ __forceinline__ Proj<C> wave_tree_sum(Proj<C> v, Proj<C>* sh, int lane) {
    for (int off = 32; off > 0; off >>= 1) {
        if (lane >= off && lane < 2 * off) st_proj<C>(sh + lane, v);
        __syncthreads();
        if (lane < off) v = proj_add_call<C>(v, ld_proj<C>(sh + lane + off));
        __syncthreads();
    }
    return v;
}

// Heavy buckets (skewed scalars, sparsely populated top windows) are cut into chunks of heavy_thr
// entries that the accumulation kernel sums like ordinary buckets (chunk mode above, into
// partials[]); this kernel then adds the chunk sums of each heavy bucket: lanes stride through
// them, then a tree through LDS.
template <class C>
__global__ void __launch_bounds__(64, 2)
msm_heavy_combine_kernel(const Proj<C>* __restrict__ partials, const uint32_t* __restrict__ order,
                         const uint32_t* __restrict__ chunk_start, Proj<C>* __restrict__ buckets) {
    extern __shared__ uint32_t lds_raw[];
    Proj<C>* sh = reinterpret_cast<Proj<C>*>(lds_raw);
    const int lane = threadIdx.x;
    const uint32_t h = blockIdx.x;
    const uint32_t beg = chunk_start[h], end = chunk_start[h + 1];
    Proj<C> acc = proj_zero<C>();
    for (uint32_t k = beg + lane; k < end; k += 64) acc = proj_add_call<C>(acc, ld_proj<C>(partials + k));
    acc = wave_tree_sum<C>(acc, sh, lane);
    if (lane == 0) st_proj<C>(buckets + order[h], acc);
}

// ---------------------------------------------------------------- 5. bucket reduction
// sum_b b * B_b per window, without the reference's per-window inversion (variable_base.rs:60-66)
// and without any doubling or function call on the device.
//
// msm_wave_reduce_kernel is a "wave program": one wave per segment of 64 * L consecutive items of
// one window; lane l owns items l, l + 64, l + 128, ... (stride 64).  Every step of the program is
// one projective addition issued from a SINGLE inlined call site (operands are selected per
// step; the earlier call-based version moved ~7 KB of scratch per addition and was
// scratch-bandwidth bound).  The program's accumulators are parked in a global slab between steps
// (ReduceSlab below):
//   steps 0 .. 2L-2   serial:  run += item_i  (i = L-1 .. 0),  wacc += run      -> run_l = sum_i x,
//                                                                                   wacc_l = sum_i i * x
//   6 steps           tree over lanes of wacc                    -> A  = sum_l wacc_l
//   6 steps           suffix scan over lanes of run              -> S_l = sum_{m >= l} run_m;  runW = S_0
//   6 steps           tree over lanes l >= 1 of S                -> Bv = sum_l l * run_l
// With item index = l + 64 i:   sum_items index * x = 64 * A + Bv,  sum_items x = runW.
// mode 1 (plain sum) stops after the serial part and a tree over run.
// mode 2 ("lean" level 1, msm_impl.h) stops after the serial part and stores every lane's (run_l, wacc_l) -- out[(program * 64
// + l) * 2 + {0, 1}] -- for a second level that works on LANES instead of segments: the 18 cross-lane steps, in which most
// lanes idle, are then issued once per window instead of once per segment.
// Equal operands (acc == x as points, the reference's doubling branch) are detected in the
// addition; the whole wave then spends three extra steps on a detour through a salt point
// (p + S) + q - S for the affected lanes.  Powers of two (64, 64 L) that weight the outputs are
// NOT applied here: they are folded into the host's Horner loop over the windows, where the
// doublings are needed anyway (msm_impl.h: fold_windows).
template <class C> struct WaveReduceIn {
    const Proj<C>* base;   // item (w, k) = base[(w * count + k) * stride + offset]
    uint32_t stride, offset, count, mode;
    uint32_t valid;        // items with flat index w * count + k >= valid are padding (infinity)
};

// branch-free projective addition with selects for the infinity cases; same = (p == q as points)
template <class C> struct ReduceField { typedef typename C::F type; };          // G1: inlined products
template <> struct ReduceField<Mnt4G2> { typedef Mnt4G2::FC type; };                 // towers: out of line (code size)
template <> struct ReduceField<Mnt6G2> { typedef Mnt6G2::FC type; };
template <class C> __device__ __forceinline__ Proj<C> proj_add_sel(const Proj<C>& p, const Proj<C>& q, bool& same) {
    typedef typename ReduceField<C>::type F;
    const bool pz = F::is_zero(p.z), qz = F::is_zero(q.z);
    typename F::T y1z2 = F::mul(p.y, q.z);
    typename F::T x1z2 = F::mul(p.x, q.z);
    typename F::T z1z2 = F::mul(p.z, q.z);
    typename F::T u = F::sub(F::mul(p.z, q.y), y1z2);
    typename F::T v = F::sub(F::mul(p.z, q.x), x1z2);
    same = !pz && !qz && F::is_zero(u) && F::is_zero(v);
    typename F::T uu = F::sqr(u);
    typename F::T vv = F::sqr(v);
    typename F::T vvv = F::mul(v, vv);
    typename F::T r = F::mul(vv, x1z2);
    typename F::T a = F::sub(F::sub(F::mul(uu, z1z2), vvv), F::dbl(r));
    Proj<C> o;
    o.x = F::mul(v, a);
    o.y = F::mul_sub_mul(F::sub(r, a), u, vvv, y1z2);
    o.z = F::mul(vvv, z1z2);
    uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
    const uint32_t* pw = reinterpret_cast<const uint32_t*>(&p);
    const uint32_t* qw = reinterpret_cast<const uint32_t*>(&q);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(Proj<C>) / 4); k++) ow[k] = pz ? qw[k] : (qz ? pw[k] : ow[k]);
    return o;
}

// A block may carry blockDim.x / 64 INDEPENDENT waves (each its own program and LDS region, so the
// exchanges need wave-level ordering only, no s_barrier).  Measured at 2^20 buckets: 1, 2, 3 or 4
// waves per block, with or without a block barrier per step, all take the same time for level 1
// -- the default is 1.
#define GH_WAVE_SYNC()                                           \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)
// The same addition WITHOUT the final selects: p and q are dead after the first five products, which is what lets the step
// fit the register budget; the caller patches the lanes with an infinite operand (pz / qz) from re-loaded operands.
template <class C> __device__ __forceinline__ Proj<C> proj_add_raw(const Proj<C>& p, const Proj<C>& q, bool& same, bool& pz, bool& qz) {
    typedef typename ReduceField<C>::type F;
#define GH_RFENCE() __builtin_amdgcn_sched_barrier(0)      // keep the written order: at most eight field elements live
    pz = F::is_zero(p.z); qz = F::is_zero(q.z);
    typename F::T y1z2 = F::mul(p.y, q.z);
    GH_RFENCE();
    typename F::T u = F::sub(F::mul(p.z, q.y), y1z2);
    GH_RFENCE();
    typename F::T x1z2 = F::mul(p.x, q.z);
    GH_RFENCE();
    typename F::T v = F::sub(F::mul(p.z, q.x), x1z2);
    GH_RFENCE();
    typename F::T z1z2 = F::mul(p.z, q.z);                 // p, q dead
    GH_RFENCE();
    same = !pz && !qz && F::is_zero(u) && F::is_zero(v);
    typename F::T vv = F::sqr(v);
    GH_RFENCE();
    typename F::T r = F::mul(vv, x1z2);                    // x1z2 dead
    GH_RFENCE();
    typename F::T vvv = F::mul(v, vv);                     // vv dead
    GH_RFENCE();
    typename F::T uu = F::sqr(u);
    GH_RFENCE();
    typename F::T a = F::sub(F::sub(F::mul(uu, z1z2), vvv), F::dbl(r));   // uu dead
    GH_RFENCE();
    Proj<C> o;
    o.x = F::mul(v, a);                                    // v dead
    GH_RFENCE();
    o.z = F::mul(vvv, z1z2);                               // z1z2 dead
    GH_RFENCE();
    // Y3 = (r - a) u - vvv y1z2 as ONE dual product on a single accumulator chain (fp_mul2s, as in the XYZZ accumulation: one
    // Montgomery reduction of fourteen saved; the kernel's scratch frame is unchanged by it: 736 -> 640 B per lane in the
    // 512-register build, 1760 -> 1792 B in the 256-register one).  The towers keep two products.
    o.y = F::mul_sub_mul1(F::sub(r, a), u, vvv, y1z2);
#undef GH_RFENCE
    return o;
}

// The three accumulators of a program (run, wacc, and tmp of the salt detour) live in a per-program slab of global memory,
// word-major (slab[(slot * NW + word) * 64 + lane]: one 256-byte row per wave instruction), and only the operand of the
// current step is in registers: with all three held in registers next to the operands of the addition the compiler
// spilled 2.4 KB per lane -- 848 scratch instructions per step against the 156-234 explicit ones now.
template <class C> struct ReduceSlab {
    static constexpr int NW = (int)(sizeof(Proj<C>) / 4);
    static constexpr size_t WORDS = (size_t)3 * NW * 64;      // per program
    static __device__ __forceinline__ Proj<C> ld(const uint32_t* slab, int slot, int lane) {
        Proj<C> v;
        uint32_t* w = reinterpret_cast<uint32_t*>(&v);
        const uint32_t* p = slab + (size_t)slot * NW * 64 + lane;
#pragma unroll
        for (int k = 0; k < NW; k++) w[k] = p[(size_t)k * 64];
        return v;
    }
    static __device__ __forceinline__ void st(uint32_t* slab, int slot, int lane, const Proj<C>& v) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
        uint32_t* p = slab + (size_t)slot * NW * 64 + lane;
#pragma unroll
        for (int k = 0; k < NW; k++) p[(size_t)k * 64] = w[k];
    }
};
template <class C, int WAVES = 2>
__global__ void __launch_bounds__(256, WAVES)
msm_wave_reduce_kernel(WaveReduceIn<C> in0, WaveReduceIn<C> in1, WaveReduceIn<C> in2, uint32_t blocks_per_input,
                       uint32_t n_inputs, uint32_t segs_per_window, int L, const Aff<C>* __restrict__ salts,
                       Proj<C>* __restrict__ out, uint32_t* __restrict__ slabs) {
    typedef typename C::F F;
    typedef ReduceSlab<C> SL;
    enum { RUN = 0, WACC = 1, TMP = 2 };
    extern __shared__ uint32_t lds_raw[];
    const int lane = threadIdx.x & 63;
    Proj<C>* sh = reinterpret_cast<Proj<C>*>(lds_raw) + 64 * (threadIdx.x >> 6);
    const uint32_t gb = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // one program per wave
    if (gb >= n_inputs * blocks_per_input) return;
    const uint32_t which = gb / blocks_per_input, blk = gb % blocks_per_input;
    const WaveReduceIn<C> in = which == 0 ? in0 : (which == 1 ? in1 : in2);
    const uint32_t w = blk / segs_per_window, seg = blk % segs_per_window;
    const uint32_t item0 = seg * 64u * (uint32_t)L;
    if ((size_t)w * in.count + item0 >= (size_t)in.valid) {   // segment of padding slots only: all sums are infinity
        if (in.mode == 2) {
            const Proj<C> z = proj_zero<C>();
            st_proj<C>(out + ((size_t)blk * 64 + lane) * 2, z); st_proj<C>(out + ((size_t)blk * 64 + lane) * 2 + 1, z);
        } else if (lane == 0) {
            Proj<C>* oz = out + ((size_t)which * blocks_per_input + blk) * 3;
            const Proj<C> z = proj_zero<C>();
            st_proj<C>(oz, z); st_proj<C>(oz + 1, z); st_proj<C>(oz + 2, z);
        }
        return;
    }
    const int NS1 = in.mode == 1 ? L : 2 * L - 1;
    const int NST = in.mode == 1 ? L + 6 : (in.mode == 2 ? NS1 : NS1 + 18);
    uint32_t* slab = slabs + (size_t)gb * SL::WORDS;
    {
        const Proj<C> z = proj_zero<C>();
        SL::st(slab, RUN, lane, z); SL::st(slab, WACC, lane, z); SL::st(slab, TMP, lane, z);
    }
    int step = 0, det = 0, salt_id = 0;
    bool mydet = false, mid_done = false;
    Proj<C>* o = out + ((size_t)which * blocks_per_input + blk) * 3;
    while (step < NST) {
        int kind, off = 0, i = 0;
        if (step < NS1) {
            if (in.mode == 1) { kind = 0; i = L - 1 - step; }
            else { kind = (step & 1) ? 1 : 0; i = L - 1 - (step >> 1); }
        } else if (in.mode == 1) { kind = 4; off = 32 >> (step - NS1); }
        else if (step < NS1 + 6) { kind = 2; off = 32 >> (step - NS1); }
        else if (step < NS1 + 12) { kind = 3; off = 1 << (step - NS1 - 6); }
        else {
            kind = 4; off = 32 >> (step - NS1 - 12);
            if (!mid_done) {   // between scan and the last tree: publish runW = S_0, drop lane 0 from the tree
                if (lane == 0) { st_proj<C>(o, SL::ld(slab, RUN, lane)); SL::st(slab, RUN, lane, proj_zero<C>()); }
                mid_done = true;
            }
        }
        const bool exch = kind >= 2;
        if (exch && det == 0) st_proj<C>(sh + lane, SL::ld(slab, kind == 2 ? WACC : RUN, lane));
        if (exch) GH_WAVE_SYNC();
        const bool to_wacc = kind == 1 || kind == 2;
        const int dst = to_wacc ? WACC : RUN;
        bool active;
        if (kind == 0) active = item0 + (uint32_t)lane + 64u * (uint32_t)i < in.count;
        else if (kind == 1) active = true;
        else active = kind == 3 ? lane + off < 64 : lane < off;
        if (det > 0) active = mydet;
        // the step's second operand (re-loadable: it is read again below for the lanes whose sum is one of the operands)
        auto load_q = [&]() -> Proj<C> {
            Proj<C> q = proj_zero<C>();
            if (det == 1 || det == 3) {
                const Aff<C> sp = ld_aff<C>(salts + salt_id);
                q.x = sp.x; q.y = det == 3 ? F::neg(sp.y) : sp.y; q.z = F::one();
            } else if (kind == 0) {
                const uint32_t k = item0 + (uint32_t)lane + 64u * (uint32_t)i;
                if (k < in.count) q = ld_proj<C>(in.base + ((size_t)w * in.count + k) * in.stride + in.offset);
            } else if (kind == 1) {
                q = SL::ld(slab, RUN, lane);
            } else {
                const int partner = lane + off;
                if (kind == 3 ? partner < 64 : lane < off) q = ld_proj<C>(sh + partner);
            }
            return q;
        };
        const int src = det >= 2 ? TMP : dst;
        bool same, pz, qz;
        Proj<C> r;
        {
            const Proj<C> q = load_q();
            const Proj<C> p = SL::ld(slab, src, lane);
            r = proj_add_raw<C>(p, q, same, pz, qz);
        }
        if (__any((pz || qz) && active)) {   // p + infinity = p, infinity + q = q: patch those lanes from the operands, read again
            const Proj<C> q = load_q();
            const Proj<C> p = SL::ld(slab, src, lane);
            uint32_t* rw = reinterpret_cast<uint32_t*>(&r);
            const uint32_t* pw = reinterpret_cast<const uint32_t*>(&p);
            const uint32_t* qw = reinterpret_cast<const uint32_t*>(&q);
#pragma unroll
            for (int k = 0; k < SL::NW; k++) rw[k] = pz ? qw[k] : (qz ? pw[k] : rw[k]);
        }
        if (exch) GH_WAVE_SYNC();
        same = same && active;
        if (det == 0) {
            const bool any_same = __any(same) != 0;
            if (active && !same) SL::st(slab, dst, lane, r);
            if (any_same) {
                mydet = same;
                if (same) {   // salt with x != p.x / p.z  (rare path: out-of-line product)
                    const Proj<C> p = SL::ld(slab, src, lane);
                    Aff<C> s0 = ld_aff<C>(salts);
                    salt_id = C::FC::eq(C::FC::mul(s0.x, p.z), p.x) ? 1 : 0;
                }
                det = 1;
            } else {
                step++;
            }
        } else {
            if (mydet) SL::st(slab, det < 3 ? TMP : dst, lane, r);
            if (det == 3) { det = 0; mydet = false; step++; } else det++;
        }
    }
    if (in.mode == 2) {
        st_proj<C>(out + ((size_t)blk * 64 + lane) * 2, SL::ld(slab, RUN, lane));
        st_proj<C>(out + ((size_t)blk * 64 + lane) * 2 + 1, SL::ld(slab, WACC, lane));
    } else if (lane == 0) {
        if (in.mode == 1) {
            st_proj<C>(o, SL::ld(slab, RUN, lane));
        } else {
            st_proj<C>(o + 1, SL::ld(slab, WACC, lane));
            st_proj<C>(o + 2, SL::ld(slab, RUN, lane));
        }
    }
}

// ---------------------------------------------------------------- 5b. bucket reduction for G2: lane groups
// The same wave program over the split field policies of section 4b: a point of the program lives in a
// lane pair (Fq2) or triple (Fq3), one coefficient per lane, so a wave carries TPW = 32 / 16 items (Fq3:
// lanes 48..63 idle) and every addition runs on inlined Fp products in registers -- the tower version of
// msm_wave_reduce_kernel has to call out-of-line products whose operands travel through scratch.
// Item index = g + TPW * i (g = group, i = slot of the group), so  sum index * x = TPW * A + Bv.
struct P3 { Fp x, y, z; };
template <class FS> __device__ __forceinline__ P3 p3_zero() { return P3{fp_zero(), FS::one(), fp_zero()}; }
template <class FS> __device__ __forceinline__ P3 p3_add_sel(const P3& p, const P3& q, bool& same) {
    const bool pz = FS::is_zero(p.z), qz = FS::is_zero(q.z);
    Fp y1z2 = FS::mul(p.y, q.z);
    Fp x1z2 = FS::mul(p.x, q.z);
    Fp z1z2 = FS::mul(p.z, q.z);
    Fp u = FS::sub(FS::mul(p.z, q.y), y1z2);
    Fp v = FS::sub(FS::mul(p.z, q.x), x1z2);
    same = !pz && !qz && FS::is_zero(u) && FS::is_zero(v);
    Fp uu = FS::sqr(u);
    Fp vv = FS::sqr(v);
    Fp vvv = FS::mul(v, vv);
    Fp r = FS::mul(vv, x1z2);
    Fp a = FS::sub(FS::sub(FS::mul(uu, z1z2), vvv), FS::dbl(r));
    P3 o;
    o.x = FS::mul(v, a);
    o.y = FS::sub(FS::mul(FS::sub(r, a), u), FS::mul(vvv, y1z2));
    o.z = FS::mul(vvv, z1z2);
#pragma unroll
    for (int k = 0; k < NL; k++) {
        o.x.l[k] = pz ? q.x.l[k] : (qz ? p.x.l[k] : o.x.l[k]);
        o.y.l[k] = pz ? q.y.l[k] : (qz ? p.y.l[k] : o.y.l[k]);
        o.z.l[k] = pz ? q.z.l[k] : (qz ? p.z.l[k] : o.z.l[k]);
    }
    return o;
}

// without the final operand selects (see proj_add_raw): the caller patches lanes with an infinite operand
template <class FS> __device__ __forceinline__ P3 p3_add_raw(const P3& p, const P3& q, bool& same, bool& pz, bool& qz) {
    pz = FS::is_zero(p.z); qz = FS::is_zero(q.z);
    Fp y1z2 = FS::mul(p.y, q.z);
    Fp u = FS::sub(FS::mul(p.z, q.y), y1z2);
    Fp x1z2 = FS::mul(p.x, q.z);
    Fp v = FS::sub(FS::mul(p.z, q.x), x1z2);
    Fp z1z2 = FS::mul(p.z, q.z);
    same = !pz && !qz && FS::is_zero(u) && FS::is_zero(v);
    Fp vv = FS::sqr(v);
    Fp r = FS::mul(vv, x1z2);
    Fp vvv = FS::mul(v, vv);
    Fp uu = FS::sqr(u);
    Fp a = FS::sub(FS::sub(FS::mul(uu, z1z2), vvv), FS::dbl(r));
    P3 o;
    o.x = FS::mul(v, a);
    o.z = FS::mul(vvv, z1z2);
    o.y = FS::sub(FS::mul(FS::sub(r, a), u), FS::mul(vvv, y1z2));
    return o;
}
struct P3Slab {   // run / wacc / tmp of a program, word-major per lane (see ReduceSlab)
    static constexpr int NW = 3 * NL;
    static constexpr size_t WORDS = (size_t)3 * NW * 64;
    static __device__ __forceinline__ P3 ld(const uint32_t* slab, int slot, int lane) {
        P3 v;
        uint32_t* w = reinterpret_cast<uint32_t*>(&v);
        const uint32_t* p = slab + (size_t)slot * NW * 64 + lane;
#pragma unroll
        for (int k = 0; k < NW; k++) w[k] = p[(size_t)k * 64];
        return v;
    }
    static __device__ __forceinline__ void st(uint32_t* slab, int slot, int lane, const P3& v) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
        uint32_t* p = slab + (size_t)slot * NW * 64 + lane;
#pragma unroll
        for (int k = 0; k < NW; k++) p[(size_t)k * 64] = w[k];
    }
};

// (A 256-register build of this kernel -- two waves per SIMD, so that inside a batch a program would share its SIMD with a wave of
// the next MSM's round kernels -- was measured at the end of round 4: 2.4 KB of scratch per lane on Fq3, batches 2-3 % SLOWER.)
template <class C, class FS, int LANES, int TPW>
__global__ void __launch_bounds__(64, 1)
msm_wave_reduce_split_kernel(WaveReduceIn<C> in0, WaveReduceIn<C> in1, WaveReduceIn<C> in2, uint32_t blocks_per_input,
                             uint32_t n_inputs, uint32_t segs_per_window, int L, const Aff<C>* __restrict__ salts,
                             Proj<C>* __restrict__ out, uint32_t* __restrict__ slabs) {
    typedef P3Slab SL;
    enum { RUN = 0, WACC = 1, TMP = 2 };
    constexpr int LT = TPW == 32 ? 5 : (TPW == 16 ? 4 : 6);
    static_assert((1 << LT) == TPW && TPW * LANES <= 64, "groups per wave");
    extern __shared__ uint32_t lds_raw[];
    P3* sh = reinterpret_cast<P3*>(lds_raw);
    const int lane = threadIdx.x & 63;
    const bool live = lane < TPW * LANES;
    const int g = live ? lane / LANES : TPW, comp = lane % LANES;
    const uint32_t gb = blockIdx.x;
    if (gb >= n_inputs * blocks_per_input) return;
    const uint32_t which = gb / blocks_per_input, blk = gb % blocks_per_input;
    const WaveReduceIn<C> in = which == 0 ? in0 : (which == 1 ? in1 : in2);
    const uint32_t w = blk / segs_per_window, seg = blk % segs_per_window;
    const uint32_t item0 = seg * (uint32_t)TPW * (uint32_t)L;
    // coefficient `comp` of coordinate e of a projective / affine point in memory
    auto ld_c = [&](const void* pt, int e) { return ld_fp(reinterpret_cast<const Fp*>(pt) + LANES * e + comp); };
    auto st_c = [&](void* pt, int e, const Fp& v) { st_fp(reinterpret_cast<Fp*>(pt) + LANES * e + comp, v); };
    auto st_p3 = [&](Proj<C>* pt, const P3& v) { st_c(pt, 0, v.x); st_c(pt, 1, v.y); st_c(pt, 2, v.z); };
    Proj<C>* o = out + ((size_t)which * blocks_per_input + blk) * 3;
    if ((size_t)w * in.count + item0 >= (size_t)in.valid) {   // segment of padding slots only
        if (g == 0) { const P3 z = p3_zero<FS>(); st_p3(o, z); st_p3(o + 1, z); st_p3(o + 2, z); }
        return;
    }
    const int NS1 = in.mode == 1 ? L : 2 * L - 1;
    const int NST = in.mode == 1 ? L + LT : NS1 + 3 * LT;
    uint32_t* slab = slabs + (size_t)gb * SL::WORDS;
    {
        const P3 z = p3_zero<FS>();
        SL::st(slab, RUN, lane, z); SL::st(slab, WACC, lane, z); SL::st(slab, TMP, lane, z);
    }
    int step = 0, det = 0, salt_id = 0;
    bool mydet = false, mid_done = false;
    auto sh_store = [&](const P3& v) {
        uint2* d = reinterpret_cast<uint2*>(sh + lane);
        const uint2* sv = reinterpret_cast<const uint2*>(&v);
#pragma unroll
        for (int k = 0; k < (int)(sizeof(P3) / 8); k++) d[k] = sv[k];
    };
    auto sh_load = [&](int src_lane) {
        P3 v;
        const uint2* sv = reinterpret_cast<const uint2*>(sh + src_lane);
        uint2* d = reinterpret_cast<uint2*>(&v);
#pragma unroll
        for (int k = 0; k < (int)(sizeof(P3) / 8); k++) d[k] = sv[k];
        return v;
    };
    while (step < NST) {
        int kind, off = 0, i = 0;
        if (step < NS1) {
            if (in.mode == 1) { kind = 0; i = L - 1 - step; }
            else { kind = (step & 1) ? 1 : 0; i = L - 1 - (step >> 1); }
        } else if (in.mode == 1) { kind = 4; off = (TPW / 2) >> (step - NS1); }
        else if (step < NS1 + LT) { kind = 2; off = (TPW / 2) >> (step - NS1); }
        else if (step < NS1 + 2 * LT) { kind = 3; off = 1 << (step - NS1 - LT); }
        else {
            kind = 4; off = (TPW / 2) >> (step - NS1 - 2 * LT);
            if (!mid_done) {   // between scan and the last tree: publish runW = S_0, drop group 0 from the tree
                if (g == 0) { st_p3(o, SL::ld(slab, RUN, lane)); SL::st(slab, RUN, lane, p3_zero<FS>()); }
                mid_done = true;
            }
        }
        const bool exch = kind >= 2;
        if (exch && det == 0) sh_store(SL::ld(slab, kind == 2 ? WACC : RUN, lane));
        if (exch) GH_WAVE_SYNC();
        const bool to_wacc = kind == 1 || kind == 2;
        const int dst = to_wacc ? WACC : RUN;
        bool active;
        if (kind == 0) active = live && item0 + (uint32_t)g + (uint32_t)TPW * (uint32_t)i < in.count;
        else if (kind == 1) active = live;
        else active = live && (kind == 3 ? g + off < TPW : g < off);
        if (det > 0) active = mydet;
        auto load_q = [&]() -> P3 {   // the step's second operand; read again below for the lanes whose sum is one of the operands
            P3 q = p3_zero<FS>();
            if (det == 1 || det == 3) {
                q.x = ld_c(salts + salt_id, 0);
                q.y = ld_c(salts + salt_id, 1);
                if (det == 3) q.y = FS::neg(q.y);
                q.z = FS::one();
            } else if (kind == 0) {
                const uint32_t k = item0 + (uint32_t)g + (uint32_t)TPW * (uint32_t)i;
                if (live && k < in.count) {
                    const Proj<C>* pt = in.base + ((size_t)w * in.count + k) * in.stride + in.offset;
                    q.x = ld_c(pt, 0); q.y = ld_c(pt, 1); q.z = ld_c(pt, 2);
                }
            } else if (kind == 1) {
                q = SL::ld(slab, RUN, lane);
            } else {
                if (live && (kind == 3 ? g + off < TPW : g < off)) q = sh_load(lane + off * LANES);
            }
            return q;
        };
        const int src = det >= 2 ? TMP : dst;
        bool same, pz, qz;
        P3 r;
        {
            const P3 q = load_q();
            const P3 p = SL::ld(slab, src, lane);
            r = p3_add_raw<FS>(p, q, same, pz, qz);
        }
        if (__any((pz || qz) && active)) {   // p + infinity = p, infinity + q = q
            const P3 q = load_q();
            const P3 p = SL::ld(slab, src, lane);
            uint32_t* rw = reinterpret_cast<uint32_t*>(&r);
            const uint32_t* pw = reinterpret_cast<const uint32_t*>(&p);
            const uint32_t* qw = reinterpret_cast<const uint32_t*>(&q);
#pragma unroll
            for (int k = 0; k < SL::NW; k++) rw[k] = pz ? qw[k] : (qz ? pw[k] : rw[k]);
        }
        if (exch) GH_WAVE_SYNC();
        same = same && active;
        if (det == 0) {
            const bool any_same = __any(same) != 0;
            if (active && !same) SL::st(slab, dst, lane, r);
            if (any_same) {
                mydet = same;
                // salt with x != p.x / p.z (all lanes run the product: the group shuffles need their partners)
                const P3 p = SL::ld(slab, src, lane);
                const bool s0_hits = FS::eq(FS::mul(ld_c(salts, 0), p.z), p.x);
                if (same) salt_id = s0_hits ? 1 : 0;
                det = 1;
            } else {
                step++;
            }
        } else {
            if (mydet) SL::st(slab, det < 3 ? TMP : dst, lane, r);
            if (det == 3) { det = 0; mydet = false; step++; } else det++;
        }
    }
    if (g == 0) {
        if (in.mode == 1) {
            st_p3(o, SL::ld(slab, RUN, lane));
        } else {
            st_p3(o + 1, SL::ld(slab, WACC, lane));
            st_p3(o + 2, SL::ld(slab, RUN, lane));
        }
    }
}


}  // namespace gh

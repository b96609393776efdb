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
//      msm_heavy_*_kernel    buckets longer than the heavy threshold (4x the mean) are cut into
//                            chunks summed by one wave each, then combined (skewed real-world
//                            witnesses -- many equal small scalars -- and the top window).
//   5. msm_reduce1/2_kernel  sum_b b * B_b per window without the reference's per-window inversion:
//                            each lane serially folds L consecutive buckets (running sum), then the
//                            64 lanes of the wave combine their (run, weighted) pairs with a
//                            log-step suffix scan + tree reduction through LDS ("wavefront-wide
//                            bucket reduction"); a second launch combines the waves of a window.
//   6. window fold           753 dependent doublings: latency-bound, done on the host
//                            (ginger_hip.hip: fold_windows_host) from the W window sums.
#pragma once
#include <hip/hip_runtime.h>
#include "ec29.h"

namespace gh {

constexpr int MSM_REDUCE_L = 8;          // buckets folded serially per lane in reduce level 1
constexpr int MSM_MAX_HEAVY_THRESHOLD = 1024;  // upper bound of the run-time heavy threshold
constexpr int MSM_SIZE_BINS = MSM_MAX_HEAVY_THRESHOLD + 2;
constexpr int MSM_HEAVY_CHUNK = 256;           // entries of a heavy bucket summed by one wave

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

// ---------------------------------------------------------------- bases: ABI -> internal layout
// in: n x (2 * DEG * 24) words (x || y, Montgomery 2^768); out: n x Aff<C> (internal 2^754)
template <class C>
__global__ void __launch_bounds__(256) msm_convert_bases_kernel(const uint32_t* in, Aff<C>* out, size_t n) {
    typedef typename C::F F;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* p = in + i * (size_t)(48 * F::DEG);
    Aff<C> a;
    a.x = F::from_abi(p);
    a.y = F::from_abi(p + 24 * F::DEG);
    uint2* q = reinterpret_cast<uint2*>(out + i);
    const uint2* s = reinterpret_cast<const uint2*>(&a);
#pragma unroll
    for (int w = 0; w < (int)(sizeof(Aff<C>) / 8); w++) q[w] = s[w];
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
// scalars: n x 24 words canonical.  digits[w * n + i] = signed digit (0 = no contribution).
// counts[w * nb + |d|] += 1, nb = 2^(c-1) + 1 (slot 0 unused).
// Signed recoding: v = bits(s, wc, c) + carry; if v > 2^(c-1): d = v - 2^c, carry = 1.
// num_windows = floor(753 / c) + 1 guarantees the top window never carries out.
// (reference digit rule, unsigned: variable_base.rs:43-50.)
static __global__ void __launch_bounds__(256)
msm_digits_kernel(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ infinity, size_t n, int c,
                  int num_windows, uint32_t nb, int32_t* __restrict__ digits, uint32_t* __restrict__ counts) {
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
    const bool skip = !valid || (infinity != nullptr && infinity[i] != 0);
    const uint32_t half = 1u << (c - 1), full_mask = (c == 32) ? 0xFFFFFFFFu : ((1u << c) - 1);
    uint32_t carry = 0;
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
        if (v > half) { d = (int32_t)v - (int32_t)(1u << c); carry = 1; } else { d = (int32_t)v; carry = 0; }
        if (skip) d = 0;
        if (valid) digits[(size_t)w * n + i] = d;
        const uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
        wave_agg_inc(counts + (size_t)w * nb, mag, d != 0, 4);
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
msm_size_hist_kernel(const uint32_t* counts, size_t total, uint32_t heavy_thr, uint32_t* size_hist) {
    __shared__ uint32_t h[MSM_SIZE_BINS];
    for (int i = threadIdx.x; i < MSM_SIZE_BINS; i += 256) h[i] = 0;
    __syncthreads();
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < total) atomicAdd(&h[msm_size_bin(counts[g], heavy_thr)], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < MSM_SIZE_BINS; i += 256) if (h[i]) atomicAdd(&size_hist[i], h[i]);
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

// plan[0] = n_heavy, plan[1] = total number of chunks; chunk_start[h] for h in [0, n_heavy]
static __global__ void msm_heavy_plan_kernel(const uint32_t* size_hist, const uint32_t* counts, const uint32_t* order,
                                             uint32_t* chunk_start, uint32_t* plan) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t n_heavy = size_hist[0];
    uint32_t run = 0;
    for (uint32_t h = 0; h < n_heavy; h++) {
        chunk_start[h] = run;
        run += (counts[order[h]] + MSM_HEAVY_CHUNK - 1) / MSM_HEAVY_CHUNK;
    }
    chunk_start[n_heavy] = run;
    plan[0] = n_heavy;
    plan[1] = run;
}

// ---------------------------------------------------------------- 3. scatter
static __global__ void __launch_bounds__(256)
msm_scatter_kernel(const int32_t* __restrict__ digits, size_t n, int num_windows, uint32_t nb,
                   uint32_t* __restrict__ cursor /* = copy of starts */, uint32_t* __restrict__ sorted) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int w = blockIdx.y;
    const int32_t d = i < n ? digits[(size_t)w * n + i] : 0;
    const uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
    const uint32_t pos = wave_agg_inc(cursor + (size_t)w * nb, mag, d != 0, 4);
    if (d != 0) sorted[pos] = (uint32_t)i | (d < 0 ? 0x80000000u : 0u);
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
template <class C, int WAVES>
__global__ void __launch_bounds__(256, WAVES)
msm_accumulate_kernel(const Aff<C>* __restrict__ bases, const uint32_t* __restrict__ sorted,
                      const uint32_t* __restrict__ starts, const uint32_t* __restrict__ counts,
                      const uint32_t* __restrict__ order, uint32_t first, uint32_t total,
                      const Aff<C>* __restrict__ salts, Proj<C>* __restrict__ buckets) {
    typedef typename C::F F;
    uint32_t t = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint32_t g = order[t];
    const uint32_t beg = starts[g], cnt = counts[g];
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
            const uint32_t e = sorted[beg + k];
            q = ld_aff<C>(bases + (e & 0x7FFFFFFFu));
            if (e >> 31) q.y = F::neg(q.y);
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
            u = F::sub(u, acc.y);
            typename F::T uu = F::sqr(u);
            v = F::sub(v, acc.x);
            typename F::T vv = F::sqr(v);
            typename F::T vvv = F::mul(v, vv);
            typename F::T r = F::mul(vv, acc.x);
            typename F::T a = F::sub(F::sub(F::mul(uu, acc.z), vvv), F::dbl(r));
            acc.x = F::mul(v, a);
            acc.y = F::sub(F::mul(u, F::sub(r, a)), F::mul(vvv, acc.y));
            acc.z = F::mul(vvv, acc.z);
        }
        if (phase == 0 || phase == 3) { k++; phase = 0; } else phase++;
    }
    st_proj<C>(buckets + g, acc);
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

// Heavy buckets (skewed scalars; the top window, whose digits are only 0/1/2): every chunk of
// MSM_HEAVY_CHUNK entries is summed by one wave (lanes stride through the chunk, then a tree
// through LDS); a second launch adds the chunk sums of each heavy bucket the same way.
template <class C>
__global__ void __launch_bounds__(64)
msm_heavy_chunk_kernel(const Aff<C>* __restrict__ bases, const uint32_t* __restrict__ sorted,
                       const uint32_t* __restrict__ starts, const uint32_t* __restrict__ counts,
                       const uint32_t* __restrict__ order, const uint32_t* __restrict__ chunk_start, uint32_t n_heavy,
                       Proj<C>* __restrict__ partials) {
    typedef typename C::F F;
    extern __shared__ uint32_t lds_raw[];
    Proj<C>* sh = reinterpret_cast<Proj<C>*>(lds_raw);
    const int lane = threadIdx.x;
    // binary search: largest h with chunk_start[h] <= blockIdx.x
    uint32_t lo = 0, hi = n_heavy;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (chunk_start[mid] <= blockIdx.x) lo = mid; else hi = mid; }
    const uint32_t g = order[lo];
    const uint32_t j = blockIdx.x - chunk_start[lo];
    const uint32_t beg = starts[g] + j * MSM_HEAVY_CHUNK;
    uint32_t cnt = counts[g] - j * MSM_HEAVY_CHUNK;
    if (cnt > (uint32_t)MSM_HEAVY_CHUNK) cnt = MSM_HEAVY_CHUNK;
    Proj<C> acc = proj_zero<C>();
    for (uint32_t k = lane; k < cnt; k += 64) {
        const uint32_t e = sorted[beg + k];
        Aff<C> q = ld_aff<C>(bases + (e & 0x7FFFFFFFu));
        if (e >> 31) q.y = F::neg(q.y);
        acc = proj_madd_call<C>(acc, q);
    }
    acc = wave_tree_sum<C>(acc, sh, lane);
    if (lane == 0) st_proj<C>(partials + blockIdx.x, acc);
}
template <class C>
__global__ void __launch_bounds__(64)
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
// Level 1: block = one wave, covers 64 * L consecutive bucket slots of ONE window
// (slots per window nbp = padded to a multiple of 64 * L; slot index == bucket weight).
// Lane l folds slots [l L, l L + L): run = sum B, wacc = sum (i) B_(lL + i)  (local weights 0..L-1).
// Wave combine: S_l = suffix sum of run;  result_w = sum_l wacc_l + L * sum_{l>=1} S_l,
// result_run = S_0.  Output (run, wacc) per wave = per segment of 64 L slots with local weights.
template <class C>
__device__ __forceinline__ void wave_weighted_combine(Proj<C> run, Proj<C> wacc, int log_unit, Proj<C>* sh, int lane,
                                                      Proj<C>& out_run, Proj<C>& out_wacc) {
    // suffix scan of run (Hillis-Steele): S_l = sum_{j >= l} run_j
    Proj<C> S = run;
    for (int off = 1; off < 64; off <<= 1) {
        st_proj<C>(sh + lane, S);
        __syncthreads();
        if (lane + off < 64) S = proj_add_call<C>(S, ld_proj<C>(sh + lane + off));
        __syncthreads();
    }
    // V_l = wacc_l + unit * S_l (l >= 1), V_0 = wacc_0
    Proj<C> V = wacc;
    if (lane >= 1) {
        Proj<C> T = S;
        for (int d = 0; d < log_unit; d++) T = proj_dbl_call<C>(T);
        V = proj_add_call<C>(V, T);
    }
    out_run = S;  // valid in lane 0
    out_wacc = wave_tree_sum<C>(V, sh, lane);
}

template <class C>
__global__ void __launch_bounds__(64)
msm_reduce1_kernel(const Proj<C>* __restrict__ buckets, uint32_t nb /* valid slots per window */, uint32_t nbp /* padded */,
                   Proj<C>* __restrict__ seg_run, Proj<C>* __restrict__ seg_wacc) {
    extern __shared__ uint32_t lds_raw[];
    Proj<C>* sh = reinterpret_cast<Proj<C>*>(lds_raw);
    const int lane = threadIdx.x;
    const uint32_t segs_per_window = nbp / (64 * MSM_REDUCE_L);
    const uint32_t w = blockIdx.x / segs_per_window, seg = blockIdx.x % segs_per_window;
    const uint32_t slot0 = seg * 64 * MSM_REDUCE_L + lane * MSM_REDUCE_L;
    Proj<C> run = proj_zero<C>(), wacc = proj_zero<C>();
    // descending: run accumulates suffixes, wacc += run after each step gives local weights 0..L-1
    for (int i = MSM_REDUCE_L - 1; i >= 0; i--) {
        const uint32_t slot = slot0 + i;
        if (slot < nb && slot > 0) run = proj_add_call<C>(run, ld_proj<C>(buckets + (size_t)w * nb + slot));
        if (i > 0) wacc = proj_add_call<C>(wacc, run);
    }
    Proj<C> orun, owacc;
    int log_unit = 0;
    while ((1 << log_unit) < MSM_REDUCE_L) log_unit++;
    wave_weighted_combine<C>(run, wacc, log_unit, sh, lane, orun, owacc);
    if (lane == 0) {
        st_proj<C>(seg_run + blockIdx.x, orun);
        st_proj<C>(seg_wacc + blockIdx.x, owacc);
    }
}

// Level 2: one wave per window combines its segments (segs_per_window <= 64 * LSEG handled by a
// serial loop per lane).  Segment s has weight offset s * U, U = 64 * L = 2^log_u.
//   window_sum = sum_s wacc_s + U * sum_s s * run_s
template <class C>
__global__ void __launch_bounds__(64)
msm_reduce2_kernel(const Proj<C>* __restrict__ seg_run, const Proj<C>* __restrict__ seg_wacc,
                   uint32_t segs_per_window, int log_u, Proj<C>* __restrict__ window_sums) {
    extern __shared__ uint32_t lds_raw[];
    Proj<C>* sh = reinterpret_cast<Proj<C>*>(lds_raw);
    const int lane = threadIdx.x;
    const uint32_t w = blockIdx.x;
    const uint32_t per_lane = (segs_per_window + 63) / 64;  // consecutive segments per lane
    Proj<C> run = proj_zero<C>(), wacc_w = proj_zero<C>(), plain = proj_zero<C>();
    for (int i = (int)per_lane - 1; i >= 0; i--) {
        const uint32_t s = lane * per_lane + i;
        if (s < segs_per_window) {
            run = proj_add_call<C>(run, ld_proj<C>(seg_run + (size_t)w * segs_per_window + s));
            plain = proj_add_call<C>(plain, ld_proj<C>(seg_wacc + (size_t)w * segs_per_window + s));
        }
        if (i > 0) wacc_w = proj_add_call<C>(wacc_w, run);
    }
    // lane-local: weighted (in units of U) = wacc_w ; lane offset = lane * per_lane units
    // total units-weighted sum = sum_l wacc_w_l + per_lane * sum_{l>=1} S_l
    Proj<C> orun, ow;
    // per_lane need not be a power of two: multiply S_l by per_lane with a small double-and-add
    {
        Proj<C> S = run;
        for (int off = 1; off < 64; off <<= 1) {
            st_proj<C>(sh + lane, S);
            __syncthreads();
            if (lane + off < 64) S = proj_add_call<C>(S, ld_proj<C>(sh + lane + off));
            __syncthreads();
        }
        Proj<C> V = wacc_w;
        if (lane >= 1) {
            Proj<C> T = proj_zero<C>();
            for (int b = 31; b >= 0; b--) {
                T = proj_dbl_call<C>(T);
                if ((per_lane >> b) & 1) T = proj_add_call<C>(T, S);
            }
            V = proj_add_call<C>(V, T);
        }
        ow = wave_tree_sum<C>(V, sh, lane);
    }
    Proj<C> pl = wave_tree_sum<C>(plain, sh, lane);
    if (lane == 0) {
        for (int d = 0; d < log_u; d++) ow = proj_dbl_call<C>(ow);
        st_proj<C>(window_sums + w, proj_add_call<C>(pl, ow));
    }
}

}  // namespace gh

// aff_kernels.h -- bucket sums in AFFINE coordinates by pairwise rounds over a flat, bucket-ordered list.
//
// What it replaces: the inner loop of msm_inner (algebra/src/msm/variable_base.rs:36-59), n x W calls of
// add_assign_mixed (short_weierstrass_projective.rs:481-519, 11 Fp-mul each).  A bucket sum is a sum of affine
// points in ANY order, and an affine addition costs 1 inversion + 2 M + 1 S; Montgomery's trick (the one
// batch_normalization uses, swp.rs:402-442) shares one inversion over a batch at 3 M per member, so an addition
// comes to 5 M + 1 S + (one safegcd inversion, ~40 M) / batch -- against 11 M projectively.
//
// Round r turns the list P_r (all buckets back to back, bucket b at [st_r[b], st_r[b] + m_r[b])) into P_(r+1) with
// m_(r+1)[b] = ceil(m_r[b] / 2): output j of bucket b is P_r[st_r[b] + 2j] + P_r[st_r[b] + 2j + 1], an odd
// leftover is copied.  The unit of work is the OUTPUT ELEMENT, not the bucket:
//   * a descriptor desc_r[o] = (index of the first input | pair flag << 31) per output o (aff_desc_kernel);
//   * the round kernel gives every lane group B output elements of its wave's contiguous chunk, interleaved
//     (o = chunk + k * TPW + g): identical trip counts in all lanes, no bucket cursor, no divergence, and the 64
//     lanes of a wave read one contiguous span of the list per iteration;
//   * forward pass: running product of the denominators x2 - x1 (parked in `prefix`), ONE inversion per lane,
//     backward pass: 1 / (x2 - x1), lambda, the sum.
// Every case of the group law is handled in place (no fallback list):
//   x1 == x2, y1 == y2 != 0  -> doubling: lambda = (3 x1^2 + a) / (2 y1)      (the reference's P == Q branch, :492)
//   x1 == x2 otherwise       -> the point at infinity, stored as a marker (x.l[0] = 0xFFFFFFFF, not a limb value)
//   marker + Q -> Q
// Both rare paths sit behind a wave-uniform `any` so the common case pays nothing for them.
// After the last round a bucket holds a few points at most; msm_accumulate_kernel<.., AFFIN = true> adds them
// projectively (identity list, markers skipped) and leaves the buckets in the form the reduction expects.
//
// Memory layout ("T64"): the intermediate lists and the running products are stored wave-tiled and 16-byte-chunk
// major -- a tile holds the 64 lane slots of one wave iteration, chunk c of slot s at ((tile * NCH + c) * 64 + s) * 16
// bytes -- so that one wave instruction moves 1 KB of consecutive bytes.  Measured before (one lane = one 208-byte
// record, 8-byte accesses: every wave instruction touches 64 different cache lines): round 0 of 2^20 pairs issued at
// 58 % of the VALU limit with 32 % of the wave cycles waiting on the L1 / address path (profiles/r02_*); the rounds
// are HBM- and L1-bound work once the arithmetic is down to 6 products per addition.  Table rows (round 0) stay
// row-major -- they are random rows -- and are fetched with 16-byte loads.
#pragma once
#include "ec29.h"

namespace gh {

constexpr uint32_t AFF_MARK = 0xFFFFFFFFu;   // x.l[0] of the infinity marker
constexpr int AFF_MAX_ROUNDS = 26;

#ifndef GH_LD_ST_FP
#define GH_LD_ST_FP
GH_HD Fp ld_fp(const Fp* p) {
    Fp r;
    const uint2* q = reinterpret_cast<const uint2*>(p);
    GH_UNROLL for (int i = 0; i < NL / 2; i++) { uint2 v = q[i]; r.l[2 * i] = v.x; r.l[2 * i + 1] = v.y; }
    return r;
}
GH_HD void st_fp(Fp* p, const Fp& a) {
    uint2* q = reinterpret_cast<uint2*>(p);
    GH_UNROLL for (int i = 0; i < NL / 2; i++) q[i] = make_uint2(a.l[2 * i], a.l[2 * i + 1]);
}
#endif

// One field element (104 bytes, 8-byte aligned) of a row-major record with 16-byte loads.  hi8 = the element
// starts 8 bytes past a 16-byte boundary (the y of a G1 point: offset 104): one 8-byte load, then six of 16.
GH_HD Fp ld_fp_wide(const Fp* p, bool hi8) {
    Fp r;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
    if (hi8) {
        const uint2 h = *reinterpret_cast<const uint2*>(w);
        r.l[0] = h.x; r.l[1] = h.y;
        GH_UNROLL for (int i = 0; i < 6; i++) {
            const uint4 v = *reinterpret_cast<const uint4*>(w + 2 + 4 * i);
            r.l[2 + 4 * i] = v.x; r.l[3 + 4 * i] = v.y; r.l[4 + 4 * i] = v.z; r.l[5 + 4 * i] = v.w;
        }
    } else {
        GH_UNROLL for (int i = 0; i < 6; i++) {
            const uint4 v = *reinterpret_cast<const uint4*>(w + 4 * i);
            r.l[4 * i] = v.x; r.l[4 * i + 1] = v.y; r.l[4 * i + 2] = v.z; r.l[4 * i + 3] = v.w;
        }
        const uint2 h = *reinterpret_cast<const uint2*>(w + 24);
        r.l[24] = h.x; r.l[25] = h.y;
    }
    return r;
}

// ---- T64 lists.  A "slot" is one lane's share of an element: the (x, y) pair of one coefficient, 52 words =
// 13 chunks of 16 bytes (chunk 6 holds x.l[24..25] and y.l[0..1]); a prefix-product slot is one Fp padded to 7 chunks.
constexpr int T64_PT_CHUNKS = 13, T64_FP_CHUNKS = 7;
GH_HD size_t t64_bytes(size_t tiles, int chunks) { return tiles * (size_t)chunks * 1024; }
GH_HD const uint4* t64_at(const void* base, size_t tile, int nch, int c, uint32_t slot) {
    return reinterpret_cast<const uint4*>(base) + ((tile * (size_t)nch + (size_t)c) * 64 + slot);
}
GH_HD uint4* t64_at(void* base, size_t tile, int nch, int c, uint32_t slot) {
    return reinterpret_cast<uint4*>(base) + ((tile * (size_t)nch + (size_t)c) * 64 + slot);
}
GH_HD Fp t64_ld_x(const void* base, size_t tile, uint32_t slot) {
    Fp r;
    GH_UNROLL for (int c = 0; c < 6; c++) {
        const uint4 v = *t64_at(base, tile, T64_PT_CHUNKS, c, slot);
        r.l[4 * c] = v.x; r.l[4 * c + 1] = v.y; r.l[4 * c + 2] = v.z; r.l[4 * c + 3] = v.w;
    }
    const uint2 h = *reinterpret_cast<const uint2*>(t64_at(base, tile, T64_PT_CHUNKS, 6, slot));
    r.l[24] = h.x; r.l[25] = h.y;
    return r;
}
GH_HD Fp t64_ld_y(const void* base, size_t tile, uint32_t slot) {
    Fp r;
    const uint2 h = *(reinterpret_cast<const uint2*>(t64_at(base, tile, T64_PT_CHUNKS, 6, slot)) + 1);
    r.l[0] = h.x; r.l[1] = h.y;
    GH_UNROLL for (int c = 0; c < 6; c++) {
        const uint4 v = *t64_at(base, tile, T64_PT_CHUNKS, 7 + c, slot);
        r.l[2 + 4 * c] = v.x; r.l[3 + 4 * c] = v.y; r.l[4 + 4 * c] = v.z; r.l[5 + 4 * c] = v.w;
    }
    return r;
}
GH_HD void t64_st_xy(void* base, size_t tile, uint32_t slot, const Fp& x, const Fp& y) {
    GH_UNROLL for (int c = 0; c < 6; c++)
        *t64_at(base, tile, T64_PT_CHUNKS, c, slot) = make_uint4(x.l[4 * c], x.l[4 * c + 1], x.l[4 * c + 2], x.l[4 * c + 3]);
    *t64_at(base, tile, T64_PT_CHUNKS, 6, slot) = make_uint4(x.l[24], x.l[25], y.l[0], y.l[1]);
    GH_UNROLL for (int c = 0; c < 6; c++)
        *t64_at(base, tile, T64_PT_CHUNKS, 7 + c, slot) = make_uint4(y.l[2 + 4 * c], y.l[3 + 4 * c], y.l[4 + 4 * c], y.l[5 + 4 * c]);
}
GH_HD Fp t64_ld_fp(const void* base, size_t tile, uint32_t slot) {
    Fp r;
    GH_UNROLL for (int c = 0; c < 6; c++) {
        const uint4 v = *t64_at(base, tile, T64_FP_CHUNKS, c, slot);
        r.l[4 * c] = v.x; r.l[4 * c + 1] = v.y; r.l[4 * c + 2] = v.z; r.l[4 * c + 3] = v.w;
    }
    const uint2 h = *reinterpret_cast<const uint2*>(t64_at(base, tile, T64_FP_CHUNKS, 6, slot));
    r.l[24] = h.x; r.l[25] = h.y;
    return r;
}
GH_HD void t64_st_fp(void* base, size_t tile, uint32_t slot, const Fp& a) {
    GH_UNROLL for (int c = 0; c < 6; c++)
        *t64_at(base, tile, T64_FP_CHUNKS, c, slot) = make_uint4(a.l[4 * c], a.l[4 * c + 1], a.l[4 * c + 2], a.l[4 * c + 3]);
    *reinterpret_cast<uint2*>(t64_at(base, tile, T64_FP_CHUNKS, 6, slot)) = make_uint2(a.l[24], a.l[25]);
}

// ---- lane-group field policies: an element of the coordinate field lives in LANES adjacent lanes, one Fp
//      coefficient per lane (G1: one lane; Fq2: lane pairs; Fq3: lane triples -- F2S / F3S of msm_kernels.h).
template <class P> struct F1S {
    typedef Fp T;
    static constexpr int LANES = 1;
    static constexpr int WAVES = 2;
    GH_HD static T one() { return fp_one<P>(); }
    GH_HD static T zero() { return fp_zero(); }
    GH_HD static T add(const T& a, const T& b) { return fp_add<P>(a, b); }
    GH_HD static T sub(const T& a, const T& b) { return fp_sub<P>(a, b); }
    GH_HD static T sub_lazy(const T& a, const T& b) { return fp_sub_lazy<P>(a, b); }   // operand of ONE product (fp29.h)
    GH_HD static T dbl(const T& a) { return fp_dbl<P>(a); }
    GH_HD static T neg(const T& a) { return fp_neg<P>(a); }
    GH_HD static T mul(const T& a, const T& b) { return fp_mul<P>(a, b); }
    GH_HD static T sqr(const T& a) { return fp_sqr<P>(a); }
    GH_HD static bool is_zero(const T& a) { return fp_is_zero(a); }
    GH_HD static bool eq(const T& a, const T& b) { return fp_eq(a, b); }
    GH_HD static T inv(const T& a) { return fp_inv<P>(a); }
};

// wave-uniform "does any lane want the rare path" (host emulation: lanes run one at a time)
GH_HD bool aff_any(bool f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __any(f ? 1 : 0) != 0;
#else
    return f;
#endif
}

// the curve coefficient a as an element of the lane-group field (coefficient `comp` of it)
template <class C, class FS> struct CurveA;
template <class FS> struct CurveA<Mnt4G1, FS> { GH_HD static Fp get(int) { return fp_dbl<P4>(fp_one<P4>()); } };            // a = 2
template <class FS> struct CurveA<Mnt6G1, FS> { GH_HD static Fp get(int) { return fp_mul_small<P6, 11>(fp_one<P6>()); } };  // a = 11
template <class FS> struct CurveA<Mnt4G2, FS> {   // a' = (26, 0)          (curves/mnt4753/g2.rs:57-75)
    GH_HD static Fp get(int comp) { return comp == 0 ? fp_mul_small<P4, 26>(fp_one<P4>()) : fp_zero(); }
};
template <class FS> struct CurveA<Mnt6G2, FS> {   // a' = (0, 0, 11)       (curves/mnt6753/g2.rs:71-100)
    GH_HD static Fp get(int comp) { return comp == 2 ? fp_mul_small<P6, 11>(fp_one<P6>()) : fp_zero(); }
};

template <class C> struct AffRoundArgs {
    const Aff<C>* rows;        // round 0: the bases / the shift table, row-major (sorted != nullptr)
    const void* in;            // later rounds: the previous round's output, T64 (sorted == nullptr)
    const uint32_t* sorted;    // round 0: list entries (row index | sign << 31); nullptr afterwards
    const uint32_t* desc;      // per output element of this launch: first input index (global) | pair << 31
    uint32_t n_out;            // output elements of this launch (one chunk of buckets, one round)
    uint32_t in_base;          // global index of the chunk's first input element in the previous round's list (the T64 lists
                               // of a chunk start at 0)
    void* prefix;              // running products, T64 (one tile per wave iteration)
    void* out;                 // this round's output list, T64
    void *stage1, *stage2;     // round 0: the two inputs of every output element as gathered by the forward pass (signs
                               // applied), T64 in output order -- the backward pass reads them back instead of gathering again
    uint32_t groups;           // lane groups in the grid
    uint32_t bmin;             // minimum batch per lane group (one inversion each)
    const uint32_t* run_if;    // not null: the launch is the fallback of the assembly kernels (asmgen/g2_rounds.py) and does nothing
                               // unless their forward kernel raised this flag (an element on the group law's rare branches)
};

// One lane of a round.  t = global lane-group index, comp = this lane's coefficient, live = lane belongs to a group.
// R0: round 0 (inputs are table rows named by the sorted list, with signs) or a later round (inputs in the previous
// round's T64 list).  A compile-time switch: with both load shapes behind a run-time select hipcc merged them into
// 26 single-dword loads with per-dword selected addresses (round 0: 13.9 -> 32.7 ms).
template <class C, class FS, bool R0> struct AffRoundLane {
    static constexpr int LANES = FS::LANES;
    static constexpr uint32_t TPW = 64 / LANES;
    typedef Fp T;

    GH_HD static const Fp* coef(const Aff<C>* pt, int e, int comp) { return reinterpret_cast<const Fp*>(pt) + LANES * e + comp; }

    struct Elem {          // one output element's inputs as this lane sees them
        bool act, pair;
        uint32_t i1, i2;   // input records (i2 == i1 for a copy)
        bool n1, n2;       // negate y (round 0 sign bits)
    };
    // Coordinate E (0: x, 1: y) of input WHICH (0 / 1) of element e, whose output position is (tile, slot).
    //   round 0, forward pass : a table row named by the list entry (16-byte loads; one gather per row -- the pass stages
    //                           what it gathered, see run())
    //   round 0, backward pass: the staged copy at the element's own position
    //   later rounds          : the previous round's T64 list at the descriptor's index
    // Measured with round 0 gathering in both passes: 14.0 ms for 18.4 M additions at 2^20 pairs, the same with 8- and
    // 16-byte loads, 20 % less per element from a 218 MB array than from the 7.85 GB table -- bound by the number of
    // random rows (75 M at ~5.4 G rows/s, the rate MI355X_MICROARCH.md gives for gathered rows), not by bytes.
    template <int E, int WHICH, bool FWD> GH_HD static T ld_pt(const AffRoundArgs<C>& a, const Elem& e, size_t tile, uint32_t slot, int comp) {
        if constexpr (R0 && FWD) {
            return ld_fp_wide(coef(a.rows + (WHICH ? e.i2 : e.i1), E, comp), LANES == 1 ? (E & 1) != 0 : ((LANES * E + comp) & 1) != 0);
        } else if constexpr (R0) {
            const void* base = WHICH ? a.stage2 : a.stage1;
            if (!e.act) tile = 0;     // padding iterations of the last wave lie beyond the list: read a valid tile, use nothing
            return E == 0 ? t64_ld_x(base, tile, slot) : t64_ld_y(base, tile, slot);
        } else {
            const uint32_t idx = WHICH ? e.i2 : e.i1;
            const size_t it = idx / TPW;
            const uint32_t is = (idx % TPW) * LANES + (uint32_t)comp;
            return E == 0 ? t64_ld_x(a.in, it, is) : t64_ld_y(a.in, it, is);
        }
    }
    // One table row (x and y of this lane's coefficient), round 0.  Every lane reads its own row, so one wave-wide load
    // touches 64 different pages; the L1 translation cache holds fewer, and with the 14 loads of a row issued wave-wide
    // each of them missed all 64 translations again (measured: 4.9e8 UTCL1 misses for 18.4 M additions -- 13 per row --
    // against 1.6e3 in a later round; 3.7 wave cycles per VALU instruction against 2.0).  The row is therefore fetched a
    // part of the wave at a time: the pages of 64 / GH_AFF_GATHER_SPLIT lanes stay resident across the row's loads
    // (4.9e8 -> 2.6e7 misses, round 0 13.8 -> 12.0 ms).
#ifndef GH_AFF_GATHER_SPLIT
#define GH_AFF_GATHER_SPLIT 4
#endif
    GH_HD static void gather_row(const AffRoundArgs<C>& a, uint32_t idx, int comp, T& x, T& y) {
        const Fp* px = coef(a.rows + idx, 0, comp);
        const Fp* py = coef(a.rows + idx, 1, comp);
        const bool hx = LANES == 1 ? false : (comp & 1) != 0, hy = LANES == 1 ? true : ((LANES + comp) & 1) != 0;
#if defined(__HIP_DEVICE_COMPILE__)
        const int part = (int)((threadIdx.x & 63u) / (64 / GH_AFF_GATHER_SPLIT));
#pragma unroll
        for (int q = 0; q < GH_AFF_GATHER_SPLIT; q++) {
            if (part == q) { x = ld_fp_wide(px, hx); y = ld_fp_wide(py, hy); }
            __builtin_amdgcn_sched_barrier(0);
        }
#else
        x = ld_fp_wide(px, hx); y = ld_fp_wide(py, hy);
#endif
    }
    GH_HD static T signed_y(T y, bool negate) {   // sign of a round-0 list entry
        const T ny = FS::neg(y);
        GH_UNROLL for (int i = 0; i < NL; i++) y.l[i] = negate ? ny.l[i] : y.l[i];
        return y;
    }

    // kind of an element: 0 copy / pass-through (d = 1), 1 generic addition, 2 doubling, 3 cancellation
    // d: the denominator whose inverse the addition needs (1 where none is needed)
    // The equal-x cases need the y coordinates: yy() delivers them behind a wave-uniform branch.
    template <class YY> GH_HD static int classify(const Elem& e, const T& x1, const T& x2, T& d, YY yy) {
        d = FS::one();
        int kind = 0;
        const bool m1 = x1.l[0] == AFF_MARK, m2 = x2.l[0] == AFF_MARK;
        const bool both = e.pair && !m1 && !m2;
        const bool eqx = both && FS::eq(x1, x2);
        if (both && !eqx) { d = FS::sub_lazy(x2, x1); kind = 1; }   // feeds products only
        if (aff_any(eqx)) {
            T y1, y2;
            yy(y1, y2);
            const bool dbl = eqx && FS::eq(y1, y2) && !FS::is_zero(y1);
            const T y2x = FS::dbl(y1);
            if (dbl) { d = y2x; kind = 2; } else if (eqx) kind = 3;
        }
        return kind;
    }

    // The loops are software pipelines: an element's descriptor is fetched three iterations ahead, its list
    // entries (round 0) two ahead, its x coordinates one ahead, so that no load an iteration consumes was issued
    // less than one iteration (>= 2000 instructions) earlier.  Measured without it: round 0 of 2^20 pairs 13.0 ms
    // for 7 ms of instructions (desc -> list entry -> table row is a chain of three dependent loads).
    template <bool ENTRIES> struct Pipe {    // ENTRIES: the inputs are named by list entries (round 0, forward pass)
        const AffRoundArgs<C>& a;
        uint32_t chunk, g, B, n_out;
        bool live;
        int dir;                // +1 forward, -1 backward
        // stage registers.  The list entries are kept RAW (as loaded) and decoded only when the element is handed
        // out one iteration later, so that nothing touches a load's result in the iteration that issued it.
        uint32_t de2;           // descriptor of element j + 2 dir
        bool act2;
        uint32_t de1, raw1, raw2;   // element j + dir: descriptor and its two list entries (round 0) in flight
        bool act1;
        GH_HD Pipe(const AffRoundArgs<C>& a_, uint32_t chunk_, uint32_t g_, uint32_t B_, uint32_t n_out_, bool live_, int dir_)
            : a(a_), chunk(chunk_), g(g_), B(B_), n_out(n_out_), live(live_), dir(dir_), de2(0), act2(false), de1(0), raw1(0), raw2(0), act1(false) {}
        GH_HD bool active(int64_t j) const { return live && j >= 0 && j < (int64_t)B && (uint64_t)chunk + (uint64_t)j * TPW + g < n_out; }
        GH_HD uint32_t desc_at(int64_t j) const { return active(j) ? a.desc[chunk + (uint32_t)j * TPW + g] : 0u; }
        GH_HD void fetch_entries(uint32_t de, bool act, uint32_t& r1, uint32_t& r2) const {
            const uint32_t ai = de & 0x7FFFFFFFu;
            const bool pair = act && (de >> 31) != 0;
            if constexpr (ENTRIES) {
                r1 = act ? a.sorted[ai] : 0u;
                r2 = pair ? a.sorted[ai + 1] : 0u;
            } else {
                r1 = act ? ai - a.in_base : 0u; r2 = r1 + 1;
            }
        }
        GH_HD Elem decode(uint32_t de, bool act, uint32_t r1, uint32_t r2) const {
            Elem e;
            e.act = act;
            e.pair = act && (de >> 31) != 0;
            if (!e.pair) r2 = r1;
            constexpr bool sg = ENTRIES;
            e.i1 = sg ? (r1 & 0x7FFFFFFFu) : r1; e.i2 = sg ? (r2 & 0x7FFFFFFFu) : r2;
            e.n1 = sg && (r1 >> 31) != 0; e.n2 = sg && (r2 >> 31) != 0;
            return e;
        }
        // prologue: element j0 resolved (returned), the entries of j0 + dir and the descriptor of j0 + 2 dir in flight
        GH_HD Elem start(int64_t j0) {
            uint32_t r1, r2;
            const uint32_t de0 = desc_at(j0);
            fetch_entries(de0, active(j0), r1, r2);
            const Elem e0 = decode(de0, active(j0), r1, r2);
            de1 = desc_at(j0 + dir); act1 = active(j0 + dir);
            fetch_entries(de1, act1, raw1, raw2);
            de2 = desc_at(j0 + 2 * dir); act2 = active(j0 + 2 * dir);
            return e0;
        }
        // called in iteration j: hands out element j + dir (its entries were fetched one iteration ago), fetches the
        // entries of j + 2 dir and the descriptor of j + 3 dir
        GH_HD Elem advance(int64_t j) {
            const Elem r = decode(de1, act1, raw1, raw2);
            de1 = de2; act1 = act2;
            fetch_entries(de1, act1, raw1, raw2);
            de2 = desc_at(j + 3 * dir); act2 = active(j + 3 * dir);
            return r;
        }
    };

#if defined(__HIP_DEVICE_COMPILE__)
#define GH_AFF_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define GH_AFF_FENCE()
#endif

    GH_HD static void run(const AffRoundArgs<C>& a, uint32_t t, int comp, bool live) {
        const uint32_t n_out = a.n_out;
        uint32_t B = (n_out + a.groups - 1) / a.groups;
        if (B < a.bmin) B = a.bmin;
        const uint32_t wv = t / TPW, g = t % TPW;
        const uint64_t chunk64 = (uint64_t)wv * TPW * B;
        if (chunk64 >= n_out) return;                         // the whole wave: nothing left for it
        const uint32_t chunk = (uint32_t)chunk64;
        const size_t tile0 = (size_t)wv * B;                    // output o = chunk + k TPW + g lives in tile tile0 + k, slot g
        const uint32_t slot = g * LANES + (uint32_t)comp;
        // ---- forward: running product of the denominators
        T acc = FS::one();
        if constexpr (R0) {
            // round 0: every table row is gathered ONCE, here, and staged with its sign applied
            Pipe<true> pp(a, chunk, g, B, n_out, live, +1);
            Elem e = pp.start(0);
            T x1, y1, x2, y2;
            gather_row(a, e.i1, comp, x1, y1);
            gather_row(a, e.i2, comp, x2, y2);
            for (uint32_t k = 0; k < B; k++) {
                const Elem en = pp.advance((int64_t)k);
                GH_AFF_FENCE();
                const T y1s = signed_y(y1, e.n1), y2s = signed_y(y2, e.n2);
                if (e.act) {
                    t64_st_xy(a.stage1, tile0 + k, slot, x1, y1s);
                    t64_st_xy(a.stage2, tile0 + k, slot, x2, y2s);
                }
                T d;
                classify(e, x1, x2, d, [&](T& o1, T& o2) { o1 = y1s; o2 = y2s; });
                GH_AFF_FENCE();
                // the next element's rows: in flight during the product.  (Requesting its x a whole iteration ahead and
                // its y here -- two bursts per row -- measured slower: 13.9 against 12.0 ms.)
                T xn1, yn1, xn2, yn2;
                gather_row(a, en.i1, comp, xn1, yn1);
                gather_row(a, en.i2, comp, xn2, yn2);
                GH_AFF_FENCE();
                acc = FS::mul(acc, d);
                if (e.act) t64_st_fp(a.prefix, tile0 + k, slot, acc);
                GH_AFF_FENCE();
                e = en; x1 = xn1; x2 = xn2; y1 = yn1; y2 = yn2;
            }
        } else {
            Pipe<false> pp(a, chunk, g, B, n_out, live, +1);
            Elem e = pp.start(0);
            T x1 = ld_pt<0, 0, true>(a, e, 0, 0, comp), x2 = ld_pt<0, 1, true>(a, e, 0, 0, comp);
            for (uint32_t k = 0; k < B; k++) {
                const Elem en = pp.advance((int64_t)k);
                const T xn1 = ld_pt<0, 0, true>(a, en, 0, 0, comp), xn2 = ld_pt<0, 1, true>(a, en, 0, 0, comp);   // element k + 1
                GH_AFF_FENCE();
                T d;
                classify(e, x1, x2, d, [&](T& o1, T& o2) { o1 = ld_pt<1, 0, true>(a, e, 0, 0, comp); o2 = ld_pt<1, 1, true>(a, e, 0, 0, comp); });
                acc = FS::mul(acc, d);
                if (e.act) t64_st_fp(a.prefix, tile0 + k, slot, acc);
                GH_AFF_FENCE();
                e = en; x1 = xn1; x2 = xn2;
            }
        }
        // ---- one inversion for the lane group's whole batch
        T inv = FS::inv(acc);
        // ---- backward
        {
            Pipe<false> pp(a, chunk, g, B, n_out, live, -1);
            Elem e = pp.start((int64_t)B - 1);
            T x1 = ld_pt<0, 0, false>(a, e, tile0 + B - 1, slot, comp), x2 = ld_pt<0, 1, false>(a, e, tile0 + B - 1, slot, comp);
            for (uint32_t k = B; k-- > 0;) {
                // this element's late operands: consumed after the first product
                T pk = FS::one();
                if (e.act && k > 0) pk = t64_ld_fp(a.prefix, tile0 + k - 1, slot);
                const T y1 = ld_pt<1, 0, false>(a, e, tile0 + k, slot, comp), y2 = ld_pt<1, 1, false>(a, e, tile0 + k, slot, comp);
                GH_AFF_FENCE();
                T d;
                const int kind = classify(e, x1, x2, d, [&](T& o1, T& o2) { o1 = y1; o2 = y2; });
                const bool m1 = x1.l[0] == AFF_MARK, m2 = x2.l[0] == AFF_MARK;
                const T sx = FS::add(x1, x2);                    // x2 is dead from here on (register budget)
                GH_AFF_FENCE();
                const T inv0 = inv;
                inv = FS::mul(inv0, d);
                GH_AFF_FENCE();
                const T dinv = FS::mul(inv0, pk);                // 1 / d_k
                GH_AFF_FENCE();
                T num = FS::sub_lazy(y2, y1);                    // feeds one product
                if (aff_any(kind == 2)) {                        // 3 x1^2 + a
                    const T xx = FS::sqr(x1);
                    const T n2 = FS::add(FS::add(FS::dbl(xx), xx), CurveA<C, FS>::get(comp));
                    GH_UNROLL for (int i = 0; i < NL; i++) num.l[i] = kind == 2 ? n2.l[i] : num.l[i];
                }
                const T lam = FS::mul(num, dinv);
                GH_AFF_FENCE();
                // the next element's x coordinates: in flight during the last two products
                const Elem en = pp.advance((int64_t)k);
                const size_t tn = tile0 + (k > 0 ? k - 1 : 0);
                const T xn1 = ld_pt<0, 0, false>(a, en, tn, slot, comp), xn2 = ld_pt<0, 1, false>(a, en, tn, slot, comp);
                GH_AFF_FENCE();
                T x3 = FS::sub(FS::sqr(lam), sx);
                GH_AFF_FENCE();
                T y3 = FS::sub(FS::mul(lam, FS::sub_lazy(x1, x3)), y1);
                GH_AFF_FENCE();
                // select the result: copy / P + marker -> P / marker + Q -> Q (rare: Q is fetched again) / cancellation
                const bool take1 = !e.pair || m2, take2 = e.pair && m1 && !m2;
                GH_UNROLL for (int i = 0; i < NL; i++) {
                    x3.l[i] = take1 ? x1.l[i] : x3.l[i];
                    y3.l[i] = take1 ? y1.l[i] : y3.l[i];
                }
                if (aff_any(take2)) {
                    const T xq = ld_pt<0, 1, false>(a, e, tile0 + k, slot, comp), yq = ld_pt<1, 1, false>(a, e, tile0 + k, slot, comp);
                    GH_UNROLL for (int i = 0; i < NL; i++) {
                        x3.l[i] = take2 ? xq.l[i] : x3.l[i];
                        y3.l[i] = take2 ? yq.l[i] : y3.l[i];
                    }
                }
                if (kind == 3) {
                    GH_UNROLL for (int i = 0; i < NL; i++) { x3.l[i] = 0; y3.l[i] = 0; }
                    x3.l[0] = AFF_MARK;
                }
                if (e.act) t64_st_xy(a.out, tile0 + k, slot, x3, y3);
                e = en; x1 = xn1; x2 = xn2;
            }
        }
#undef GH_AFF_FENCE
    }
};

// ---- planning (plain index arithmetic; host-callable bodies for the CPU emulation test)
// cnt[(r - 1) * stride + b] = m_r[b] = ceil(counts[b] / 2^r), r = 1 .. R
GH_HD void aff_counts_body(const uint32_t* counts, uint32_t b, int R, size_t stride, uint32_t* cnt) {
    uint32_t m = counts[b];
    for (int r = 1; r <= R; r++) { m = (m + 1) >> 1; cnt[(size_t)(r - 1) * stride + b] = m; }
}
// descriptor of output o of a round: st_in / m_in describe the input list, st_out the output list
GH_HD uint32_t aff_desc_body(const uint32_t* st_in, const uint32_t* m_in, const uint32_t* st_out, uint32_t total, uint32_t o) {
    uint32_t lo = 0, hi = total;          // last b with st_out[b] <= o
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (st_out[mid] <= o) lo = mid; else hi = mid; }
    const uint32_t j = o - st_out[lo];
    const uint32_t ai = st_in[lo] + 2 * j;
    return ai | ((2 * j + 1 < m_in[lo]) ? 0x80000000u : 0u);
}

#if defined(__HIPCC__)
static __global__ void __launch_bounds__(256)
aff_counts_kernel(const uint32_t* __restrict__ counts, uint32_t total, int R, size_t stride, uint32_t* __restrict__ cnt) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < total) aff_counts_body(counts, b, R, stride, cnt);
}
// descriptors of the outputs [o_base, o_base + n_out) of a round (one chunk of buckets), stored chunk-relative
static __global__ void __launch_bounds__(256)
aff_desc_kernel(const uint32_t* __restrict__ st_in, const uint32_t* __restrict__ m_in, const uint32_t* __restrict__ st_out,
                uint32_t total, uint32_t o_base, uint32_t n_out, uint32_t* __restrict__ desc) {
    for (uint32_t o = blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += gridDim.x * blockDim.x)
        desc[o] = aff_desc_body(st_in, m_in, st_out, total, o_base + o);
}
// Chunks of buckets with about equal numbers of list entries (the scratch lists of the rounds are sized per chunk):
// bq[j] = first bucket of chunk j (bq[K] = total); tab[j * (R + 1) + r] = global index of chunk j's first element in
// round r's list (row K: the list lengths).
static __global__ void aff_chunks_kernel(const uint32_t* __restrict__ starts, const uint32_t* __restrict__ counts,
                                         const uint32_t* __restrict__ st, const uint32_t* __restrict__ cnt, uint32_t total, int R,
                                         size_t stride, uint32_t K, uint32_t* __restrict__ bq, uint32_t* __restrict__ tab) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > K) return;
    const uint32_t n0 = starts[total - 1] + counts[total - 1];
    uint32_t b = total;
    if (j < K) {
        const uint32_t target = (uint32_t)(((uint64_t)n0 * j) / K);
        uint32_t lo = 0, hi = total;     // first bucket with starts[b] >= target
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (starts[mid] < target) lo = mid + 1; else hi = mid; }
        b = lo;
    }
    bq[j] = b;
    for (int r = 0; r <= R; r++) {
        const uint32_t* s_r = r == 0 ? starts : st + (size_t)(r - 1) * stride;
        const uint32_t* m_r = r == 0 ? counts : cnt + (size_t)(r - 1) * stride;
        tab[(size_t)j * (R + 1) + r] = b < total ? s_r[b] : s_r[total - 1] + m_r[total - 1];
    }
}

// The inversion between the forward and the backward assembly kernel of a round (asmgen/g2_rounds.py): one wave per T64 tile
// of running products (tile = wave of the round kernels, slot = lane = lane group * LANES + coefficient), inverted in the
// tower in place -- one Fp inversion per lane group (F2S / F3S inv: the norm forms of fp2.rs / fp3.rs inverse).
template <class FS>
__global__ void __launch_bounds__(256) aff_inv_kernel(void* accs, uint32_t waves, uint32_t n_out, uint32_t B, const uint32_t* flag) {
    constexpr uint32_t TPW = 64 / FS::LANES;
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave >= waves || (uint64_t)wave * TPW * B >= n_out) return;     // that wave of the round kernels had nothing to do
    if (flag && *flag != 0) return;                                      // the round is redone by aff_round_kernel
    Fp v = t64_ld_fp(accs, wave, lane);
    GH_UNROLL for (int i = 0; i < NL; i++) v.l[i] &= LM;                 // the idle lane of a triple wave reads an unwritten slot
    const Fp r = FS::inv(v);
    if (lane < TPW * FS::LANES) t64_st_fp(accs, wave, lane, r);
}

// The exceptions of an assembly round (asmgen/g2_rounds.py): pairs that need the group law's rare branches -- P + P (the
// reference's doubling branch, swp.rs:492), P - P, an infinity marker among the inputs.  The assembly kernels treat such an
// element as a copy and append its index to the list behind the control block (ctl[0]: the list overflowed, the whole round is
// redone by aff_round_kernel; ctl[1]: entries; indices from ctl[16]); here every listed element is recomputed by one lane
// group, with an inversion of its own, and overwrites the copy.  kind: 0 pass-through, 1 generic (a false positive of the
// list's superset test), 2 doubling, 3 cancellation -- the cases of AffRoundLane::classify.
constexpr uint32_t AFF_FIX_CAP = 16384;
template <class C, class FS, bool R0>
__global__ void __launch_bounds__(256, FS::WAVES) aff_fix_kernel(AffRoundArgs<C> a, const uint32_t* ctl) {
    constexpr int LANES = FS::LANES;
    constexpr uint32_t TPW = 64 / LANES;
    typedef Fp T;
    if (ctl[0] != 0) return;
    uint32_t count = ctl[1];
    if (count > AFF_FIX_CAP) count = AFF_FIX_CAP;
    if (count == 0) return;
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const bool live = lane < TPW * LANES;
    const uint32_t g = live ? lane / LANES : TPW - 1;
    const int comp = (int)(lane % LANES);
    for (uint32_t base = wave * TPW; base < count; base += nwaves * TPW) {      // wave-uniform trip count: the lane-group ops shuffle
        uint32_t e = base + g;
        const bool act = live && e < count;
        if (e >= count) e = count - 1;
        const uint32_t o = ctl[16 + e];
        const uint32_t ai = a.desc[o] & 0x7FFFFFFFu;                            // only pairs are listed
        T x1, y1, x2, y2;
        if constexpr (R0) {
            const uint32_t r1 = a.sorted[ai], r2 = a.sorted[ai + 1];
            const Fp* p1 = reinterpret_cast<const Fp*>(a.rows + (r1 & 0x7FFFFFFFu));
            const Fp* p2 = reinterpret_cast<const Fp*>(a.rows + (r2 & 0x7FFFFFFFu));
            x1 = ld_fp(p1 + comp); y1 = ld_fp(p1 + LANES + comp);
            x2 = ld_fp(p2 + comp); y2 = ld_fp(p2 + LANES + comp);
            const T n1 = FS::neg(y1), n2 = FS::neg(y2);
            GH_UNROLL for (int i = 0; i < NL; i++) {
                y1.l[i] = (r1 >> 31) ? n1.l[i] : y1.l[i];
                y2.l[i] = (r2 >> 31) ? n2.l[i] : y2.l[i];
            }
        } else {
            const uint32_t i1 = ai - a.in_base, i2 = i1 + 1;
            x1 = t64_ld_x(a.in, i1 / TPW, (i1 % TPW) * LANES + (uint32_t)comp); y1 = t64_ld_y(a.in, i1 / TPW, (i1 % TPW) * LANES + (uint32_t)comp);
            x2 = t64_ld_x(a.in, i2 / TPW, (i2 % TPW) * LANES + (uint32_t)comp); y2 = t64_ld_y(a.in, i2 / TPW, (i2 % TPW) * LANES + (uint32_t)comp);
        }
        const bool m1 = x1.l[0] == AFF_MARK, m2 = x2.l[0] == AFF_MARK;          // a marker is written to every lane of its group
        const bool both = !m1 && !m2;
        const bool eqx = FS::eq(x1, x2), eqy = FS::eq(y1, y2), y0 = FS::is_zero(y1);
        const int kind = !both ? 0 : (!eqx ? 1 : ((eqy && !y0) ? 2 : 3));
        const T dx = FS::sub(x2, x1), y2x = FS::dbl(y1), one = FS::one();
        T d, num = FS::sub(y2, y1);
        GH_UNROLL for (int i = 0; i < NL; i++) d.l[i] = kind == 1 ? dx.l[i] : (kind == 2 ? y2x.l[i] : one.l[i]);
        const T dinv = FS::inv(d);
        {
            const T xx = FS::sqr(x1);
            const T n2 = FS::add(FS::add(FS::dbl(xx), xx), CurveA<C, FS>::get(comp));      // 3 x1^2 + a
            GH_UNROLL for (int i = 0; i < NL; i++) num.l[i] = kind == 2 ? n2.l[i] : num.l[i];
        }
        const T lam = FS::mul(num, dinv);
        T x3 = FS::sub(FS::sub(FS::sqr(lam), x1), x2);
        T y3 = FS::sub(FS::mul(lam, FS::sub(x1, x3)), y1);
        const bool take1 = m2, take2 = m1 && !m2;                               // P + marker -> P, marker + Q -> Q (marker + marker: P, a marker)
        GH_UNROLL for (int i = 0; i < NL; i++) {
            x3.l[i] = take1 ? x1.l[i] : (take2 ? x2.l[i] : (kind == 3 ? 0u : x3.l[i]));
            y3.l[i] = take1 ? y1.l[i] : (take2 ? y2.l[i] : (kind == 3 ? 0u : y3.l[i]));
        }
        if (kind == 3) x3.l[0] = AFF_MARK;
        if (act) t64_st_xy(a.out, o / TPW, (o % TPW) * LANES + (uint32_t)comp, x3, y3);
    }
}

template <class C, class FS, bool R0>
__global__ void __launch_bounds__(256, FS::WAVES) aff_round_kernel(AffRoundArgs<C> a) {
    if (a.run_if && *a.run_if == 0) return;
    constexpr uint32_t TPW = 64 / FS::LANES;
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const bool live = lane < TPW * FS::LANES;
    const uint32_t t = wave * TPW + (live ? lane / FS::LANES : TPW - 1);
    AffRoundLane<C, FS, R0>::run(a, t, (int)(lane % FS::LANES), live);
}
#endif

}  // namespace gh

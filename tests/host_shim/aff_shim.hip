// aff_shim.hip -- CPU emulation of the affine-round bucket accumulation (ginger-lib_amd/csrc/aff_kernels.h):
// the SAME __host__ __device__ lane bodies the GPU kernels run, driven lane by lane on the host, so that the
// index arithmetic and every case of the group law can be checked without a GPU (tests/test_aff_host.py).
// Test infrastructure; built host-only (hipcc --offload-host-only), never linked into the product.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "../../ginger-lib_amd/csrc/aff_kernels.h"

using namespace gh;

template <class C>
static int run_tree(const uint64_t* bases_xy, size_t n_bases, const uint32_t* sorted, size_t n_entries, const uint32_t* starts,
                    const uint32_t* counts, uint32_t total, int R, uint32_t waves, uint32_t bmin, uint64_t* out_xyz,
                    uint32_t* n_marks) {
    typedef typename C::F F;
    typedef F1S<typename C::PF> FS;
    std::vector<Aff<C>> in0(n_bases);
    for (size_t i = 0; i < n_bases; i++) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(bases_xy + 24 * i);
        in0[i].x = F::from_abi(w);
        in0[i].y = F::from_abi(w + 24);
    }
    const size_t stride = (total + 63) & ~(size_t)63;
    std::vector<uint32_t> cnt((size_t)R * stride, 0), st((size_t)R * stride, 0), nout(R + 1, 0);
    for (uint32_t b = 0; b < total; b++) aff_counts_body(counts, b, R, stride, cnt.data());
    nout[0] = (uint32_t)n_entries;
    for (int r = 1; r <= R; r++) {
        uint32_t run = 0;
        for (uint32_t b = 0; b < total; b++) { st[(size_t)(r - 1) * stride + b] = run; run += cnt[(size_t)(r - 1) * stride + b]; }
        nout[r] = run;
    }
    auto tiles = [](uint32_t n_el) { return (size_t)(n_el + 63) / 64 + 1; };
    std::vector<uint4> bufA(t64_bytes(tiles(nout[1]), T64_PT_CHUNKS) / 16), bufB(t64_bytes(tiles(R >= 2 ? nout[2] : 0), T64_PT_CHUNKS) / 16);
    std::vector<uint4> prefix(t64_bytes(tiles(nout[1]), T64_FP_CHUNKS) / 16);
    std::vector<uint4> st1(bufA.size()), st2(bufA.size());
    const void* in = nullptr;
    *n_marks = 0;
    for (int r = 0; r < R; r++) {
        const uint32_t n_out = nout[r + 1];
        void* out = (r & 1) ? (void*)bufB.data() : (void*)bufA.data();
        if (n_out) {
            const uint32_t* st_in = r == 0 ? starts : st.data() + (size_t)(r - 1) * stride;
            const uint32_t* m_in = r == 0 ? counts : cnt.data() + (size_t)(r - 1) * stride;
            std::vector<uint32_t> desc(n_out);
            for (uint32_t o = 0; o < n_out; o++) desc[o] = aff_desc_body(st_in, m_in, st.data() + (size_t)r * stride, total, o);
            AffRoundArgs<C> a;
            a.rows = in0.data(); a.in = in; a.sorted = r == 0 ? sorted : nullptr; a.desc = desc.data(); a.n_out = n_out; a.in_base = 0;
            a.prefix = prefix.data(); a.out = out; a.stage1 = st1.data(); a.stage2 = st2.data(); a.groups = waves * 64; a.bmin = bmin;
            for (uint32_t t = 0; t < a.groups; t++) {
                if (r == 0) AffRoundLane<C, FS, true>::run(a, t, 0, true);
                else AffRoundLane<C, FS, false>::run(a, t, 0, true);
            }
            for (uint32_t o = 0; o < n_out; o++) if (t64_ld_x(out, o >> 6, o & 63u).l[0] == AFF_MARK) (*n_marks)++;
        }
        in = out;
    }
    // finish: what msm_accumulate_kernel<.., AFFIN = true> does (markers skipped)
    const uint32_t* stR = st.data() + (size_t)(R - 1) * stride;
    const uint32_t* mR = cnt.data() + (size_t)(R - 1) * stride;
    for (uint32_t b = 0; b < total; b++) {
        Proj<C> acc = proj_zero<C>();
        for (uint32_t k = 0; k < mR[b]; k++) {
            const uint32_t e = stR[b] + k;
            const Aff<C> q{t64_ld_x(in, e >> 6, e & 63u), t64_ld_y(in, e >> 6, e & 63u)};
            if (q.x.l[0] == AFF_MARK) continue;
            acc = proj_madd<C, typename C::F>(acc, q);
        }
        uint32_t* w = reinterpret_cast<uint32_t*>(out_xyz + 36 * (size_t)b);
        F::to_abi(w, acc.x); F::to_abi(w + 24, acc.y); F::to_abi(w + 48, acc.z);
    }
    return 0;
}

extern "C" int aff_tree_host(int curve, const uint64_t* bases_xy, size_t n_bases, const uint32_t* sorted, size_t n_entries,
                             const uint32_t* starts, const uint32_t* counts, uint32_t total, int R, uint32_t waves, uint32_t bmin,
                             uint64_t* out_xyz, uint32_t* n_marks) {
    if (curve == 0) return run_tree<Mnt4G1>(bases_xy, n_bases, sorted, n_entries, starts, counts, total, R, waves, bmin, out_xyz, n_marks);
    if (curve == 2) return run_tree<Mnt6G1>(bases_xy, n_bases, sorted, n_entries, starts, counts, total, R, waves, bmin, out_xyz, n_marks);
    return -1;
}

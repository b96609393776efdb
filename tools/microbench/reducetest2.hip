#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../ginger-lib_amd/csrc/msm_kernels.h"
using namespace gh;
typedef Mnt4G1 C;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

template <int MODE>
__global__ void __launch_bounds__(64) k(const Proj<C>* buckets, uint32_t nb, uint32_t nbp, Proj<C>* out) {
    extern __shared__ uint32_t lds_raw[];
    Proj<C>* sh = reinterpret_cast<Proj<C>*>(lds_raw);
    const int lane = threadIdx.x;
    const uint32_t slot0 = lane * MSM_REDUCE_L;
    Proj<C> run = proj_zero<C>(), wacc = proj_zero<C>();
    for (int i = MSM_REDUCE_L - 1; i >= 0; i--) {
        const uint32_t slot = slot0 + i;
        if (slot < nb && slot > 0) run = proj_add_call<C>(run, ld_proj<C>(buckets + slot));
        if (i > 0) wacc = proj_add_call<C>(wacc, run);
    }
    if (MODE >= 1) {
        Proj<C> S = run;
        for (int off = 1; off < 64; off <<= 1) {
            st_proj<C>(sh + lane, S);
            __syncthreads();
            if (lane + off < 64) S = proj_add_call<C>(S, ld_proj<C>(sh + lane + off));
            __syncthreads();
        }
        run = S;
    }
    if (MODE >= 2) {
        if (lane >= 1) {
            Proj<C> T = run;
            for (int d = 0; d < 3; d++) T = proj_dbl_call<C>(T);
            wacc = proj_add_call<C>(wacc, T);
        }
    }
    if (MODE >= 3) wacc = wave_tree_sum<C>(wacc, sh, lane);
    st_proj<C>(out + lane, wacc);
    st_proj<C>(out + 64 + lane, run);
}
int main() {
    Proj<C>*buckets, *out;
    CK(hipMalloc(&buckets, 1024 * sizeof(Proj<C>)));
    CK(hipMalloc(&out, 128 * sizeof(Proj<C>)));
    CK(hipMemset(buckets, 0, 1024 * sizeof(Proj<C>)));
    size_t lds = 64 * sizeof(Proj<C>);
#define RUN(M) printf("mode %d ...", M); fflush(stdout); hipLaunchKernelGGL((k<M>), dim3(2), dim3(64), lds, 0, (const Proj<C>*)buckets, 9u, 512u, out); CK(hipDeviceSynchronize()); printf(" done\n"); fflush(stdout);
    RUN(0) RUN(1) RUN(2) RUN(3)
    return 0;
}

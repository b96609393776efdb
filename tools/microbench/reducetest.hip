// Repro harness for the bucket-reduction kernels on zero (infinity) and random data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../../ginger-lib_amd/csrc/msm_kernels.h"
using namespace gh;
typedef Mnt4G1 C;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
int main(int argc, char** argv) {
    int W = 2; uint32_t nb = 9, nbp = 512, segs = 1;
    Proj<C>*buckets, *seg_run, *seg_wacc, *wsums;
    CK(hipMalloc(&buckets, W * nb * sizeof(Proj<C>)));
    CK(hipMalloc(&seg_run, W * segs * sizeof(Proj<C>)));
    CK(hipMalloc(&seg_wacc, W * segs * sizeof(Proj<C>)));
    CK(hipMalloc(&wsums, W * sizeof(Proj<C>)));
    CK(hipMemset(buckets, 0, W * nb * sizeof(Proj<C>)));
    size_t lds = 64 * sizeof(Proj<C>);
    printf("reduce1 on infinity buckets ..."); fflush(stdout);
    hipLaunchKernelGGL((msm_reduce1_kernel<C>), dim3(W * segs), dim3(64), lds, 0, (const Proj<C>*)buckets, nb, nbp, seg_run, seg_wacc);
    CK(hipDeviceSynchronize());
    printf(" done\nreduce2 ..."); fflush(stdout);
    hipLaunchKernelGGL((msm_reduce2_kernel<C>), dim3(W), dim3(64), lds, 0, (const Proj<C>*)seg_run, (const Proj<C>*)seg_wacc, segs, 9, wsums);
    CK(hipDeviceSynchronize());
    printf(" done\n"); fflush(stdout);
    return 0;
}

// Does an out-of-line device function call of proj_add work on this stack (gfx950, ROCm 7.2)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../ginger-lib_amd/csrc/ec29.h"
using namespace gh;
typedef Mnt4G1 C;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

__global__ void k_call(Proj<C>* io, int mode) {
    extern __shared__ uint32_t raw[];
    Proj<C>* sh = reinterpret_cast<Proj<C>*>(raw);
    int lane = threadIdx.x;
    Proj<C> a = io[lane], b = io[64 + lane];
    if (mode == 0) {
        a = proj_add_call<C>(a, b);
    } else if (mode == 1) {
        if (lane & 1) a = proj_add_call<C>(a, b);
    } else if (mode == 2) {
        for (int off = 32; off > 0; off >>= 1) {
            if (lane >= off && lane < 2 * off) sh[lane] = a;
            __syncthreads();
            if (lane < off) a = proj_add_call<C>(a, sh[lane + off]);
            __syncthreads();
        }
    } else if (mode == 3) {
        for (int i = 0; i < 4; i++) { a = proj_add_call<C>(a, b); if (i > 0) b = proj_add_call<C>(b, a); }
    }
    io[128 + lane] = a;
}

int main() {
    Proj<C>* d;
    CK(hipMalloc(&d, 192 * sizeof(Proj<C>)));
    CK(hipMemset(d, 0, 192 * sizeof(Proj<C>)));  // all-zero points: Z = 0 -> infinity fast paths
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {  // nonzero garbage coordinates: exercises the full formula
            Proj<C> h[128];
            for (int i = 0; i < 128; i++) {
                uint32_t* w = (uint32_t*)&h[i];
                for (unsigned k = 0; k < sizeof(Proj<C>) / 4; k++) w[k] = (i * 2654435761u + k * 40503u) & 0x0FFFFFFF;
            }
            CK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
        }
        for (int mode = 0; mode < 4; mode++) {
            printf("pass %d mode %d ...", pass, mode); fflush(stdout);
            hipLaunchKernelGGL(k_call, dim3(4), dim3(64), 64 * sizeof(Proj<C>), 0, d, mode);
            CK(hipDeviceSynchronize());
            printf(" done\n"); fflush(stdout);
        }
    }
    return 0;
}

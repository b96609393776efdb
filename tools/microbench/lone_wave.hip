// What a lone wave per SIMD loses, per kind of field operation (DESIGN.md section 4, G2 rounds at one wave per SIMD):
// the same dependent chain of operations timed with 1 and with 2 resident waves per SIMD.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iginger-lib_amd/csrc -o build/lone_wave tools/microbench/lone_wave.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "fp29.h"
using namespace gh;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// fp_mul with the a*b terms and the m*p terms on separate accumulators (joined when the column closes; no overflow possible)
template <class P> __device__ __forceinline__ Fp fp_mul_2c(const Fp& a, const Fp& b) {
    uint32_t m[NL];
    Fp r;
    uint64_t acc = 0, accm = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) accm += (uint64_t)m[i] * P::P[k - i];
        m[k] = (((uint32_t)acc + (uint32_t)accm) * P::INV) & LM;
        accm += (uint64_t)m[k] * P::P[0];
        acc = (acc + accm) >> LB;
        accm = 0;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL; k++) {
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) accm += (uint64_t)m[i] * P::P[k - i];
        const uint64_t s = acc + accm;
        r.l[k - NL] = (uint32_t)s & LM;
        acc = s >> LB;
        accm = 0;
    }
    fp_cond_sub<P>(r);
    return r;
}

// add / sub with the carry rippling through 13 packed 58-bit pairs instead of 26 limbs (half the dependent chain)
template <class P> __device__ __forceinline__ Fp fp_add_packed(const Fp& a, const Fp& b) {
    uint64_t t[NL / 2];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < NL / 2; i++) {
        const uint64_t x = ((uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << LB)) + ((uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << LB)) + c;
        t[i] = x & ((1ull << (2 * LB)) - 1);
        c = x >> (2 * LB);
    }
    // t - p with borrow, select
    uint64_t d[NL / 2];
    int64_t bw = 0;
#pragma unroll
    for (int i = 0; i < NL / 2; i++) {
        const int64_t x = (int64_t)t[i] - (int64_t)((uint64_t)P::P[2 * i] | ((uint64_t)P::P[2 * i + 1] << LB)) + bw;
        d[i] = (uint64_t)x & ((1ull << (2 * LB)) - 1);
        bw = x >> 63;
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < NL / 2; i++) {
        const uint64_t v = bw ? t[i] : d[i];
        r.l[2 * i] = (uint32_t)v & LM;
        r.l[2 * i + 1] = (uint32_t)(v >> LB);
    }
    return r;
}

template <int KIND> __global__ void __launch_bounds__(64) chain_kernel(uint32_t* io, int iters) {
    Fp x, y, z, w;
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    for (int i = 0; i < NL; i++) {
        x.l[i] = (io[i] + t * 7u) & (LM >> 2); y.l[i] = (io[i + NL] ^ t) & (LM >> 2);
        z.l[i] = (io[i] * 3u + t) & (LM >> 2); w.l[i] = (io[i] * 5u + 11u * t) & (LM >> 2);
    }
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) { x = fp_mul<P4>(x, y); }                                               // plain product, one chain
        if (KIND == 1) { Fp t = fp_mul2<P4>(x, y, z, w); w = z; z = y; y = x; x = t; }         // dual product, every operand loop-variant
        if (KIND == 6) { x = fp_mul_2c<P4>(x, y); }                                            // plain product, two chains
        if (KIND == 7) { Fp t = fp_mul<P4>(x, y); y = x; x = t; }                              // plain product, operands rotate
        if (KIND == 2) { x = fp_add<P4>(x, y); x = fp_sub<P4>(x, z); x = fp_add<P4>(x, w); x = fp_sub<P4>(x, y); }   // serial carry chains only
        if (KIND == 3) { Fp t = fp_mul2<P4>(x, y, z, w); t = fp_sub<P4>(t, y); t = fp_add<P4>(t, z); w = z; z = y; y = x; x = t; }       // the rounds' mix
        if (KIND == 4) { x = fp_mul_small<P4, 13>(x); }
        if (KIND == 8) { x = fp_add_packed<P4>(x, y); x = fp_add_packed<P4>(x, z); x = fp_add_packed<P4>(x, w); x = fp_add_packed<P4>(x, y); }
        if (KIND == 9) { x = fp_add<P4>(x, y); x = fp_add<P4>(x, z); x = fp_add<P4>(x, w); x = fp_add<P4>(x, y); }
        if (KIND == 5) { x = fp_mul<P4>(x, y); z = fp_mul<P4>(z, w); }                         // two independent products
    }
    for (int i = 0; i < NL; i++) io[(size_t)t * NL + i + 64] = x.l[i] ^ z.l[i];
}

template <int KIND> static void run(const char* name, uint32_t* d, int iters, int cus) {
    for (int waves = 1; waves <= 2; waves++) {
        const int blocks = cus * 4 * waves;
        // one wave per block; LDS reservation keeps exactly `waves` blocks per SIMD: 160 KB / CU -> 4 or 8 blocks per CU
        const size_t lds = waves == 1 ? 36 * 1024 : 18 * 1024;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(chain_kernel<KIND>, dim3(blocks), dim3(64), lds, 0, d, 8);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(chain_kernel<KIND>, dim3(blocks), dim3(64), lds, 0, d, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s waves/SIMD %d: %8.3f ms  %8.2f us per iteration and wave\n", name, waves, ms, ms * 1e3 / iters);
    }
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    uint32_t* d;
    CK(hipMalloc(&d, (size_t)cus * 8 * 64 * NL * 4 + 4096));
    CK(hipMemset(d, 0x5a, (size_t)cus * 8 * 64 * NL * 4 + 4096));
    printf("%s, %d CUs\n", p.name, cus);
    run<0>("fp_mul (one accumulator chain)", d, 2000, cus);
    run<7>("fp_mul, operands rotate", d, 2000, cus);
    run<6>("fp_mul, a*b and m*p on two accumulators", d, 2000, cus);
    run<5>("2 x fp_mul, independent", d, 1000, cus);
    run<1>("fp_mul2 (three chains)", d, 1500, cus);
    run<2>("add, sub, add, sub (carry chains)", d, 4000, cus);
    run<3>("fp_mul2 + sub + add", d, 1500, cus);
    run<4>("fp_mul_small<13>", d, 4000, cus);
    run<9>("4 x fp_add", d, 4000, cus);
    run<8>("4 x fp_add, carries through 13 packed pairs", d, 4000, cus);
    return 0;
}

// Micro-benchmarks that fix the arithmetic design on gfx950:
//   (1) raw VALU instruction rates that a 753-bit Montgomery multiply can be built from
//   (2) whole Fp-mul variants (32-bit CIOS, 29-bit reduced radix FIPS, u128 12-limb)
// Build: hipcc -O3 --offload-arch=gfx950 -o mb tools/microbench/mb.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "../../ginger-lib_amd/csrc/constants_gen.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef unsigned __int128 u128;

// ------------------------------------------------------------------ raw rates
#define RAW_ITERS 2048
template <int OP>
__global__ void __launch_bounds__(256) raw_kernel(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u + seed;
    uint64_t acc[8];
    double dacc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { acc[k] = a * (k + 1) + ((uint64_t)b << 32); dacc[k] = (double)(a + k); }
    double da = (double)a * 1.0000001, db = (double)b * 0.99999;
    for (int it = 0; it < RAW_ITERS; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (OP == 0) {  // v_mad_u64_u32
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
            } else if (OP == 1) {  // v_mul_lo_u32
                uint32_t x = (uint32_t)acc[k];
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(a));
                acc[k] = x;
            } else if (OP == 2) {  // v_mul_hi_u32
                uint32_t x = (uint32_t)acc[k];
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(a));
                acc[k] = x;
            } else if (OP == 3) {  // v_fma_f64
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dacc[k]) : "v"(da), "v"(db));
            } else if (OP == 4) {  // v_mad_u32_u24
                uint32_t x = (uint32_t)acc[k];
                asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
                acc[k] = x;
            } else if (OP == 5) {  // v_add_co_u32 + v_addc_co_u32 pair (64-bit add)
                uint32_t lo = (uint32_t)acc[k], hi = (uint32_t)(acc[k] >> 32);
                asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
                acc[k] = lo | ((uint64_t)hi << 32);
            } else if (OP == 6) {  // v_add_u32 (full-rate reference)
                uint32_t x = (uint32_t)acc[k];
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));
                acc[k] = x;
            } else if (OP == 7) {  // v_mul_hi_u32_u24
                uint32_t x = (uint32_t)acc[k];
                asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(a));
                acc[k] = x;
            } else if (OP == 8) {  // v_lshrrev_b64
                asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(acc[k]));
            } else if (OP == 9) {  // v_alignbit_b32
                uint32_t x = (uint32_t)acc[k];
                asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(x) : "v"(a));
                acc[k] = x;
            } else if (OP == 10) {  // mad + addc pattern (mad carry-out consumed)
                uint32_t c = (uint32_t)(acc[(k + 1) & 7]);
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[k]), "+v"(c) : "v"(a), "v"(b) : "vcc");
                a ^= c;
            } else if (OP == 11) {  // v_dot4_u32_u8
                uint32_t x = (uint32_t)acc[k];
                asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
                acc[k] = x;
            } else if (OP == 12) {  // v_add_f64
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(dacc[k]) : "v"(da));
            } else if (OP == 13) {  // v_mad_u64_u32 with SGPR carry-out dest (not vcc) and constant multiplier in sgpr
                asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[k]) : "v"(a), "s"(0x12345677u) : "s20", "s21");
            }
        }
    }
    uint64_t s = 0;
    double ds = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { s += acc[k]; ds += dacc[k]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ (uint32_t)ds;
}

// ------------------------------------------------------------------ Fp-mul variants
__device__ __constant__ uint32_t c_p6_32[24] = GH_P6_P_32;
__device__ __constant__ uint32_t c_p4_32[24] = GH_P4_P_32;

struct P6 { static constexpr uint32_t inv32 = GH_P6_INV32; static __device__ __forceinline__ const uint32_t* p() { return c_p6_32; } };
struct P4 { static constexpr uint32_t inv32 = GH_P4_INV32; static __device__ __forceinline__ const uint32_t* p() { return c_p4_32; } };

// Variant A: CIOS on 24x32-bit limbs.
template <class P>
__device__ __forceinline__ void mul_cios32(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    const uint32_t* p = P::p();
    uint32_t t[26];
#pragma unroll
    for (int i = 0; i < 26; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 24; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 24; j++) {
            uint64_t x = (uint64_t)a[i] * b[j] + t[j] + c;
            t[j] = (uint32_t)x; c = x >> 32;
        }
        uint64_t x = (uint64_t)t[24] + c;
        t[24] = (uint32_t)x; t[25] = (uint32_t)(x >> 32);
        uint32_t m = t[0] * P::inv32;
        c = ((uint64_t)m * p[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 24; j++) {
            uint64_t y = (uint64_t)m * p[j] + t[j] + c;
            t[j - 1] = (uint32_t)y; c = y >> 32;
        }
        x = (uint64_t)t[24] + c;
        t[23] = (uint32_t)x; t[24] = t[25] + (uint32_t)(x >> 32);
    }
    // conditional subtract
    uint32_t d[24]; uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < 24; i++) { uint64_t x = (uint64_t)t[i] - p[i] - bw; d[i] = (uint32_t)x; bw = (x >> 32) & 1; }
#pragma unroll
    for (int i = 0; i < 24; i++) r[i] = bw ? t[i] : d[i];
}

// Variant C: 26 x 29-bit limbs, column-wise (FIPS) with one 64-bit accumulator.
// Montgomery radix is 2^(29*26) = 2^754.  Here we only benchmark throughput, p is given in radix 2^29.
#define RB 29
#define RN 26
#define RMASK ((1u << RB) - 1)
struct P6r {
    uint32_t p[RN]; uint32_t inv;
};
__device__ __constant__ P6r c_p6r;
__device__ __forceinline__ void mul_rr29(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint32_t m[RN];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < RN; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * c_p6r.p[k - i];
        m[k] = ((uint32_t)acc * c_p6r.inv) & RMASK;
        acc += (uint64_t)m[k] * c_p6r.p[0];
        acc >>= RB;
    }
#pragma unroll
    for (int k = RN; k < 2 * RN; k++) {
#pragma unroll
        for (int i = k - RN + 1; i < RN; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - RN + 1; i < RN; i++) acc += (uint64_t)m[i] * c_p6r.p[k - i];
        r[k - RN] = (uint32_t)acc & RMASK;
        acc >>= RB;
    }
}

// Variant D: 12 x 64-bit limbs via unsigned __int128 (what the reference does on CPU; compiler lowers).
__device__ __constant__ uint64_t c_p6_64[12] = GH_P6_P_64;
__device__ __forceinline__ void mul_u128(uint64_t* r, const uint64_t* a, const uint64_t* b) {
    uint64_t t[14];
#pragma unroll
    for (int i = 0; i < 14; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) { u128 x = (u128)a[i] * b[j] + t[j] + c; t[j] = (uint64_t)x; c = (uint64_t)(x >> 64); }
        u128 x = (u128)t[12] + c; t[12] = (uint64_t)x; t[13] = (uint64_t)(x >> 64);
        uint64_t m = t[0] * GH_P6_INV64;
        c = (uint64_t)(((u128)m * c_p6_64[0] + t[0]) >> 64);
#pragma unroll
        for (int j = 1; j < 12; j++) { u128 y = (u128)m * c_p6_64[j] + t[j] + c; t[j - 1] = (uint64_t)y; c = (uint64_t)(y >> 64); }
        x = (u128)t[12] + c; t[11] = (uint64_t)x; t[12] = t[13] + (uint64_t)(x >> 64);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) r[i] = t[i];
}

#define MUL_ITERS 64
template <int V, int NW>
__global__ void __launch_bounds__(256) mul_kernel(uint32_t* io) {
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (V == 0 || V == 3) {
        uint32_t a[24], b[24];
#pragma unroll
        for (int i = 0; i < 24; i++) { a[i] = io[tid * 64 + i]; b[i] = io[tid * 64 + 32 + i]; }
        for (int it = 0; it < MUL_ITERS; it++) {
            if (V == 0) { mul_cios32<P6>(a, a, b); mul_cios32<P6>(b, a, b); }
            else { mul_cios32<P4>(a, a, b); mul_cios32<P4>(b, a, b); }
        }
#pragma unroll
        for (int i = 0; i < 24; i++) io[tid * 64 + i] = a[i] ^ b[i];
    } else if (V == 1) {
        uint32_t a[RN], b[RN];
#pragma unroll
        for (int i = 0; i < RN; i++) { a[i] = io[tid * 64 + i] & RMASK; b[i] = io[tid * 64 + 32 + i] & RMASK; }
        for (int it = 0; it < MUL_ITERS; it++) { mul_rr29(a, a, b); mul_rr29(b, a, b); }
#pragma unroll
        for (int i = 0; i < RN; i++) io[tid * 64 + i] = a[i] ^ b[i];
    } else if (V == 2) {
        uint64_t a[12], b[12];
        uint64_t* io64 = (uint64_t*)io;
#pragma unroll
        for (int i = 0; i < 12; i++) { a[i] = io64[tid * 32 + i]; b[i] = io64[tid * 32 + 16 + i]; }
        for (int it = 0; it < MUL_ITERS; it++) { mul_u128(a, a, b); mul_u128(b, a, b); }
#pragma unroll
        for (int i = 0; i < 12; i++) io64[tid * 32 + i] = a[i] ^ b[i];
    }
}

template <typename F>
static float time_it(F launch, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    int ncu = prop.multiProcessorCount;
    uint32_t* d; size_t nthreads = (size_t)ncu * 8 * 256;
    CK(hipMalloc(&d, nthreads * 64 * 4));
    std::vector<uint32_t> h(nthreads * 64);
    uint64_t s = 88172645463325252ULL;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s; }
    // keep operands < 2^753 in both layouts
    for (size_t t = 0; t < nthreads; t++) { h[t * 64 + 23] &= 0xFFFF; h[t * 64 + 32 + 23] &= 0xFFFF; }
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // radix-29 modulus
    {
        P6r pr; uint32_t p32[24] = GH_P6_P_32;
        for (int i = 0; i < RN; i++) {
            int bit = i * RB; uint64_t v = 0;
            for (int k = 0; k < 3; k++) { int w = bit / 32 + k; if (w < 24) v |= (w - bit / 32) < 2 ? ((uint64_t)p32[w] << (32 * (w - bit / 32))) : 0; }
            pr.p[i] = (uint32_t)((v >> (bit % 32)) & RMASK);
        }
        // inv = -p^-1 mod 2^29
        uint32_t x = 1; for (int i = 0; i < 6; i++) x *= 2 - pr.p[0] * x;
        pr.inv = (0u - x) & RMASK;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(c_p6r), &pr, sizeof(pr)));
    }

    const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_fma_f64", "v_mad_u32_u24", "add_co+addc pair",
                           "v_add_u32", "v_mul_hi_u32_u24", "v_lshrrev_b64", "v_alignbit_b32", "mad_u64+addc pair", "v_dot4_u32_u8",
                           "v_add_f64", "v_mad_u64_u32 sgpr"};
    int grid = ncu * 8;  // 8 blocks of 256 = 32 waves/CU = 8 waves/SIMD
#define RAW(OP) { float ms = time_it([&] { raw_kernel<OP><<<grid, 256>>>(d, 1); }, 5); \
        double ops = (double)grid * 256 * RAW_ITERS * 8; \
        printf("raw %-22s %8.3f ms  %8.2f Gop/s(lane)  %6.2f cyc/wave-instr/SIMD @2.4GHz\n", names[OP], ms, ops / ms / 1e6, \
               2.4e9 * (ms / 1e3) / (ops / 64 / (ncu * 4))); }
    RAW(6) RAW(0) RAW(13) RAW(1) RAW(2) RAW(3) RAW(12) RAW(4) RAW(7) RAW(5) RAW(8) RAW(9) RAW(10) RAW(11)

#define MUL(V, NAME, BLK) { int g = ncu * BLK; float ms = time_it([&] { mul_kernel<V, 0><<<g, 256>>>(d); }, 3); \
        double muls = (double)g * 256 * MUL_ITERS * 2; \
        printf("fpmul %-28s blocks/CU %d  %8.3f ms  %8.2f G Fp-mul/s\n", NAME, BLK, ms, muls / ms / 1e6); }
    MUL(0, "cios32 p6", 8) MUL(0, "cios32 p6", 4) MUL(0, "cios32 p6", 2)
    MUL(3, "cios32 p4", 8)
    MUL(1, "rr29 fips p6", 8) MUL(1, "rr29 fips p6", 4) MUL(1, "rr29 fips p6", 2)
    MUL(2, "u128 12x64 p6", 8) MUL(2, "u128 12x64 p6", 4)
    CK(hipFree(d));
    return 0;
}

// scan_kernels.h -- exclusive prefix sum over u32 counters (bucket histogram -> bucket offsets).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gh {

// ---------------------------------------------------------------- 2. scans (generic u32 exclusive scan, 3 kernels)
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;  // per thread
__global__ void __launch_bounds__(SCAN_BLOCK) scan_partials_kernel(const uint32_t* in, uint32_t* block_sums, size_t n) {
    __shared__ uint32_t sh[SCAN_BLOCK];
    size_t base = ((size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x) * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) s += in[base + k];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = SCAN_BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0];
}
// single block scans the block sums in place (exclusive); nblocks can be large -> loop
__global__ void __launch_bounds__(1024) scan_block_sums_kernel(uint32_t* block_sums, size_t nblocks) {
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t running;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (size_t base = 0; base < nblocks; base += 1024) {
        size_t i = base + threadIdx.x;
        uint32_t v = i < nblocks ? block_sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            uint32_t t = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        uint32_t incl = sh[threadIdx.x];
        uint32_t r = running;
        if (i < nblocks) block_sums[i] = r + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) running = r + incl;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(SCAN_BLOCK) scan_final_kernel(const uint32_t* in, const uint32_t* block_sums, uint32_t* out, size_t n) {
    __shared__ uint32_t sh[SCAN_BLOCK];
    size_t base = ((size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x) * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = base + k < n ? in[base + k] : 0; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        uint32_t t = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = block_sums[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { if (base + k < n) out[base + k] = run; run += v[k]; }
}


}  // namespace gh

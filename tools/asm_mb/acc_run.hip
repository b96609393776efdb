// Runs the assembly G1 accumulation kernel (build/gh_asm.hsaco) on SYNTHETIC inputs to separate the kernel's own issue rate from
// the workload's shape: uniform or Poisson-like list lengths, small (cache-resident) or large (TLB-hostile) tables.
//   hipcc -O2 --offload-arch=gfx950 -o build/asm_mb/acc_run tools/asm_mb/acc_run.hip
//   build/asm_mb/acc_run build/gh_asm.hsaco <log2 table rows> <log2 tasks> <entries per task> <0 uniform | 1 spread>
// The limbs are random 29-bit words, not curve points: the update's instruction stream does not depend on the values.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <algorithm>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void fill(uint32_t* p, size_t n, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    uint32_t v = (uint32_t)x & 0x1FFFFFFFu;
    if (i % 26 == 25) v &= 0x03FFFFFFu;   // top limb small: values below p
    p[i] = v;
}
struct Task { uint32_t beg, cnt; uint64_t dst; };
int main(int argc, char** argv) {
    if (argc < 6) { printf("usage\n"); return 1; }
    const int log_rows = atoi(argv[2]), log_tasks = atoi(argv[3]), K = atoi(argv[4]), spread = atoi(argv[5]);
    const char* kname = argc > 6 ? argv[6] : "gh_asm_acc_g1_p4";
    FILE* f = fopen(argv[1], "rb");
    if (!f) { printf("no code object\n"); return 1; }
    std::vector<char> co; char buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) co.insert(co.end(), buf, buf + n);
    fclose(f);
    hipModule_t m; CK(hipModuleLoadData(&m, co.data()));
    hipFunction_t fn; CK(hipModuleGetFunction(&fn, m, kname));
    const size_t rows = (size_t)1 << log_rows, tasks = (size_t)1 << log_tasks;
    uint32_t *d_tab, *d_sorted, *d_salts; Task* d_tasks; uint32_t* d_out;
    CK(hipMalloc(&d_tab, rows * 208));
    CK(hipMalloc(&d_salts, 2 * 208));
    CK(hipMalloc(&d_out, tasks * 312));
    fill<<<(unsigned)((rows * 52 + 255) / 256), 256>>>(d_tab, rows * 52, 1);
    fill<<<1, 256>>>(d_salts, 104, 2);
    std::vector<uint32_t> cnt(tasks);
    uint64_t x = 88172645463325252ull; size_t total = 0;
    for (size_t t = 0; t < tasks; t++) {
        uint32_t c = K;
        if (spread) {   // roughly Poisson(K): sum of K/4 draws of {2..6}
            c = 0;
            for (int j = 0; j < K / 4; j++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; c += 2 + (uint32_t)(x % 5); }
        }
        cnt[t] = c;
    }
    if (spread) std::sort(cnt.begin(), cnt.end(), [](uint32_t a, uint32_t b) { return a > b; });
    std::vector<Task> ht(tasks);
    for (size_t t = 0; t < tasks; t++) { ht[t].beg = (uint32_t)total; ht[t].cnt = cnt[t]; total += cnt[t]; }
    std::vector<uint32_t> hs(total + 1);
    for (size_t i = 0; i < total; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hs[i] = (uint32_t)((x >> 11) & (rows - 1)) | ((uint32_t)(x >> 63) << 31); }
    CK(hipMalloc(&d_sorted, (total + 1) * 4));
    CK(hipMalloc(&d_tasks, tasks * sizeof(Task)));
    for (size_t t = 0; t < tasks; t++) ht[t].dst = (uint64_t)(uintptr_t)(d_out + t * 78);
    CK(hipMemcpy(d_sorted, hs.data(), (total + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tasks, ht.data(), tasks * sizeof(Task), hipMemcpyHostToDevice));
    struct { const void *bases, *sorted, *tasks, *salts; uint32_t n, pad; } args = {d_tab, d_sorted, d_tasks, d_salts, (uint32_t)tasks, 0};
    size_t size = sizeof args;
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    float best = 1e30f;
    for (int rep = 0; rep < 6; rep++) {
        CK(hipEventRecord(e0, 0));
        CK(hipModuleLaunchKernel(fn, (unsigned)((tasks + 255) / 256), 1, 1, 256, 1, 1, 0, 0, nullptr, extra));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2 && ms < best) best = ms;
    }
    const double updates = (double)total - tasks;   // the first entry of a task is a copy
    const double simds = prop.multiProcessorCount * 4.0;
    const double cyc = best * 1e-3 * prop.clockRate * 1e3 / (updates / 64.0 / simds);
    printf("%s rows 2^%d (%.0f MB) tasks 2^%d entries %d %s: %.3f ms, %.0f updates, %.0f SIMD cycles per wave-update (epilogue included)\n", kname,
           log_rows, rows * 208 / 1e6, log_tasks, K, spread ? "spread+sorted" : "uniform", best, updates, cyc);
    return 0;
}

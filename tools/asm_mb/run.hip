// Times the kernels of build/asm_mb/asm_mb.hsaco (tools/asm_mb/gen.py): 512 blocks of 256 threads = two waves per SIMD on 256 CUs.
//   hipcc -O2 --offload-arch=gfx950 -o build/asm_mb/run tools/asm_mb/run.hip && build/asm_mb/run build/asm_mb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    std::string dir = argc > 1 ? argv[1] : "build/asm_mb";
    int iters = argc > 2 ? atoi(argv[2]) : 200;
    FILE* f = fopen((dir + "/asm_mb.hsaco").c_str(), "rb");
    if (!f) { printf("no code object\n"); return 1; }
    std::vector<char> co;
    char buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) co.insert(co.end(), buf, buf + n);
    fclose(f);
    hipModule_t m; CK(hipModuleLoadData(&m, co.data()));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    printf("device %s, %d CUs, %.2f GHz; blocks = 2 per CU; iters %d\n", prop.gcnArchName, cus, ghz, iters);
    FILE* t = fopen((dir + "/asm_mb.txt").c_str(), "r");
    char name[128]; int units, ninstr;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    bool warmed = false;
    while (fscanf(t, "%127s %d %d", name, &units, &ninstr) == 3) {
        hipFunction_t fn; CK(hipModuleGetFunction(&fn, m, name));
        struct { unsigned iters, pad; } args = {(unsigned)iters, 0};
        size_t size = sizeof args;
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
        if (!warmed) {   // the clocks ramp up over the first few hundred ms of load: the first kernels of a run read 25 % slow without this
            for (int r = 0; r < 300; r++) CK(hipModuleLaunchKernel(fn, cus * 2, 1, 1, 256, 1, 1, 0, 0, nullptr, extra));
            CK(hipDeviceSynchronize());
            warmed = true;
        }
        for (int waves = 2; waves >= 1; waves--) {
            const int blocks = cus * waves;     // LDS (79 872 B per block) caps a CU at two blocks = two waves per SIMD
            float ms = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0, 0));
                CK(hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, 0, nullptr, extra));
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                if (t < ms) ms = t;
            }
            // per SIMD: `waves` waves x iters x block; cycles per block-instance per SIMD slot
            const double cyc = ms * 1e-3 * ghz * 1e9 / ((double)iters * waves);
            printf("%-24s waves/SIMD %d: %8.3f ms  %9.0f cycles per block and wave-slot  %6.2f cycles/instr (%d instr)  %8.0f cycles per unit\n",
                   name, waves, ms, cyc, cyc / ninstr, ninstr, cyc / units);
        }
    }
    return 0;
}

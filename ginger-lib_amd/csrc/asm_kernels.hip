// asm_kernels.hip -- loader and launcher of the hand-allocated gfx950 assembly kernels (ginger-lib_amd/asmgen/*.py).
//
// The kernels are generated, assembled and linked into ONE code object at build time (asmgen/build.py ->
// build/gh_asm.hsaco) and embedded into this library as bytes (build/asm_blob.S, .incbin); here the code object is
// handed to hipModuleLoadData once per process and its kernels are launched with hipModuleLaunchKernel on the caller's
// stream.  They replace hipcc-compiled kernels one for one (same inputs, same outputs: csrc/msm_kernels.h names the
// counterpart next to each), so every parity test runs through them; GH_ACC_ASM=0 selects the hipcc kernels for A/B runs.
#include "asm_kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

extern "C" const unsigned char gh_asm_hsaco[];
extern "C" const unsigned char gh_asm_hsaco_end[];

namespace gh_asm {

using gh_rt::g;
using gh_rt::g_err;

namespace {
constexpr int NTT_ASM_KMIN = 6, NTT_ASM_KMAX = 8;
struct State {
    bool tried = false;
    hipModule_t mod = nullptr;
    hipFunction_t acc_g1[2] = {nullptr, nullptr};   // [0]: p4 (MNT4-753 G1), [1]: p6 (MNT6-753 G1)
    hipFunction_t aff[4][2][2] = {};                // [kind: f2, f3, f1p4, f1p6][fwd][r0]
    hipFunction_t mb_mulpair = nullptr;
    hipFunction_t ntt[2][3] = {};                   // [p4, p6][k - 6]
};
State s;

int load_locked() {
    if (s.tried) return s.mod ? GH_OK : GH_E_HIP;
    s.tried = true;
    const size_t bytes = (size_t)(gh_asm_hsaco_end - gh_asm_hsaco);
    if (bytes < 64 || memcmp(gh_asm_hsaco, "\177ELF", 4) != 0) {
        g_err = "assembly code object missing from the library (build/gh_asm.hsaco was not embedded)";
        return GH_E_HIP;
    }
    hipModule_t m = nullptr;
    hipError_t e;
    // GH_ASM_HSACO=<file>: another build of the same kernels (asmgen/build.py) instead of the embedded one -- A/B runs of
    // generator variants on one card in one process start (measurement knob; the default is the embedded code object)
    static std::vector<char> override_co;
    if (const char* path = getenv("GH_ASM_HSACO")) {
        FILE* fco = fopen(path, "rb");
        if (!fco) { g_err = std::string("GH_ASM_HSACO: cannot open ") + path; return GH_E_BAD_ARG; }
        char buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, fco)) > 0) override_co.insert(override_co.end(), buf, buf + got);
        fclose(fco);
        e = hipModuleLoadData(&m, override_co.data());
    } else {
        e = hipModuleLoadData(&m, gh_asm_hsaco);
    }
    if (e != hipSuccess) {
        g_err = std::string("hipModuleLoadData(assembly kernels) failed: ") + hipGetErrorString(e);
        return GH_E_HIP;
    }
    static const char* names[2] = {"gh_asm_acc_g1_p4", "gh_asm_acc_g1_p6"};
    for (int i = 0; i < 2; i++) {
        e = hipModuleGetFunction(&s.acc_g1[i], m, names[i]);
        if (e != hipSuccess) {
            g_err = std::string("hipModuleGetFunction(") + names[i] + ") failed: " + hipGetErrorString(e);
            hipModuleUnload(m);
            return GH_E_HIP;
        }
    }
    static const char* kinds[4] = {"f2", "f3", "f1p4", "f1p6"};
    for (int t = 0; t < 4; t++)
        for (int fw = 0; fw < 2; fw++)
            for (int r0 = 0; r0 < 2; r0++) {
                char nm[64];
                snprintf(nm, sizeof nm, "gh_asm_aff_%s_%s_%s", kinds[t], fw ? "fwd" : "bwd", r0 ? "r0" : "rn");
                e = hipModuleGetFunction(&s.aff[t][fw][r0], m, nm);
                if (e != hipSuccess) {
                    g_err = std::string("hipModuleGetFunction(") + nm + ") failed: " + hipGetErrorString(e);
                    hipModuleUnload(m);
                    return GH_E_HIP;
                }
            }
    for (int f = 0; f < 2; f++)
        for (int k = NTT_ASM_KMIN; k <= NTT_ASM_KMAX; k++) {
            char nm[64];
            snprintf(nm, sizeof nm, "gh_asm_ntt_p%d_k%d", f ? 6 : 4, k);
            e = hipModuleGetFunction(&s.ntt[f][k - NTT_ASM_KMIN], m, nm);
            if (e != hipSuccess) {
                g_err = std::string("hipModuleGetFunction(") + nm + ") failed: " + hipGetErrorString(e);
                hipModuleUnload(m);
                return GH_E_HIP;
            }
        }
    e = hipModuleGetFunction(&s.mb_mulpair, m, "gh_asm_mb_mulpair");
    if (e != hipSuccess) {
        g_err = std::string("hipModuleGetFunction(gh_asm_mb_mulpair) failed: ") + hipGetErrorString(e);
        hipModuleUnload(m);
        return GH_E_HIP;
    }
    s.mod = m;
    g.at_shutdown.push_back([] {
        if (s.mod) hipModuleUnload(s.mod);
        s = State();
    });
    return GH_OK;
}
}  // namespace

bool enabled() {
    static const bool on = !(getenv("GH_ACC_ASM") && atoi(getenv("GH_ACC_ASM")) == 0);
    return on;
}

int acc_g1_launch(int prime, const void* bases, const uint32_t* sorted, const AccTask* tasks, const void* salts,
                  uint32_t n_tasks, hipStream_t st) {
    if (n_tasks == 0) return GH_OK;
    if (int rc = load_locked()) return rc;
    struct {
        const void* bases;
        const void* sorted;
        const void* tasks;
        const void* salts;
        uint32_t n_tasks;
        uint32_t pad;
    } args = {bases, sorted, tasks, salts, n_tasks, 0};
    size_t size = sizeof args;
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    HIPCHK(hipModuleLaunchKernel(s.acc_g1[prime == 6 ? 1 : 0], (n_tasks + 255) / 256, 1, 1, 256, 1, 1, 0, st, nullptr, extra));
    return GH_OK;
}

bool aff_enabled() {
    static const bool on = !(getenv("GH_AFF_ASM") && atoi(getenv("GH_AFF_ASM")) == 0);
    return on;
}

bool aff_g1_enabled() {
    static const bool on = !(getenv("GH_AFF_ASM_G1") && atoi(getenv("GH_AFF_ASM_G1")) == 0);
    return on && aff_enabled();
}

int aff_launch(int tower, bool fwd, bool r0, const AffArgs& a, uint32_t waves, hipStream_t st) {
    if (waves == 0 || a.n_out == 0) return GH_OK;
    if (tower < 0 || tower > 3) { g_err = "internal: aff_launch kind"; return GH_E_BAD_ARG; }
    // what the kernels' 32-bit index arithmetic assumes (asmgen/g2_rounds.py): element indices times 4 and a wave's list span
    if ((waves & 3u) || a.n_out >= (1u << 30) || (uint64_t)a.B * 13312u >= (1ull << 32) || a.B == 0) {
        g_err = "internal: affine round outside the assembly kernels' index range";
        return GH_E_UNSUPPORTED;
    }
    if (int rc = load_locked()) return rc;
    AffArgs args = a;
    size_t size = sizeof args;
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    HIPCHK(hipModuleLaunchKernel(s.aff[tower][fwd ? 1 : 0][r0 ? 1 : 0], waves / 4, 1, 1, 256, 1, 1, 0, st, nullptr, extra));
    return GH_OK;
}

bool ntt_enabled() {
    static const bool on = !(getenv("GH_NTT_ASM") && atoi(getenv("GH_NTT_ASM")) == 0);
    return on;
}

// 32-bit byte offsets into the data (96 B per element) and the factor tables (104 B): N <= 2^25
bool ntt_supported(int log_n, int k) { return k >= NTT_ASM_KMIN && k <= NTT_ASM_KMAX && log_n >= 8 && log_n <= 25 && k <= log_n; }

int ntt_pass_launch(int prime, int k, const NttAsmArgs& a, hipStream_t st) {
    if (!ntt_supported((int)a.log_n, k) || a.log_ns + (uint32_t)k > a.log_n || a.n_waves != (1u << (a.log_n - 8)) ||
        (a.post_stride != 0 && a.post_stride != 104)) {
        g_err = "internal: NTT pass outside the assembly kernel's range";
        return GH_E_UNSUPPORTED;
    }
    if (int rc = load_locked()) return rc;
    NttAsmArgs args = a;
    size_t size = sizeof args;
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    HIPCHK(hipModuleLaunchKernel(s.ntt[prime == 6 ? 1 : 0][k - NTT_ASM_KMIN], (a.n_waves + 3) / 4, 1, 1, 256, 1, 1, 0, st, nullptr, extra));
    return GH_OK;
}

int measure_fpmul_peak(double* products_per_s, hipStream_t st) {
    if (int rc = load_locked()) return rc;
    const uint32_t iters = 200, blocks = (uint32_t)g.num_cus * 2u;      // two blocks of 4 waves per CU = two waves per SIMD
    struct { uint32_t iters, pad; } args = {iters, 0};
    size_t size = sizeof args;
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    float best = 0;
    for (int rep = 0; rep < 4; rep++) {        // the first launch pays the code upload
        HIPCHK(hipEventRecord(e0, st));
        HIPCHK(hipModuleLaunchKernel(s.mb_mulpair, blocks, 1, 1, 256, 1, 1, 0, st, nullptr, extra));
        HIPCHK(hipEventRecord(e1, st));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && (best == 0 || ms < best)) best = ms;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (products_per_s) *products_per_s = best > 0 ? (double)blocks * 256.0 * 2.0 * iters / (best * 1e-3) : 0.0;
    return GH_OK;
}

int kernel_resources(const char* which, uint32_t* scratch, uint32_t* vgprs, uint32_t* lds) {
    if (int rc = load_locked()) return rc;
    hipFunction_t f = nullptr;
    const std::string w = which ? which : "";
    if (w == "g1_acc_p4") f = s.acc_g1[0];
    else if (w == "g1_acc_p6") f = s.acc_g1[1];
    else if (w == "g2_f2_fwd_r0") f = s.aff[0][1][1];
    else if (w == "g2_f2_bwd_r0") f = s.aff[0][0][1];
    else if (w == "g2_f2_bwd_rn") f = s.aff[0][0][0];
    else if (w == "g2_f3_fwd_r0") f = s.aff[1][1][1];
    else if (w == "g2_f3_bwd_r0") f = s.aff[1][0][1];
    else if (w == "g2_f3_bwd_rn") f = s.aff[1][0][0];
    else if (w == "ntt_p4_k8") f = s.ntt[0][2];
    else if (w == "ntt_p6_k8") f = s.ntt[1][2];
    else { g_err = "unknown kernel name"; return GH_E_BAD_ARG; }
    int v = 0;
    HIPCHK(hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, f));
    if (scratch) *scratch = (uint32_t)v;
    HIPCHK(hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_NUM_REGS, f));
    if (vgprs) *vgprs = (uint32_t)v;
    HIPCHK(hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, f));
    if (lds) *lds = (uint32_t)v;
    return GH_OK;
}

}  // namespace gh_asm

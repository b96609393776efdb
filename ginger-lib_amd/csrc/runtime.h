// runtime.h -- process-wide device context shared by the translation units of libginger_hip.so
// (one TU per curve so that hipcc can build them in parallel; see __graft_entry__.py build()).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/ginger_hip.h"
#include "fp29.h"

namespace gh_rt {

struct Domain {
    int log_n = 0;
    gh::Fp* tw = nullptr;          // w^i
    gh::Fp* coset = nullptr;       // g^i
    gh::Fp* coset_inv = nullptr;   // size_inv * g^-i
    gh::Fp size_inv;               // internal form
    gh::Fp* d_size_inv = nullptr;  // the same on the device (the assembly pass reads its final factor from memory)
    uint32_t* scratch = nullptr;
    uint32_t* scratch2 = nullptr;  // second ping-pong vector: an odd number of passes ends in the caller's buffer without a copy
    bool scratch2_failed = false;
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct Ctx {
    bool ready = false;
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;       // transforms, bucket sort
    hipStream_t stream_acc = nullptr;   // MSM accumulation (lowest priority: the filler of the pipeline)
    hipStream_t stream_red = nullptr;   // MSM bucket reduction (highest priority: short latency chains)
    hipStream_t stream_acc2 = nullptr;  // second half of a split affine round (msm_impl.h launch_tree): same priority as stream_acc
    hipEvent_t tev[2] = {nullptr, nullptr};   // fork / join of a split round
    hipEvent_t ev[8];
    hipEvent_t pev[4][8];               // MSM stage events, one set per job in flight (job k of a batch uses set k & 3)
    std::map<int, Domain> domains[2];
    std::map<std::string, DevBuf> pool;
    int window_override = 0;
    int affine_mode = 2;        // G1 bucket sums by affine rounds (aff_kernels.h): 0 never, 1 always, 2 when the list fills the chip
    gh_msm_timing_t last_msm{};
    std::vector<gh_msm_timing_t> batch_tm;   // per-MSM timings of the last batch call
    float last_fft_ms = 0;
    int dedup_mode = 1;                 // gh_msm_set_dedup: add up the scalars of equal bases (keys with a shift table)
    size_t scratch_reserved = 0;        // scratch_guard(): stack-frame scratch the runtime already holds for this context's queues
    std::vector<std::function<void()>> at_shutdown;   // releases of function-local device / pinned allocations
};

extern Ctx g;
extern std::string g_err;

int ensure_init();
std::mutex& api_mutex();     // the lock every ABI entry point holds (one device context per process)
int pool_get(const char* name, size_t bytes, void** out);
size_t pool_cap(const char* name);                 // current capacity of a cached buffer (0 if none)
void pool_release(const char* prefix);             // free every cached buffer whose name starts with prefix ("" = all)
int device_scan(const uint32_t* in, uint32_t* out, size_t n, const char* tmpname, hipStream_t stream = nullptr);   // nullptr: g.stream
int auto_window(size_t n, int deg);
void dist_teardown_locked();                      // dist.hip: gh_shutdown (API lock held) destroys the communicator with the context

// No C++ exception leaves the library (include/ginger_hip.h: "nothing is thrown"; the reference's multi_scalar_mul is
// infallible, variable_base.rs:85-90, and the Rust shim falls back to the CPU path on a non-zero status): every extern "C"
// entry point is a function-try-block whose handler returns api_exception() -- std::bad_alloc -> GH_E_NOMEM, anything else
// -> GH_E_HIP, the message in gh_last_error().  Called from inside a catch handler only.
int api_exception() noexcept;

// Kernels with KB-scale stack frames (the out-of-line EC functions of the cold paths: 2-17 KB per lane, build/*.log) make the
// runtime reserve frame x 64 lanes x resident waves of scratch at dispatch -- up to 9 GB for the MNT6 G2 instances -- and a
// reservation that fails does so inside the runtime's queue handler, which ends the process (the round-3 abort inside
// gh_msm_cached with 640 MB free: DESIGN.md section 9-4b).  scratch_guard() asks the code object for the kernel's frame and
// refuses the launch with GH_E_NOMEM when the card cannot hold the reservation; GH_LAUNCH is hipLaunchKernelGGL behind it.
int scratch_guard(const void* kernel, size_t threads);
#define GH_LAUNCH(kern, grid, block, shmem, st, ...)                                                              \
    do {                                                                                                          \
        const dim3 g_ = (grid), b_ = (block);                                                                     \
        if (int rc_ = gh_rt::scratch_guard((const void*)(kern), (size_t)g_.x * g_.y * g_.z * b_.x * b_.y * b_.z)) \
            return rc_;                                                                                           \
        hipLaunchKernelGGL(kern, g_, b_, shmem, st, __VA_ARGS__);                                                 \
    } while (0)

#define HIPCHK(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            char b_[512];                                                                    \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            gh_rt::g_err = b_;                                                               \
            return e_ == hipErrorOutOfMemory ? GH_E_NOMEM : GH_E_HIP;                        \
        }                                                                                    \
    } while (0)

// ---- per-curve entry points (msm_<curve>.hip)
struct BasesBase {
    gh_curve_t curve;
    size_t n = 0;
    void* d_points = nullptr;   // n x Aff<C>, internal layout
    uint8_t* d_inf = nullptr;   // n bytes or null
    void* d_table = nullptr;    // precomputed shift table: pre_W rows of n x Aff<C> (row w = 2^(pre_c w) P), or null
    int pre_c = 0, pre_W = 0;   // window bits, rows of the table
    int pre_G = 1;              // bucket sets: row j = 2^(pre_c pre_G j) P, window w = j pre_G + g reads row j and files into set g
                                // (1 = full table, one bucket set; GH_TABLE_ROWS caps the rows: a partial table)
    // equal bases (msm_kernels.h "equal bases"): groups of indices that hold the same point up to sign, found when the shift table is built
    uint32_t* d_dup_starts = nullptr;   // n_dup_groups + 1
    uint32_t* d_dup_members = nullptr;  // base index | negative << 31, the canonical base first
    uint32_t* d_dup_chunks = nullptr;   // 3 per chunk (first member, end, group) followed by n_dup_groups + 1 chunk offsets per group
    uint32_t n_dup_groups = 0, n_dup_members = 0, n_dup_chunks = 0;
    uint8_t aff_asm_off = 0;    // G2: an MSM over this key overflowed the exception list of the assembly rounds (a key with many equal
                                // bases, e.g. a proving key's b_g2_query under an assignment with equal values: every pair of such bases
                                // in a bucket is a doubling) -- later MSMs go straight to the C++ round kernel, which doubles inline
    uint32_t magic = 0x6768424au;
};
struct MsmOps {
    int (*upload)(const uint64_t* bases, const uint8_t* infinity, size_t n, int canonical, BasesBase** out);
    int (*run)(BasesBase* h, const void* d_scalars, size_t n_scalars, uint64_t* out_xyz);
    int (*host)(const uint64_t* bases, const uint8_t* infinity, size_t n_bases, const uint64_t* scalars,
                size_t n_scalars, uint64_t* out_xyz);
    int (*proj_add)(uint64_t* acc_xyz, const uint64_t* p_xyz);
    int (*to_affine)(const uint64_t* xyz, uint64_t* out_xy, uint8_t* is_infinity);
    int (*precompute)(BasesBase* h, int window_bits, int max_rows);
    int (*batch)(BasesBase* const* hs, const void* const* d_scalars, const size_t* n_scalars, int count, uint64_t* out_xyz);
    int (*proj_mul)(const uint64_t* p_xyz, const uint64_t* scalar12, uint64_t* out_xyz);
    int (*proj_neg)(uint64_t* xyz);
    int (*acc_lists)(const void* points, const uint32_t* sorted, const uint32_t* starts, const uint32_t* counts,
                     const uint32_t* order, uint32_t total, void* out_proj, hipStream_t st);
};
const MsmOps* msm_ops_mnt4753_g1();
const MsmOps* msm_ops_mnt4753_g2();
const MsmOps* msm_ops_mnt6753_g1();
const MsmOps* msm_ops_mnt6753_g2();

// ---- transforms (ntt.hip)
int fft_run(gh_field_t field, void* d_data, uint32_t log_n, uint32_t flags);
int vec_op(gh_field_t field, int op, void* d_a, const void* d_b, const uint64_t* scalar12, size_t n);
int witness_map(gh_field_t field, void* d_a, void* d_b, void* d_c, uint32_t log_n, const uint64_t* d1,
                const uint64_t* d2, const uint64_t* d3, void* d_h);
int sap_witness_map(gh_field_t field, void* d_a, void* d_c, uint32_t log_n, const uint64_t* d1, const uint64_t* d2, void* d_h);
int batch_inverse(gh_field_t field, void* d_a, size_t n);
int lagrange_coefficients(gh_field_t field, uint32_t log_n, const uint64_t* tau12, void* d_out);

}  // namespace gh_rt

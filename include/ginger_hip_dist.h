/* ginger_hip_dist.h -- the one exchange step of the multi-GPU MSM, behind the C ABI.
 *
 * The reference has no distributed component (SURVEY.md sections 5 and 8e).  sum_i s_i P_i is a sum in a
 * commutative group: the (base, scalar) pairs are cut into contiguous shards, one process per GPU runs a
 * complete single-GPU MSM over its shard (gh_msm_resident*, ginger_hip.h), and the partial sums are combined.
 * EC addition is not an RCCL reduction operator (a limb-wise ncclSum of projective coordinates is
 * meaningless), so the "all-reduce of partial sums" is ONE all-gather of the raw projective limbs
 * (36 * deg u64 = 288 / 576 / 864 bytes per rank for G1 / MNT4-G2 / MNT6-G2) followed by a
 * (world - 1)-addition fold in rank order on every rank (add_assign, short_weierstrass_projective.rs:574-617).
 * The affine image of the result does not depend on the shard count.
 *
 * Transport: RCCL over xGMI (gh_dist_init_rccl; librccl.so is loaded on first use, a single-GPU process never
 * loads it), or any all-gather the host already has (gh_dist_init_custom: MPI, a Rust host's own channel, or
 * the gloo-backed callback the CPU tests use).  One communicator per process, like the device context.
 */
#ifndef GINGER_HIP_DIST_H
#define GINGER_HIP_DIST_H

#include "ginger_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GH_DIST_UNIQUE_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */
#define GH_E_DIST (-7)              /* RCCL / transport failure; see gh_last_error() */

/* Rank 0 creates the id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by whatever channel
 * the launcher has (torchrun's store, MPI_Bcast, a file). */
int gh_dist_unique_id(void* out_id128);
/* Collective over all ranks: ncclCommInitRank on the GPU this process is bound to (gh_init). */
int gh_dist_init_rccl(const void* id128, int rank, int world);

/* Bring-your-own transport: fn must gather `bytes` bytes from every rank into recv (rank-major) on all ranks
 * and return 0 on success. */
typedef int (*gh_allgather_fn)(void* ctx, const void* send, void* recv, size_t bytes);
int gh_dist_init_custom(gh_allgather_fn fn, void* ctx, int rank, int world);

/* Loads and binds librccl without creating anything (no collective): lets every rank find out LOCALLY whether RCCL is
 * usable before the ranks agree -- over the launcher's own channel -- to enter the collective gh_dist_init_rccl, where a
 * rank that failed alone would leave the others waiting.  Resolution: an RCCL already mapped into the process (a host that
 * imported torch carries torch's copy; two RCCLs in one process are one too many), else $GH_RCCL_PATH, else
 * $ROCM_PATH/lib/librccl.so.1 (/opt/rocm), else the bare soname. */
int gh_dist_probe_rccl(void);

/* rank / world of the communicator (world as RCCL reports it: ncclCommCount); GH_E_DIST if none. */
int gh_dist_info(int* rank, int* world);
/* What actually carries the exchange: *rccl_ranks = ncclCommCount of the live RCCL communicator, 0 when the transport is a
 * custom callback (or there is none); *rccl_version = ncclGetVersion of the bound librccl (0: none bound); path = the file
 * its symbols come from.  Any pointer may be NULL. */
int gh_dist_transport(int* rccl_ranks, int* rccl_version, char* path, size_t path_cap);

/* out = partial(rank 0) + partial(rank 1) + ... + partial(world - 1), the same on every rank.
 * partial_xyz / out_xyz: projective MSM results in the ABI format (3 * 12 * deg u64); they may alias.
 * *exchange_us (may be NULL) receives the duration of the all-gather alone, microseconds. */
int gh_partials_allgather_fold(gh_curve_t curve, const uint64_t* partial_xyz, uint64_t* out_xyz, double* exchange_us);

/* The same for `count` partial sums at once (the results of a pipelined batch of MSMs, gh_msm_resident_dev_batch): ONE
 * all-gather of count * 36 * deg u64 per rank, then count folds.  partials_xyz / outs_xyz: count x (36 * deg) u64, may
 * alias.  *exchange_us: the all-gather(s) alone. */
int gh_partials_allgather_fold_batch(gh_curve_t curve, const uint64_t* partials_xyz, size_t count, uint64_t* outs_xyz, double* exchange_us);

/* Destroys the communicator (also done by gh_shutdown). */
int gh_dist_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif /* GINGER_HIP_DIST_H */

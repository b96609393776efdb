/* ginger_hip.h -- C ABI of the MI355X (gfx950) implementation of ginger-lib's Groth16 prover hot
 * path: variable-base multi-scalar multiplication over MNT4-753 / MNT6-753 G1 and G2, and the
 * radix-2 evaluation-domain transforms over their scalar fields.
 *
 * The reference (ZencashOfficial/ginger-lib, 100 % Rust) has no FFI; these entry points are what
 * a `#[cfg(feature = "gpu")]` dispatch inside the two reference modules would bind
 * (INTEGRATION.md shows the Rust `extern "C"` block and the safe wrappers):
 *
 *   algebra/src/msm/variable_base.rs:85-90   VariableBaseMSM::multi_scalar_mul   -> gh_msm_*
 *   algebra/src/fft/domain.rs:65-94          EvaluationDomain::new                -> gh_domain_supported
 *   algebra/src/fft/domain.rs:113-138        fft / fft_in_place / ifft(_in_place) -> gh_fft_*
 *   algebra/src/fft/domain.rs:155-179        coset_fft / coset_ifft (_in_place)   -> gh_fft_* + GH_FFT_COSET
 *   algebra/src/fft/domain.rs:245-256        divide_by_vanishing_poly_on_coset_in_place -> gh_vec_scale_*
 *   algebra/src/fft/domain.rs:289-302        mul_polynomials_in_evaluation_domain -> gh_vec_mul_*
 *   proof-systems/src/groth16/r1cs_to_qap.rs:121-166  witness_map (transform part) -> gh_witness_map(_dev)
 *
 * Data formats (exactly the reference's in-memory values, marshalled field by field because the
 * Rust structs are not repr(C) -- SURVEY.md section 8b):
 *   field element  : 12 little-endian uint64_t limbs (BigInteger768, biginteger/macros.rs:4),
 *                    MONTGOMERY form with R = 2^768 (fp_768.rs:24-30) for curve coordinates and
 *                    FFT data; CANONICAL integers (< r) for MSM scalars (what `into_repr()` yields,
 *                    groth16/prover.rs:241-267).
 *   G1 affine base : x || y                       = 24 u64, plus one uint8_t infinity flag
 *   MNT4 G2 base   : x.c0 x.c1 y.c0 y.c1          = 48 u64        (Fq2, fields/models/fp2.rs)
 *   MNT6 G2 base   : x.c0 x.c1 x.c2 y.c0 y.c1 y.c2 = 72 u64       (Fq3, fields/models/fp3.rs)
 *   MSM result     : homogeneous projective X || Y || Z (each one base-field element, same
 *                    component order), Z == 0 <=> infinity (short_weierstrass_projective.rs:285-290,
 *                    :372-390).  As in the reference, only the affine image X/Z, Y/Z is canonical.
 *
 * Every function returns GH_OK (0) or a negative GH_E_* code; nothing is thrown, no host pointer
 * is retained after return.  Calls are serialised by an internal lock and may come from any thread.
 * There is NO CPU fallback inside this library: without a usable gfx950 device every call fails
 * with GH_E_NO_DEVICE (the Rust shim decides what to do with a non-zero status).
 */
#ifndef GINGER_HIP_H
#define GINGER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GH_OK 0
#define GH_E_BAD_ARG (-1)       /* null pointer, inconsistent lengths */
#define GH_E_UNSUPPORTED (-2)   /* domain too large for the field's 2-adicity (domain.rs:69-71) or for memory */
#define GH_E_NO_DEVICE (-3)     /* no gfx950 device / HIP runtime unusable */
#define GH_E_HIP (-4)           /* a HIP runtime call failed; see gh_last_error() */
#define GH_E_NOMEM (-5)
#define GH_E_BAD_HANDLE (-6)

#define GH_FFT_INVERSE 1u /* ifft: use group_gen^-1 and multiply by size_inv (domain.rs:134-138) */
#define GH_FFT_COSET 2u   /* coset variant: distribute_powers with g = 17 (domain.rs:140-179) */

/* Curves / fields are selected by id so that one entry point serves the four instantiations. */
typedef enum {
    GH_MNT4753_G1 = 0, /* base field p4, scalars mod p6                           */
    GH_MNT4753_G2 = 1, /* base field Fq2 = p4[X]/(X^2-13)                         */
    GH_MNT6753_G1 = 2, /* base field p6, scalars mod p4                           */
    GH_MNT6753_G2 = 3  /* base field Fq3 = p6[X]/(X^3-11)                         */
} gh_curve_t;

typedef enum {
    GH_MNT4753_FR = 0, /* = MNT6-753 Fq, 2-adicity 30 (fields/mnt6753/fq.rs:93)   */
    GH_MNT6753_FR = 1  /* = MNT4-753 Fq, 2-adicity 15 (fields/mnt4753/fq.rs:94)   */
} gh_field_t;

/* ---- lifecycle ------------------------------------------------------------------------- */
/* Bind this process to one GPU.  devices == NULL selects $LOCAL_RANK (or 0).  Only devices[0]
 * is used: the deployment model is one process per GPU (multi-GPU MSM: ginger_hip_dist.h /
 * the Python launcher shard pairs across ranks and fold the partial sums). Idempotent. */
int gh_init(const int* devices, int n_devices);
int gh_shutdown(void);
const char* gh_last_error(void);
/* Human-readable device line ("AMD Instinct MI355X, 256 CUs, gfx950"); valid until gh_shutdown. */
const char* gh_device_name(void);

/* ---- multi-scalar multiplication -------------------------------------------------------- */
/* out = sum_{i < min(n_bases, n_scalars)} scalars[i] * bases[i]
 * Semantics of variable_base.rs:10-83: zip-truncation to the shorter input, zero scalars and
 * infinity bases contribute nothing, empty input -> (0, 1, 0).
 *   bases      n_bases  * (24 * deg) u64   (deg = 1, 2, 3 by curve)
 *   infinity   n_bases  bytes (non-zero = point at infinity), may be NULL (= none)
 *   scalars    n_scalars * 12 u64, canonical
 *   out_xyz    3 * 12 * deg u64                                                          */
int gh_msm(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases,
           const uint64_t* scalars, size_t n_scalars, uint64_t* out_xyz);

/* Bases are static per proving key (groth16/mod.rs:158-170): upload once, reuse per proof. */
typedef struct gh_bases* gh_bases_t;
int gh_bases_upload(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases,
                    gh_bases_t* out_handle);
/* The same from the reference's serialised form, so that a proving key file (Parameters::write,
 * groth16/mod.rs:188-239) can go to the device query by query without a Montgomery conversion on the host:
 * n_points records of  x || y || infinity  with every base-field coefficient as 96 little-endian bytes of its
 * canonical integer (GroupAffine::write, short_weierstrass_projective.rs:185-192; Fp768::write, fp_768.rs:784-789),
 * i.e. 193 / 385 / 577 bytes per G1 / MNT4-G2 / MNT6-G2 point.  GH_E_BAD_ARG where FromBytes would fail
 * (a coefficient >= p, a flag byte other than 0 / 1).                                                  */
int gh_bases_upload_wire(gh_curve_t curve, const uint8_t* bytes, size_t n_points, gh_bases_t* out_handle);
/* Synthetic key (benchmarks, full-size tests): n distinct resident bases P_i = P_0 + i * H along
 * an addition chain, generated on the device from two affine points (x || y, Montgomery).  Both curves' G1 have
 * cofactor 1 and multiples of the G2 generator stay in the subgroup, so every P_i is a valid base.        */
int gh_bases_generate_chain(gh_curve_t curve, const uint64_t* p0_xy, const uint64_t* step_xy, size_t n, gh_bases_t* out_handle);
/* Copy `count` resident bases starting at `first` back to the host (x || y Montgomery limbs, as gh_bases_upload takes them). */
int gh_bases_download(gh_bases_t handle, size_t first, size_t count, uint64_t* out_xy);
int gh_bases_free(gh_bases_t handle);
size_t gh_bases_len(gh_bases_t handle);
/* Optional, once per resident key: build the shift table 2^(c w) P_i, w = 0 .. floor(752/c), in
 * device memory (W x the footprint of the bases; GH_E_NOMEM if it does not fit, and the handle
 * stays usable without it).  Later gh_msm_resident* calls on the handle then file all windows into
 * ONE bucket set: fewer additions per pair (c = 21 at 2^20 pairs: 36 instead of 48), one bucket
 * reduction, no window fold.  Results are the same group element as without the table (the affine
 * image is what variable_base.rs:85-90 + into_affine() define).  window_bits 0 = choose from n.
 * GH_E_UNSUPPORTED if a base has 2-power order (2^(c w) P = infinity has no affine form).   */
int gh_bases_precompute(gh_bases_t handle, int window_bits);
int gh_bases_precomputed_window(gh_bases_t handle); /* c of the table, 0 if none */
/* The same with at most max_rows rows (a PARTIAL table, for keys whose full table does not fit next to the others: four
 * 2^24-base G1 queries are 4 x 126 GB at c = 21, eight rows of each are 4 x 28 GB).  Row j holds 2^(c G j) P_i with
 * G = ceil(windows / max_rows); window w = j G + g reads row j and files into bucket set g, so there are G bucket sets
 * instead of one (or one per window), folded with c doublings each.  2^24 pairs, c = 21: full table 320 ms, 8 rows 330 ms,
 * 4 rows 340 ms, no table 391 ms per MSM.  max_rows 0 = no cap (gh_bases_precompute); window_bits 0 = 21 for a capped table of
 * 2^20 bases or more, the full table's choice otherwise.  Same results, same error conventions as gh_bases_precompute. */
int gh_bases_precompute_rows(gh_bases_t handle, int window_bits, int max_rows);
int gh_bases_table_rows(gh_bases_t handle);         /* rows of the table, 0 if none */
/* scalars on the host */
int gh_msm_resident(gh_bases_t handle, const uint64_t* scalars, size_t n_scalars, uint64_t* out_xyz);

/* ---- content-addressed resident keys: the drop-in for an UNCHANGED caller of VariableBaseMSM::multi_scalar_mul
 * (algebra/src/msm/variable_base.rs:85-90), which hands over (bases, scalars) slices on every call.
 * gh_msm_cached has gh_msm's signature and gh_msm's result -- a pure function of its arguments -- but remembers the
 * bases it has seen: they are identified by FOUR 64-bit hash lanes over EVERY coordinate limb and infinity flag of the first
 * min(n_bases, n_scalars) bases (never by address: a buffer reused for other bases is another key).  Two lanes select the
 * cache entry, the other two -- an independent pair -- must match as well before the resident copy is used; all four are keyed
 * with per-process random seeds, so colliding inputs cannot be prepared in advance.  Residual assumption: two different
 * base vectors do not agree in all 256 bits (`collisions` counts entries that agreed in the first 128 only: they are misses).
 * A repeat moves only the scalars, a new key is uploaded like gh_msm does.  From the table_after-th sighting of a key
 * (default 2; >= 4096 bases) its shift table (gh_bases_precompute) is built; if the table cannot be built (memory) the key
 * stays on the per-window path and the call still succeeds.  The cache is bounded: least recently used keys are freed once
 * the device bytes held exceed max_bytes (default: half of what hipMemGetInfo reports free at the first call); it is
 * emptied by gh_key_cache_clear and by gh_shutdown.
 * rust/algebra-hip-sys binds multi_scalar_mul to this entry point.                                  */
typedef struct {
    uint64_t entries, bytes;          /* keys resident now, device bytes they hold (points + flags + shift tables) */
    uint64_t hits, misses, evictions, tables_built;
    uint64_t collisions;              /* lookups that matched an entry's selection lanes but not its verification lanes */
} gh_key_cache_stats_t;
int gh_msm_cached(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases,
                  const uint64_t* scalars, size_t n_scalars, uint64_t* out_xyz);
#define GH_KEY_CACHE_AUTO ((size_t)-1)   /* max_bytes: half of what hipMemGetInfo reports free at the next gh_msm_cached */
int gh_key_cache_config(size_t max_bytes, int table_after /* 0 = never build tables */);
int gh_key_cache_clear(void);
int gh_key_cache_stats(gh_key_cache_stats_t* out);
/* The hash gh_msm_cached keys on (low half returned, high half through *hi if not NULL).  Host-only: needs no device. */
uint64_t gh_bases_content_hash(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases, uint64_t* hi);
/* All four lanes of a key's identity: out4[0..1] select the cache entry, out4[2..3] verify it.  Host-only.  Values differ
 * from process to process (random seeds); within a process equal content gives equal lanes. */
int gh_bases_key_id(gh_curve_t curve, const uint64_t* bases, const uint8_t* infinity, size_t n_bases, uint64_t* out4);
/* Fault injection for the test-suite (0 = off; never set by a product caller).  bit 0: the selection lanes of every key
 * identity are constant, i.e. every key collides with every other of its size.  The environment variable
 * GH_TEST_TABLE_NOMEM=1 makes every shift-table build take its out-of-memory path (scratch pool dropped, GH_E_NOMEM). */
int gh_test_hooks(int flags);
/* scalars already in device memory (from gh_dev_alloc); used by the benchmark's HBM-resident timing
 * and by a device-resident prover pipeline.  The call is synchronous on the library stream. */
int gh_msm_resident_dev(gh_bases_t handle, const void* d_scalars, size_t n_scalars, uint64_t* out_xyz);

/* `count` MSMs on one curve, back to back: out_xyz + i * (36 * deg) receives
 *   sum_j d_scalars[i][j] * bases(handles[i])[j]            (each exactly as gh_msm_resident_dev).
 * The prover issues its large MSMs in sequence (groth16/prover.rs:273-325: a_query, b_g1_query,
 * h_query, l_query on G1); given as one batch, the library pipelines them over HIP streams: the
 * bucket sort of MSM i+1 and the bucket reduction + window fold of MSM i-1 run beside the
 * accumulation of MSM i.  handles may repeat; results do not depend on the batching.           */
int gh_msm_resident_dev_batch(const gh_bases_t* handles, const void* const* d_scalars, const size_t* n_scalars, int count,
                              uint64_t* out_xyz);

/* ---- fixed-base multi-scalar multiplication --------------------------------------------------
 * FixedBaseMSM (algebra/src/msm/fixed_base.rs:7-79): out[i] = scalars[i] * g for ONE base g, through a window
 * table of g -- what the Groth16 parameter generator spends its time in (proof-systems/src/groth16/generator.rs:
 * 225-296: every query of the proving key is one such call).
 *   gh_fixed_base_window  = FixedBaseMSM::get_mul_window_size(num_scalars)                          (:7-13)
 *   gh_fixed_base_table   = get_window_table(scalar_size, window, g): g projective (ABI format), kept on the
 *                           device in affine form, ceil(scalar_size / window) rows of 2^window points (:15-43)
 *   gh_fixed_base_msm     = multi_scalar_mul(scalar_size, window, table, v): scalars are CANONICAL integers
 *                           (v[i].into_repr(), 12 u64 each), out_xyz receives n projective points (3 * 12 * deg
 *                           u64 each).  Same group elements as the reference; as there, only into_affine() of a
 *                           result is canonical.                                                    (:45-78) */
typedef struct gh_fixed_table* gh_fixed_table_t;
int gh_fixed_base_window(size_t num_scalars);
int gh_fixed_base_table(gh_curve_t curve, const uint64_t* g_xyz, size_t scalar_size, int window, gh_fixed_table_t* out_table);
int gh_fixed_base_msm(gh_fixed_table_t table, const uint64_t* scalars, size_t n, uint64_t* out_xyz);
/* The same followed by batch_normalization + into_affine on the device (what the parameter generator does to every query:
 * proof-systems/src/groth16/generator.rs:247-335; short_weierstrass_projective.rs:402-442, :663-678): n affine points as
 * x || y coefficients -- Montgomery 2^768 limbs, or with canonical != 0 the plain integers GroupAffine::write serialises
 * (:185-192) -- and n infinity flags (infinity = GroupAffine::zero() = (0, 1, true)).                                    */
int gh_fixed_base_msm_affine(gh_fixed_table_t table, const uint64_t* scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf, int canonical);
int gh_fixed_base_free(gh_fixed_table_t table);

/* Window size override for sweeps (0 = automatic).  Affects subsequent MSM calls. */
int gh_msm_set_window(int c);
/* Equal bases of a key (a proving key holds the same point for every variable with the same polynomial) are found when the
 * key's shift table is built and their scalars are added up before every MSM over it (sum s_i P = (sum s_i) P: the result is
 * the same group element).  0 switches that off for tables built afterwards (tests that want the bucket paths under many
 * equal bases; A/B runs); default 1. */
int gh_msm_set_dedup(int on);
int gh_msm_get_window(gh_curve_t curve, size_t n);
/* Bucket sums in AFFINE coordinates: pairwise rounds over the flat bucket-ordered list, the inversions of a
 * lane's whole batch shared by Montgomery's trick (5 M + 1 S + a share of one safegcd inversion per addition
 * instead of the 11 M of add_assign_mixed); P + P, P - P and sums through infinity handled in place.  Same
 * results as the projective kernels.  mode: 0 never, 1 always, 2 (default) where measured faster: G2 MSMs
 * long enough to fill the chip (G1 ties with its projective kernel and stays on it).                      */
int gh_msm_set_affine(int mode);

/* Time spent by the last MSM call in its phases, milliseconds (device phases by HIP events on the
 * library stream, host fold by a host clock).  Any pointer may be NULL. */
typedef struct {
    float sort_ms;        /* digit extraction + bucket sort */
    float accumulate_ms;  /* msm_accumulate_kernel alone (dominant kernel), HIP events around the launch */
    float heavy_ms;       /* wave-cooperative path for over-long buckets */
    float reduce_ms;      /* bucket running-sum reduction kernels */
    float fold_ms;        /* window fold (host) + D2H of window sums */
    float total_ms;
    int window_bits;
    int num_windows;
    unsigned long long accumulate_madds; /* mixed additions issued (all buckets) */
    unsigned int heavy_buckets;
} gh_msm_timing_t;
int gh_msm_last_timing(gh_msm_timing_t* out);
/* the same for MSM `index` of the last gh_msm_resident_dev_batch call (device phases of neighbouring
 * MSMs overlap there, so the phases of one MSM no longer add up to its share of the wall time) */
int gh_msm_batch_timing(int index, gh_msm_timing_t* out);

/* ---- evaluation-domain transforms -------------------------------------------------------- */
/* 1 if EvaluationDomain::new(num_coeffs) would be Some(..) (domain.rs:65-72), else 0;
 * *log_n receives log2(next_power_of_two(num_coeffs)).                                     */
int gh_domain_supported(gh_field_t field, size_t num_coeffs, uint32_t* log_n);

/* out[0 .. 2^log_n) = transform of `in` padded with zeros / truncated to 2^log_n elements
 * (Vec::resize in domain.rs:121,135).  in/out are host arrays of 12-u64 Montgomery elements and
 * may alias.  flags: 0 = fft, GH_FFT_INVERSE = ifft, GH_FFT_COSET = coset_fft,
 * GH_FFT_INVERSE | GH_FFT_COSET = coset_ifft.  Natural order in and out.                    */
int gh_fft(gh_field_t field, const uint64_t* in, size_t n_in, uint64_t* out, uint32_t log_n, uint32_t flags);

/* Device-resident variant: d_data holds 2^log_n elements in device memory and is transformed in
 * place (a library-owned scratch buffer of the same size is used internally).               */
int gh_fft_dev(gh_field_t field, void* d_data, uint32_t log_n, uint32_t flags);

/* Pointwise helpers on device-resident vectors of n elements (witness_map glue,
 * groth16/r1cs_to_qap.rs:137-166):  a[i] = a[i] * b[i];  a[i] = a[i] - b[i];  a[i] = a[i] * s  */
int gh_vec_mul_dev(gh_field_t field, void* d_a, const void* d_b, size_t n);
int gh_vec_sub_dev(gh_field_t field, void* d_a, const void* d_b, size_t n);
int gh_vec_scale_dev(gh_field_t field, void* d_a, const uint64_t* scalar12, size_t n);
/* Host-buffer forms of the same (copy in, compute, copy out). */
int gh_vec_mul(gh_field_t field, uint64_t* a, const uint64_t* b, size_t n);
int gh_vec_scale(gh_field_t field, uint64_t* a, const uint64_t* scalar12, size_t n);

/* QAP witness map, device resident: the transform part of R1CStoQAP::witness_map
 * (proof-systems/src/groth16/r1cs_to_qap.rs:121-166).  d_a, d_b, d_c hold the 2^log_n evaluations
 * of the A, B, C rows at the assignment (:105-119, :141-151; computed by the caller -- circuit
 * synthesis is CPU scalar code and out of scope); they are overwritten.  d_h receives the
 * 2^log_n + 1 coefficients of h:
 *   a = coset_fft(ifft(a)); b = coset_fft(ifft(b)); ab = a .* b; c = coset_fft(ifft(c));
 *   ab = (ab - c) * (g^N - 1)^-1; ab = coset_ifft(ab);
 *   h[0] = ab[0] - d3 - d1 d2;  h[i] = ab[i] (0 < i < N-1);  h[N-1] = 0;  h[N] = d1 d2
 * (h starts as zeros and `*h_i *= ...` in :124-129 leaves it zero -- reproduced as is).
 * d1, d2, d3: 12-u64 Montgomery field elements on the host.                                  */
int gh_witness_map_dev(gh_field_t field, void* d_a, void* d_b, void* d_c, uint32_t log_n,
                       const uint64_t* d1, const uint64_t* d2, const uint64_t* d3, void* d_h);
/* Host-buffer form: a, b, c of 2^log_n elements each (not modified), h of 2^log_n + 1 elements. */
int gh_witness_map(gh_field_t field, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint32_t log_n,
                   const uint64_t* d1, const uint64_t* d2, const uint64_t* d3, uint64_t* h);

/* GM17 / SAP witness map, device resident: the transform part of R1CStoSAP::witness_map
 * (proof-systems/src/gm17/r1cs_to_sap.rs:194-240).  d_a, d_c hold the 2^log_n evaluations the caller built
 * (:158-192 a, :207-230 c; host scalar code); both are overwritten; d_h receives 2^log_n + 1 coefficients:
 *   a = ifft(a); h = 2 d1 a; h[0] -= d2 + d1^2; a = coset_fft(a); aa = a .* a; c = coset_fft(ifft(c));
 *   aa = coset_ifft((aa - c) / Z); h[i] += aa[i] (i < N - 1); h[N] = d1^2                                   */
int gh_sap_witness_map_dev(gh_field_t field, void* d_a, void* d_c, uint32_t log_n, const uint64_t* d1, const uint64_t* d2, void* d_h);
int gh_sap_witness_map(gh_field_t field, const uint64_t* a, const uint64_t* c, uint32_t log_n, const uint64_t* d1,
                       const uint64_t* d2, uint64_t* h);

/* batch_inversion (algebra/src/fields/mod.rs:412-442): a[i] <- 1 / a[i] for n Montgomery elements, zeros are left alone. */
int gh_batch_inverse_dev(gh_field_t field, void* d_a, size_t n);
int gh_batch_inverse(gh_field_t field, uint64_t* a, size_t n);

/* EvaluationDomain::evaluate_all_lagrange_coefficients (algebra/src/fft/domain.rs:183-219): out[i] = L_i(tau) for the
 * domain of size 2^log_n; tau a Montgomery element; tau inside the domain gives the indicator vector (:189-199). */
int gh_lagrange_coefficients_dev(gh_field_t field, uint32_t log_n, const uint64_t* tau12, void* d_out);
int gh_lagrange_coefficients(gh_field_t field, uint32_t log_n, const uint64_t* tau12, uint64_t* out);

/* Duration of the kernels of the last gh_fft / gh_fft_dev call (HIP events), milliseconds. */
int gh_fft_last_kernel_ms(float* ms);

/* ---- measurement support (no counterpart in the reference) --------------------------------
 * gh_measure_fpmul_peak: 753-bit Montgomery products per second of THIS card, measured now (about 30 ms): every lane of a
 * full-chip launch runs two interleaved products per iteration on the register plan of the hot kernels (asmgen/microbench.py).
 * bench.py prices `valu.frac` against it.  gh_kernel_resources: scratch bytes per lane (stack frame; spills), registers and
 * LDS bytes of a generated kernel as the loaded code object reports them; `which` is one of g1_acc_p4, g1_acc_p6,
 * g2_f2_fwd_r0, g2_f2_bwd_r0, g2_f2_bwd_rn, g2_f3_fwd_r0, g2_f3_bwd_r0, g2_f3_bwd_rn, ntt_p4_k8, ntt_p6_k8 (the 8-stage NTT pass
 * over MNT6-753 Fr / MNT4-753 Fr). */
int gh_measure_fpmul_peak(double* products_per_s);
int gh_kernel_resources(const char* which, uint32_t* scratch_bytes_per_lane, uint32_t* registers, uint32_t* lds_bytes);

/* ---- device memory (plain pointers; for callers that keep vectors resident) -------------- */
int gh_dev_alloc(void** d_ptr, size_t bytes);
int gh_dev_free(void* d_ptr);
int gh_dev_upload(void* d_dst, const void* h_src, size_t bytes);
int gh_dev_download(void* h_dst, const void* d_src, size_t bytes);
int gh_dev_sync(void);
/* Release the library's cached scratch buffers (bucket lists, affine-round lists, staging); they are
 * re-allocated on demand.  For a host that is done with a large key and wants the HBM back; resident
 * bases, shift tables and domain tables are not touched.                                          */
int gh_dev_trim(void);

/* ---- group helpers on the host side of the boundary -------------------------------------- */
/* acc = acc + p for two projective points in the MSM result format (used to fold the per-GPU
 * partial sums after the all-gather: SURVEY.md section 8e).  Runs on the host; no device needed. */
int gh_proj_add(gh_curve_t curve, uint64_t* acc_xyz, const uint64_t* p_xyz);
/* out = scalar * p for ONE projective point (canonical 12-u64 scalar): the handful of single scalar
 * multiplications next to the MSMs in create_proof (groth16/prover.rs:278, :296, :311, :325-327;
 * GroupProjective::mul_assign, short_weierstrass_projective.rs:521-540).  Runs on the host.    */
int gh_proj_mul(gh_curve_t curve, const uint64_t* p_xyz, const uint64_t* scalar12, uint64_t* out_xyz);
/* xyz = -xyz (sub_assign in prover.rs:331 is add of the negation).  Host side. */
int gh_proj_neg(gh_curve_t curve, uint64_t* xyz);
/* x||y (Montgomery) and *is_infinity from a projective result: the reference's into_affine()
 * (short_weierstrass_projective.rs:663-678).  Runs on the host.                              */
int gh_proj_to_affine(gh_curve_t curve, const uint64_t* xyz, uint64_t* out_xy, uint8_t* is_infinity);

#ifdef __cplusplus
}
#endif
#endif /* GINGER_HIP_H */

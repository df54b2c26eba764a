/*
 * friedrich_amd.h -- C ABI of the MI355X-native dense-linear-algebra core for friedrich.
 *
 * This is the drop-in boundary for friedrich's crate-private `algebra` module and for the
 * `nalgebra::Cholesky<f64, Dynamic>` methods `GaussianProcess` calls on its private
 * `covmat_cholesky` field (reference: src/lib.rs:39, src/gaussian_process/mod.rs:78).  The reference
 * has no FFI of its own; every entry point below cites the reference item (file:line, relative to the
 * friedrich 0.5.1 source tree) it replaces.  INTEGRATION.md shows the Rust `extern "C"` block and the
 * replacement bodies a maintainer would add.
 *
 * Conventions
 *  - All matrices are f64, column-major with an explicit leading dimension (nalgebra `DMatrix` layout;
 *    `EMatrix::as_matrix()` has ld = capacity != nrows, src/algebra/extendable_matrix.rs:52-55).
 *  - Every `const double*` / `double*` data argument may be a HOST pointer or a DEVICE (HIP) pointer; the
 *    library classifies it with hipPointerGetAttributes and stages host data itself.
 *  - Every function returns an fr_status; FR_OK == 0.  fr_last_error(ctx) gives the message.  The reference
 *    panics where this ABI returns FR_NOT_POSITIVE_DEFINITE / FR_SINGULAR_SOLVE / FR_SHAPE; the host shim
 *    turns the status back into the reference's panic text.
 *  - Work is enqueued on the context's HIP stream; functions that return host results synchronise that
 *    stream before returning, functions that only touch device memory do not.
 *  - Thread safety: every entry point that takes a context or a factor takes the context's lock (fr_last_error and
 *    fr_ctx_comm_info excepted: they return a pointer into / a snapshot of state the caller must not race with), so any
 *    number of host threads may call into one context / one fr_chol concurrently (friedrich's GaussianProcess is
 *    Send + Sync: concurrent &self predicts are legal); the calls of one context take turns on its streams and
 *    workspaces.  Handles of different contexts are independent and run in parallel -- also on the same GPU: the
 *    persistent solves claim their blocks in order of arrival and need no co-residency (trsv.hip), so a second
 *    context's kernels holding CUs only slow them down.
 */
#ifndef FRIEDRICH_AMD_H
#define FRIEDRICH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FR_ABI_VERSION 2 /* 2: FR_PROF_COUNT = 8; fr_grad_terms collective on collectively created handles */

typedef enum {
    FR_OK = 0,
    FR_NOT_POSITIVE_DEFINITE = 1, /* a pivot was <= 0 / NaN and no usable substitute: algebra/mod.rs:85,90;
                                     multivariate_normal.rs:57.  fr_chol_info() gives the column. */
    FR_SINGULAR_SOLVE = 2,        /* exact zero on the factor's diagonal: mod.rs:203,263,345 */
    FR_UNSUPPORTED_KERNEL = 3,    /* kernel program not expressible on device -> caller keeps the nalgebra path */
    FR_SHAPE = 4,                 /* shape assertion of the reference violated: mod.rs:153,177,178,231,253,293,334,374 */
    FR_INVALID_ARGUMENT = 5,
    FR_OUT_OF_MEMORY = 6,
    FR_HIP_ERROR = 7,
    FR_RCCL_ERROR = 8,
    FR_NO_DEVICE = 9
} fr_status;

/* ---- kernel program: POD description of a friedrich `Kernel` (src/parameters/kernel.rs) -------------
 * Reverse-polish list: leaves push k(x,y), FR_K_SUM / FR_K_PROD pop two and push k1+k2 / k1*k2
 * (KernelSum kernel.rs:132-211, KernelProd :221-307).  params[] follow get_parameters() order. */
typedef enum {
    FR_K_LINEAR = 0,            /* kernel.rs:342-402   [c]                */
    FR_K_POLYNOMIAL = 1,        /* :411-485            [alpha, c, d]      */
    FR_K_SQUAREDEXP = 2,        /* :496-601 (Gaussian) [ls, ampl]         */
    FR_K_EXPONENTIAL = 3,       /* :612-706            [ls, ampl]         */
    FR_K_MATERN1 = 4,           /* :717-813            [ls, ampl]         */
    FR_K_MATERN2 = 5,           /* :824-925            [ls, ampl]         */
    FR_K_HYPERTAN = 6,          /* :934-1001           [alpha, c]         */
    FR_K_MULTIQUADRIC = 7,      /* :1010-1070          [c]                */
    FR_K_RATIONALQUADRATIC = 8, /* :1079-1157          [alpha, ls]        */
    FR_K_SUM = 100,
    FR_K_PROD = 101
} fr_kernel_kind;

#define FR_KPROG_MAX_OPS 15

typedef struct {
    int32_t kind;    /* fr_kernel_kind */
    int32_t nparams; /* number of values in params[] (0 for SUM/PROD) */
    double params[3];
} fr_kernel_op;

typedef struct {
    int32_t nops;
    int32_t reserved;
    fr_kernel_op ops[FR_KPROG_MAX_OPS];
} fr_kprog;

typedef struct fr_ctx fr_ctx;   /* device, HIP stream, workspaces, RCCL communicator */
typedef struct fr_chol fr_chol; /* device-resident Cholesky factor (+ the training inputs it was built from) */

/* ---- context ------------------------------------------------------------------------------------- */
int fr_abi_version(void);
/* device < 0: current HIP device.  Fails with FR_NO_DEVICE when no gfx950 GPU is visible -- there is no
 * CPU fallback behind this ABI. */
int fr_ctx_create(fr_ctx** out, int device);
void fr_ctx_destroy(fr_ctx* ctx);
/* Use an existing hipStream_t (e.g. torch's current stream) instead of the context's own. */
int fr_ctx_set_stream(fr_ctx* ctx, void* hip_stream);
int fr_ctx_synchronize(fr_ctx* ctx);
const char* fr_last_error(const fr_ctx* ctx);
/* Tunables (everything else the library decides from the problem size; the tier / threshold knobs of the reservation --
 *   "reserve_rows1" / "reserve_rows2" / "reserve_rows4" (16384 / 8192 / 4096), "reserve_rows1_cu" / "reserve_rows2_cu" (12288 / 6144),
 *   "cu_reserve_min_rows" (4096) -- exist for the in-process sweeps of scripts/optset_ab.py):
 *   "nb"             outer Cholesky block: 0 (default) = chosen from the matrix size, else a multiple of 128 in [128, 4096]
 *   "nb_switch_rows" 16384 (default): with nb > 512 on one GPU, panels of 512 columns once at most this many rows remain
 *   "nb_big_rows"    22528 (default): with the automatic nb = 1024 (one GPU, N >= 18432), panels of 2048 columns while more than
 *                    this many rows remain; 0: never
 *   "lookahead"      1 (default): factor the next panel on a second stream under the trailing update
 *   "xcd_reserve"    -1 (default): while the panel chain bounds a single-GPU factorisation (from the first panel on), the trailing update keeps off
 *                    the panel stream's XCDs (1 XCD below 16384 trailing rows, 2 below 8192, 4 below 4096, nb <= 512 only: DESIGN.md
 *                    section 5); 0: never; 1..4: that many XCDs for the whole factorisation
 *   "cu_reserve"     how the reservation above is carried out.  0: whole XCDs -- the trailing update's workgroups retire on
 *                    the reserved XCDs, the panel stream's launches carry 8 / R times the workgroups and only those dealt to the
 *                    reserved XCDs work;  1 (default; above 4096 trailing rows, with the tiers at 12288 / 6144 rows): R CUs of every shader
 *                    engine of every XCD (the same number of CUs) -- the trailing
 *                    update runs as resident workgroups that claim tiles and vacate those CUs, the panel stream's launches need
 *                    no idle workgroups (which otherwise wait for a slot on the busy XCDs: DESIGN.md section 5, round 5)
 *   "xcd_reserve_big_rows" 0 (default: never): with panels wider than 512 columns, one XCD is set aside while at most this many
 *                    rows remain (measured and left off: DESIGN.md section 5, round 5)
 *   "k4_flat"        -1 (default): full 128 x 128 diagonal blocks are factored by the flat variant of the diagonal-block kernel
 *                    (one row per lane, rank-4 MFMA updates: potf2.hip) wherever that kernel has its CU to itself -- XCDs set aside,
 *                    no second stream, sharded chain -- and by the staged variant that fits beside a GEMM workgroup elsewhere; 0 / 1:
 *                    one variant throughout.  The two round differently (both within the parity tolerance): a factor is a
 *                    function of the input AND of the options that select kernels
 *   "panel_chain"    2 (default): a panel of 256 / 384 / 512 columns -- its diagonal block with the inverse blocks, and the rows below
 *                    it -- is factored by ONE resident launch (potf2.hip: panel_chain_kernel: the flat diagonal-block body on one
 *                    workgroup, 16-row slabs resident in LDS that hand over through memory, 32- / 64-row workgroups for the rows
 *                    below) instead of kb / 128 diagonal-block kernels and 2 kb / 128 - 1 launch-bound products; the look-ahead
 *                    update then follows on the panel stream.  1: only where the launch has its CUs to itself (no second stream:
 *                    small matrices, the Schur block of fr_chol_add_rows; the diagonal blocks of a sharded factorisation); 0: the
 *                    chain of launches.  A hand-off that times out (the workgroups did not get CUs side by side) raises the status
 *                    word; the factorisation is then repeated on the chain of launches (counter "panel_chain_fallbacks").  Like
 *                    "k4_flat" the option selects kernels: the factor is the same to round-off (1e-13), not to the bit
 *   "dist_schedule"  sharded (multi-GPU) factorisation, how a panel step travels: 0 = the owner solves the whole panel, one
 *                    broadcast; 1 (default) = diagonal block broadcast, rows below scattered / solved per rank / all-gathered;
 *                    2 = as 1 with the chain of diagonal blocks running ahead of the bulk rows on a second communicator
 *                     -- opt-in until it has run over RCCL on a multi-GPU node: bench.py preflights it
 *                    under the watchdog and falls back 2 -> 1 -> 0.  Every rank must use the same value.
 *   "comm_timeout_ms" 120000 (default): how long a sharded operation may wait for its collectives (a stream that does not
 *                    drain, an RCCL call that does not return) before the communicators are aborted and the call returns
 *                    FR_RCCL_ERROR; 0 = for ever.  See fr_ctx_comm_finalize.
 *   "splitk"         1 (default): products with few result tiles and a deep contraction are cut along K; 0: never
 *   "narrow_max"     16 (default): solves with at most this many right-hand sides use memory-bound kernels instead of the
 *                    128-wide GEMM tiles;  "narrow_batched_max" (-1: by size): up to this many right-hand sides the persistent
 *                    solve runs in column groups of 16
 *   "bigleaf_max"    -1 (default: 4096): solves with 2 .. this many right-hand sides against >= 4096 rows run left-looking over
 *                    2048-row blocks with explicit inverses of the 2048 x 2048 diagonal blocks (built on demand, extended after
 *                    fr_chol_add_rows); 0: never.  "bigleaf_min" (-1: by measurement): at least this many right-hand sides
 *   "leaf512"        1 (default): wide triangular solves (>= 256 right-hand sides) end in 512-row leaves against
 *                    explicit inverses of the 512 x 512 diagonal blocks, built on demand; 0: 128-row leaves only
 *   "trsv"           1 (default): solves with few right-hand sides run as one persistent launch per direction; 0: the
 *                    recursive GEMM / matrix-vector path (what a solve falls back to after a timed-out hand-off)
 *   "tri_inverse"    1 (default): the gradient terms form L^-1 and K^-1 = W^T W skipping the structural zeros
 *   "grad_shard_min" 4096 (default): with a communicator attached, fr_grad_terms of a factor with at least this many rows is split
 *                    over the ranks (row blocks of L^-1 per rank, partial K^-1 and partial reductions, one all-gather of p + 2
 *                    scalars); below, every rank computes the whole.  Every rank must use the same value.  Only handles that
 *                    came out of a collective factorisation (fr_chol_from_inputs / fr_chol_refactor with the communicator
 *                    attached) are split: a handle one rank made for itself (fr_chol_from_matrix, fr_chol_upload_l) stays local.
 *   "refine"         -1 (default): automatic -- see fr_chol_conditioning; 0: never; 1: always.  "refine_threshold": 30
 *   "predict_assoc"  0 (default): predict as the reference associates it, prior + (K^-1 K*)^T y  (mod.rs:234-241,
 *                    two n x m triangular solves);  1: prior + K*^T (K^-1 y), the same value up to rounding with two
 *                    n x 1 solves instead */
int fr_ctx_set_option(fr_ctx* ctx, const char* name, int64_t value);
/* Observability: "solve_retries" = how often an entry point of this context repeated its work on the recursive path because a
 * persistent solve gave up on a hand-off (0 in normal operation); "comm_timeouts" = how often a wait for a collective ran out; "stale_status_drops" = time-outs
 * left unread by an entry point that returned early and dropped by the next one;
 * "pool_bytes" = bytes held by the workspace pool; "panel_chain_launches" / "panel_chain_fallbacks" = resident panel launches taken /
 * factorisations repeated on the chain of launches after one of them timed out.
 * Environment read when a context is created (operators / tests; none is needed in normal use):
 *   FRIEDRICH_AMD_DIST_SCHEDULE = 0 | 1 | 2, FRIEDRICH_AMD_COMM_TIMEOUT_MS   the options of the same name without touching the host program
 *   FRIEDRICH_AMD_RCCL_PATH         the librccl to dlopen (it has to match the process's HIP runtime)
 *   FRIEDRICH_AMD_ROCTX = 1         roctx ranges around the entry points (rocprofv3 --marker-trace)
 *   FRIEDRICH_AMD_SMALL_TILES       A/B override of the 32-row-tile rule (gemm_f64.hip)
 *   FRIEDRICH_AMD_LOCAL_SYNC = 1    in-process transport: synchronise around every collective instead of ordering by events
 *   test hooks: FRIEDRICH_AMD_TEST_MAX_WORKGROUPS (cap on the grid of the persistent solves: many blocks per workgroup),
 *   FRIEDRICH_AMD_TEST_FORCE_SOLVE_TIMEOUT = 1 (every persistent solve reports a time-out: the retry path),
 *   FRIEDRICH_AMD_TEST_COMM_HANG = "schedule,rank,nth" (that rank skips its nth collective under that schedule: the watchdog). */
int fr_ctx_get_counter(fr_ctx* ctx, const char* name, int64_t* out);

/* Per-kernel-class timing with HIP events on the context's stream (bench.py's roofline leg). */
typedef enum {
    FR_PROF_GRAM = 0,      /* K1 Gram assembly kernels */
    FR_PROF_POTF2 = 1,     /* K4 diagonal-block factor + inverse */
    FR_PROF_GEMM_PANEL = 2,/* K5 panel / solve GEMMs (TRSM recast as GEMM) */
    FR_PROF_SYRK = 3,      /* K6 trailing-update SYRK (the dominant FP64-MFMA kernel) */
    FR_PROF_GEMM_SOLVE = 4,/* K5/K6 GEMMs issued by the triangular solves */
    FR_PROF_REDUCE = 5,    /* K7 epilogue reductions */
    FR_PROF_COMM = 6,      /* RCCL collectives */
    FR_PROF_SYRK_CHAIN = 7,/* K6 as launched while CUs are set aside for the panel chain: resident workgroups that claim tiles
                              (own kernel symbol syrk_lower_persist_f64_kernel, so that class 3 stays the launches of syrk_lower_f64_kernel) */
    FR_PROF_COUNT = 8
} fr_prof_class;
/* enable: 0 = off, 1 = every class, otherwise a mask with bit (class + 1) set for each class to time */
int fr_ctx_profile_enable(fr_ctx* ctx, int enable);
int fr_ctx_profile_reset(fr_ctx* ctx);
/* total milliseconds, launch count, and algorithmic flops / bytes accumulated for one class */
int fr_ctx_profile_get(fr_ctx* ctx, int prof_class, double* ms, int64_t* launches, double* flops, double* bytes);

/* ---- multi-GPU (one process per GPU; RCCL over xGMI) ---------------------------------------------- */
#define FR_COMM_ID_BYTES 128
int fr_comm_unique_id(void* out_id /* FR_COMM_ID_BYTES */);
int fr_ctx_comm_init(fr_ctx* ctx, int rank, int world_size, const void* unique_id);
/* In-process transport: the ranks are host threads of one process sharing a device (all contexts naming the same
 * group_id form one communicator).  Lets the sharded path run on a 1-GPU box; not a performance path. */
int fr_ctx_comm_init_local(fr_ctx* ctx, int group_id, int rank, int world_size);
int fr_ctx_comm_info(const fr_ctx* ctx, int* rank, int* world_size);
/* Detach the communicator (RCCL: both communicators of the context; local: leave the group): the context is single-rank
 * again and a new communicator can be attached.  abort != 0 (and always after a time-out / failed peer, when the context's
 * communicator is "lost") tears it down without waiting for the peers (ncclCommAbort) -- the recovery step of a host that
 * falls back to another schedule after FR_RCCL_ERROR: every rank finalizes, the host distributes a fresh id, every rank
 * calls fr_ctx_comm_init again.  No reference counterpart. */
int fr_ctx_comm_finalize(fr_ctx* ctx, int abort);
/* Collective self-test of the communicator the way the factorisation uses it (one broadcast from rank 0 and one
 * all-gather of small device buffers on the panel stream, results verified on the host).  Every rank calls it.
 * No reference counterpart (friedrich is single-process); FR_OK when no communicator is attached. */
int fr_ctx_comm_selftest(fr_ctx* ctx);

/* ---- src/conversion/mod.rs: the `Input` trait --------------------------------------------------------- */
/* Input::to_dmatrix / into_dmatrix (conversion/mod.rs:58-201) turn the caller's samples -- Vec<Vec<f64>> (one Vec per
 * sample, :121-146), a row-major ndarray (:149-201), a single Vec<f64> sample (:95-118), a DMatrix (:65-92) -- into a
 * column-major DMatrix by copying element by element on the host; the matrix would then be copied once more to the GPU.
 * fr_inputs_to_device does both in one pass: the samples go through a pinned bounce buffer of the context (grow-only)
 * straight to the device, and row-major data are transposed there.  *out_dev is a device-resident column-major n x d
 * matrix with leading dimension *out_ld (a multiple of 64) that every entry point of this header accepts as a data pointer;
 * release it with fr_device_free.  `data` may itself be a device pointer (FR_LAYOUT_COLMAJOR / FR_LAYOUT_ROWMAJOR). */
typedef enum {
    FR_LAYOUT_COLMAJOR = 0, /* const double*: element (r, c) at data[r + c * stride]   (DMatrix; stride >= n) */
    FR_LAYOUT_ROWMAJOR = 1, /* const double*: element (r, c) at data[r * stride + c]   (ndarray standard layout, a single Vec<f64>
                               sample with n = 1; stride >= d) */
    FR_LAYOUT_ROWPTRS = 2   /* const double* const*: data[r] points to the d values of sample r (Vec<Vec<f64>>); stride unused */
} fr_layout;
int fr_inputs_to_device(fr_ctx* ctx, int layout, const void* data, int64_t n, int64_t d, int64_t stride, double** out_dev,
                        int64_t* out_ld);
void fr_device_free(fr_ctx* ctx, double* dev);

/* ---- src/algebra/mod.rs --------------------------------------------------------------------------- */
/* make_covariance_matrix (algebra/mod.rs:41-54): out[r,c] = k(A.row(r), B.row(c)), out is n1 x n2. */
int fr_gram(fr_ctx* ctx, const fr_kprog* kernel, const double* A, int64_t n1, int64_t lda, const double* B,
            int64_t n2, int64_t ldb, int64_t d, double* out, int64_t ldo);

/* make_cholesky_cov_matrix (algebra/mod.rs:59-92) + Cholesky::new / new_with_substitute (nalgebra;
 * called at :83,:90): K = lower(k(x_i,x_j)) + noise^2 I, factored on the device.  `noise` is the
 * standard deviation (squared at :78).  has_eps/eps = cholesky_epsilon: Option<f64>.
 * capacity_hint >= n reserves room for fr_chol_add_rows (EMatrix-style growth otherwise).
 * On FR_NOT_POSITIVE_DEFINITE *out is still returned (free it) and fr_chol_info gives the column. */
int fr_chol_from_inputs(fr_ctx* ctx, const fr_kprog* kernel, const double* X, int64_t n, int64_t ldx, int64_t d,
                        double noise, int has_eps, double eps, int64_t capacity_hint, fr_chol** out);
/* Same factorisation on the inputs already resident in the handle (the optimizer's re-fit,
 * optimizer.rs:133-136, :267-270; fit_parameters mod.rs:426-429). */
int fr_chol_refactor(fr_chol* chol, const fr_kprog* kernel, double noise, int has_eps, double eps);
/* DMatrix::cholesky() / Cholesky::new_with_substitute on an explicit symmetric matrix (only the lower
 * triangle is read): multivariate_normal.rs:57. */
int fr_chol_from_matrix(fr_ctx* ctx, const double* A, int64_t n, int64_t lda, int has_eps, double eps,
                        fr_chol** out);
/* add_rows_cholesky_cov_matrix (algebra/mod.rs:97-126) + Cholesky::insert_column: append the last nb_new
 * rows of Xall (n_all x d) to the factor as one blocked bordered update.  No epsilon, no failure check
 * (plain sqrt => NaN), exactly like the reference. */
int fr_chol_add_rows(fr_chol* chol, const fr_kprog* kernel, const double* Xall, int64_t n_all, int64_t ldx,
                     int64_t d, int64_t nb_new, double noise);
/* make_gradient_covariance_matrices (algebra/mod.rs:129-155) is never materialised; see fr_grad_terms. */

/* ---- nalgebra::Cholesky methods used on covmat_cholesky -------------------------------------------- */
/* n, capacity (row capacity of the device buffers), d, number of substituted pivots, failing column (-1) */
int fr_chol_info(const fr_chol* chol, int64_t* n, int64_t* capacity, int64_t* d, int64_t* n_subst,
                 int64_t* fail_col);
/* Conditioning report of the factor as it stands (refreshed by every way a factor comes into being or changes: from_inputs,
 * refactor, from_matrix, upload_l, add_rows; sharded factors: the maximum over every rank's blocks, the same on every rank, so the
 * ranks take the same refinement decision): *max_estimate = the largest estimate max|W_ij| * max L_jj over the
 * 128 x 128 diagonal blocks (W = explicit inverse of the block), *refined = 1 when the handle applies a step of iterative
 * refinement behind every product with an inverse block (option "refine": -1 automatic, the default: on when an estimate
 * exceeds "refine_threshold", 30).  No reference counterpart: nalgebra substitutes, which needs no such step. */
int fr_chol_conditioning(const fr_chol* chol, double* max_estimate, int* refined);
/* ordered list of columns where cholesky_epsilon replaced the pivot ("pivot indices" of BASELINE.json) */
int fr_chol_substitutions(const fr_chol* chol, int64_t* idx, int64_t max_idx);
/* Cholesky::solve_mut (mod.rs:235; solve = clone + solve_mut :298,:379): B <- K^-1 B, B is n x m, in place */
int fr_chol_solve(fr_chol* chol, double* B, int64_t m, int64_t ldb);
/* Cholesky::l().solve_lower_triangular (mod.rs:203, 260-263, 342-345): B <- L^-1 B in place.
 * FR_SINGULAR_SOLVE when the factor has an exact zero on its diagonal. */
int fr_chol_solve_lower(fr_chol* chol, double* B, int64_t m, int64_t ldb);
/* Cholesky::inverse (optimizer.rs:32,169): out <- K^-1 (n x n, both triangles) */
int fr_chol_inverse(fr_chol* chol, double* out, int64_t ldo);
/* Cholesky::l() / unpack() (upper_fill = 0: zeros) or the raw serde image of the factor
 * (upper_fill = 1: NaN above the diagonal, algebra/mod.rs:67) */
int fr_chol_download_l(fr_chol* chol, double* out, int64_t ldo, int upper_fill);
/* Rebuild a device factor from a deserialised one (serde round trip, mod.rs:58): L n x n lower, X n x d. */
int fr_chol_upload_l(fr_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* X, int64_t ldx,
                     int64_t d, int64_t capacity_hint, fr_chol** out);
void fr_chol_free(fr_chol* chol);

/* ---- src/gaussian_process/mod.rs ------------------------------------------------------------------- */
/* y = residual training outputs (training_outputs.as_vector(), n values); prior_q = prior.prior(&inputs)
 * (m values, may be NULL for a zero prior); Xq is m x d. */
/* likelihood (mod.rs:196-220) */
int fr_likelihood(fr_chol* chol, const fr_kprog* kernel, const double* y, double noise, double* out);
/* Cached alpha = K^-1 y (the reference's own todo.md:10 -- "cache K^-1 y").  Hands the residual
 * training outputs (training_outputs.as_vector(), n values, host or device) to the factor; fr_predict_mean called with
 * y == NULL then returns prior + K*^T alpha, solving K alpha = y (two single-column solves) only when the factor changed
 * since the last solve (fr_chol_refactor, fr_chol_add_rows).  After fr_chol_add_rows the row count changed: hand the
 * n_all outputs over again (GaussianProcess::add_samples, mod.rs:173-190, updates both).  Values agree with the reference's
 * association (K^-1 K*)^T y to rounding. */
int fr_chol_set_targets(fr_chol* chol, const double* y);
/* predict (mod.rs:226-244).  y == NULL: use the targets / alpha cached by fr_chol_set_targets. */
int fr_predict_mean(fr_chol* chol, const fr_kprog* kernel, const double* y, const double* Xq, int64_t m,
                    int64_t ldq, const double* prior_q, double* out_mean);
/* predict_variance (mod.rs:248-273) */
int fr_predict_variance(fr_chol* chol, const fr_kprog* kernel, const double* Xq, int64_t m, int64_t ldq,
                        double* out_var);
/* predict_mean_variance (mod.rs:290-326) */
int fr_predict_mean_variance(fr_chol* chol, const fr_kprog* kernel, const double* y, const double* Xq, int64_t m,
                             int64_t ldq, const double* prior_q, double* out_mean, double* out_var);
/* predict_covariance (mod.rs:329-350): out_cov is m x m */
int fr_predict_covariance(fr_chol* chol, const fr_kprog* kernel, const double* Xq, int64_t m, int64_t ldq,
                          double* out_cov, int64_t ldc);
/* sample_at (mod.rs:371-392) + MultivariateNormal::new (multivariate_normal.rs:54-59): posterior mean,
 * covariance (may be NULL) and cholesky(cov).unpack().  FR_NOT_POSITIVE_DEFINITE mirrors the expect() at
 * multivariate_normal.rs:57. */
int fr_posterior(fr_chol* chol, const fr_kprog* kernel, const double* y, const double* Xq, int64_t m, int64_t ldq,
                 const double* prior_q, double* out_mean, double* out_cov, int64_t ldc, double* out_cov_l,
                 int64_t ldl);

/* DMatrix::gemm / gemm_tr (mod.rs:348, 383; A.4): C <- alpha * op(A) * op(B) + beta * C on the FP64 matrix cores.
 * trans_a / trans_b: 0 = as stored, 1 = transposed.  op(A) is M x K, op(B) is K x N, C is M x N.  C must not
 * alias A or B. */
int fr_gemm(fr_ctx* ctx, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
            int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc);
/* The rows of a panel against its factored diagonal block, by itself (one step of algebra/mod.rs:81-91's factorisation as the sharded
 * schedule cuts it, DESIGN.md section 6: R1 and the slice solves):  S (rows x kb, device, leading dimension lds) <- S L^-T  against a
 * factored kb x kb lower block L (device, ldl) and the explicit inverses of its 128 x 128 diagonal blocks (dinv: block s at
 * dinv + s * 128 * 128, leading dimension 128), ONE launch.  A diagnostic entry like fr_gemm: probes, tests, the cost model of the
 * sharded schedule (scripts/dist_model.py). */
int fr_panel_rows_solve(fr_ctx* ctx, double* S, int64_t lds, int64_t rows, const double* L, int64_t ldl, int64_t kb,
                        const double* dinv);

/* ---- src/parameters/kernel.rs heuristics ----------------------------------------------------------- */
/* fit_bandwidth_mean (kernel.rs:94-113): mean Euclidean distance over the n(n-1)/2 row pairs */
int fr_mean_pairwise_distance(fr_ctx* ctx, const double* X, int64_t n, int64_t ldx, int64_t d, double* out);

/* ---- src/parameters/prior.rs ------------------------------------------------------------------------ */
/* LinearPrior::fit (prior.rs:139-159): least squares of [1 | X] w = y, where the reference runs nalgebra's SVD solve on
 * the host.  Tall-skinny Householder QR on the device + the SVD of the (d + 1) x (d + 1) triangle on the host: the same
 * singular-value solve (eps = 0) without forming the normal equations and without moving X.  out_weights: d values,
 * *out_intercept: the constant term.  d <= 62 on the device (FR_UNSUPPORTED_KERNEL beyond: keep the nalgebra path). */
int fr_linear_prior_fit(fr_ctx* ctx, const double* X, int64_t n, int64_t ldx, int64_t d, const double* y,
                        double* out_weights, double* out_intercept);

/* ---- src/gaussian_process/optimizer.rs ------------------------------------------------------------- */
/* The per-iteration reductions of gradient_marginal_likelihood (:24-60, scaled = 0) and
 * scaled_gradient_marginal_likelihood (:159-203, scaled = 1) without materialising K^-1 G_q products on
 * the host: out_grad has nb_parameters (+1 when !scaled: the noise gradient, :54-57) entries,
 * *out_scale = y^T K^-1 y / n (:174, scaled only). */
int fr_grad_terms(fr_chol* chol, const fr_kprog* kernel, const double* y, double noise, int scaled,
                  double* out_grad, double* out_scale);

#ifdef __cplusplus
}
#endif
#endif /* FRIEDRICH_AMD_H */

//! Raw bindings to `libfriedrich_amd.so` (include/friedrich_amd.h, FR_ABI_VERSION 2).
//!
//! One declaration per export of the header, same order, same argument order; `tests/test_rust_shim.py` parses both files
//! and fails when they drift.  Nothing here is safe to call directly: `super::device` wraps what friedrich needs.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct fr_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct fr_chol {
    _private: [u8; 0],
}

pub const FR_ABI_VERSION: c_int = 2;

// fr_status
pub const FR_OK: c_int = 0;
pub const FR_NOT_POSITIVE_DEFINITE: c_int = 1;
pub const FR_SINGULAR_SOLVE: c_int = 2;
pub const FR_UNSUPPORTED_KERNEL: c_int = 3;
pub const FR_SHAPE: c_int = 4;
pub const FR_INVALID_ARGUMENT: c_int = 5;
pub const FR_OUT_OF_MEMORY: c_int = 6;
pub const FR_HIP_ERROR: c_int = 7;
pub const FR_RCCL_ERROR: c_int = 8;
pub const FR_NO_DEVICE: c_int = 9;

// fr_kernel_kind
pub const FR_K_LINEAR: i32 = 0;
pub const FR_K_POLYNOMIAL: i32 = 1;
pub const FR_K_SQUAREDEXP: i32 = 2;
pub const FR_K_EXPONENTIAL: i32 = 3;
pub const FR_K_MATERN1: i32 = 4;
pub const FR_K_MATERN2: i32 = 5;
pub const FR_K_HYPERTAN: i32 = 6;
pub const FR_K_MULTIQUADRIC: i32 = 7;
pub const FR_K_RATIONALQUADRATIC: i32 = 8;
pub const FR_K_SUM: i32 = 100;
pub const FR_K_PROD: i32 = 101;
pub const FR_KPROG_MAX_OPS: usize = 15;

// fr_layout
pub const FR_LAYOUT_COLMAJOR: c_int = 0;
pub const FR_LAYOUT_ROWMAJOR: c_int = 1;
pub const FR_LAYOUT_ROWPTRS: c_int = 2;

// fr_prof_class
pub const FR_PROF_GRAM: c_int = 0;
pub const FR_PROF_POTF2: c_int = 1;
pub const FR_PROF_GEMM_PANEL: c_int = 2;
pub const FR_PROF_SYRK: c_int = 3;
pub const FR_PROF_GEMM_SOLVE: c_int = 4;
pub const FR_PROF_REDUCE: c_int = 5;
pub const FR_PROF_COMM: c_int = 6;
pub const FR_PROF_SYRK_CHAIN: c_int = 7;

pub const FR_COMM_ID_BYTES: usize = 128;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct fr_kernel_op {
    pub kind: i32,
    pub nparams: i32,
    pub params: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct fr_kprog {
    pub nops: i32,
    pub reserved: i32,
    pub ops: [fr_kernel_op; FR_KPROG_MAX_OPS],
}

extern "C" {
    // ---- context
    pub fn fr_abi_version() -> c_int;
    pub fn fr_ctx_create(out: *mut *mut fr_ctx, device: c_int) -> c_int;
    pub fn fr_ctx_destroy(ctx: *mut fr_ctx);
    pub fn fr_ctx_set_stream(ctx: *mut fr_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn fr_ctx_synchronize(ctx: *mut fr_ctx) -> c_int;
    pub fn fr_last_error(ctx: *const fr_ctx) -> *const c_char;
    pub fn fr_ctx_set_option(ctx: *mut fr_ctx, name: *const c_char, value: i64) -> c_int;
    pub fn fr_ctx_get_counter(ctx: *mut fr_ctx, name: *const c_char, out: *mut i64) -> c_int;
    pub fn fr_ctx_profile_enable(ctx: *mut fr_ctx, enable: c_int) -> c_int;
    pub fn fr_ctx_profile_reset(ctx: *mut fr_ctx) -> c_int;
    pub fn fr_ctx_profile_get(ctx: *mut fr_ctx, prof_class: c_int, ms: *mut f64, launches: *mut i64, flops: *mut f64,
                              bytes: *mut f64) -> c_int;
    // ---- multi-GPU
    pub fn fr_comm_unique_id(out_id: *mut c_void) -> c_int;
    pub fn fr_ctx_comm_init(ctx: *mut fr_ctx, rank: c_int, world_size: c_int, unique_id: *const c_void) -> c_int;
    pub fn fr_ctx_comm_init_local(ctx: *mut fr_ctx, group_id: c_int, rank: c_int, world_size: c_int) -> c_int;
    pub fn fr_ctx_comm_info(ctx: *const fr_ctx, rank: *mut c_int, world_size: *mut c_int) -> c_int;
    pub fn fr_ctx_comm_finalize(ctx: *mut fr_ctx, abort: c_int) -> c_int;
    pub fn fr_ctx_comm_selftest(ctx: *mut fr_ctx) -> c_int;
    // ---- src/conversion/mod.rs
    pub fn fr_inputs_to_device(ctx: *mut fr_ctx, layout: c_int, data: *const c_void, n: i64, d: i64, stride: i64,
                               out_dev: *mut *mut f64, out_ld: *mut i64) -> c_int;
    pub fn fr_device_free(ctx: *mut fr_ctx, dev: *mut f64);
    // ---- src/algebra/mod.rs
    pub fn fr_gram(ctx: *mut fr_ctx, kernel: *const fr_kprog, a: *const f64, n1: i64, lda: i64, b: *const f64, n2: i64,
                   ldb: i64, d: i64, out: *mut f64, ldo: i64) -> c_int;
    pub fn fr_chol_from_inputs(ctx: *mut fr_ctx, kernel: *const fr_kprog, x: *const f64, n: i64, ldx: i64, d: i64,
                               noise: f64, has_eps: c_int, eps: f64, capacity_hint: i64, out: *mut *mut fr_chol) -> c_int;
    pub fn fr_chol_refactor(chol: *mut fr_chol, kernel: *const fr_kprog, noise: f64, has_eps: c_int, eps: f64) -> c_int;
    pub fn fr_chol_from_matrix(ctx: *mut fr_ctx, a: *const f64, n: i64, lda: i64, has_eps: c_int, eps: f64,
                               out: *mut *mut fr_chol) -> c_int;
    pub fn fr_chol_add_rows(chol: *mut fr_chol, kernel: *const fr_kprog, xall: *const f64, n_all: i64, ldx: i64, d: i64,
                            nb_new: i64, noise: f64) -> c_int;
    // ---- nalgebra::Cholesky methods used on covmat_cholesky
    pub fn fr_chol_info(chol: *const fr_chol, n: *mut i64, capacity: *mut i64, d: *mut i64, n_subst: *mut i64,
                        fail_col: *mut i64) -> c_int;
    pub fn fr_chol_conditioning(chol: *const fr_chol, max_estimate: *mut f64, refined: *mut c_int) -> c_int;
    pub fn fr_chol_substitutions(chol: *const fr_chol, idx: *mut i64, max_idx: i64) -> c_int;
    pub fn fr_chol_solve(chol: *mut fr_chol, b: *mut f64, m: i64, ldb: i64) -> c_int;
    pub fn fr_chol_solve_lower(chol: *mut fr_chol, b: *mut f64, m: i64, ldb: i64) -> c_int;
    pub fn fr_chol_inverse(chol: *mut fr_chol, out: *mut f64, ldo: i64) -> c_int;
    pub fn fr_chol_download_l(chol: *mut fr_chol, out: *mut f64, ldo: i64, upper_fill: c_int) -> c_int;
    pub fn fr_chol_upload_l(ctx: *mut fr_ctx, l: *const f64, n: i64, ldl: i64, x: *const f64, ldx: i64, d: i64,
                            capacity_hint: i64, out: *mut *mut fr_chol) -> c_int;
    pub fn fr_chol_free(chol: *mut fr_chol);
    // ---- src/gaussian_process/mod.rs
    pub fn fr_likelihood(chol: *mut fr_chol, kernel: *const fr_kprog, y: *const f64, noise: f64, out: *mut f64) -> c_int;
    pub fn fr_chol_set_targets(chol: *mut fr_chol, y: *const f64) -> c_int;
    pub fn fr_predict_mean(chol: *mut fr_chol, kernel: *const fr_kprog, y: *const f64, xq: *const f64, m: i64, ldq: i64,
                           prior_q: *const f64, out_mean: *mut f64) -> c_int;
    pub fn fr_predict_variance(chol: *mut fr_chol, kernel: *const fr_kprog, xq: *const f64, m: i64, ldq: i64,
                               out_var: *mut f64) -> c_int;
    pub fn fr_predict_mean_variance(chol: *mut fr_chol, kernel: *const fr_kprog, y: *const f64, xq: *const f64, m: i64,
                                    ldq: i64, prior_q: *const f64, out_mean: *mut f64, out_var: *mut f64) -> c_int;
    pub fn fr_predict_covariance(chol: *mut fr_chol, kernel: *const fr_kprog, xq: *const f64, m: i64, ldq: i64,
                                 out_cov: *mut f64, ldc: i64) -> c_int;
    pub fn fr_posterior(chol: *mut fr_chol, kernel: *const fr_kprog, y: *const f64, xq: *const f64, m: i64, ldq: i64,
                        prior_q: *const f64, out_mean: *mut f64, out_cov: *mut f64, ldc: i64, out_cov_l: *mut f64,
                        ldl: i64) -> c_int;
    pub fn fr_gemm(ctx: *mut fr_ctx, trans_a: c_int, trans_b: c_int, m: i64, n: i64, k: i64, alpha: f64, a: *const f64,
                   lda: i64, b: *const f64, ldb: i64, beta: f64, c: *mut f64, ldc: i64) -> c_int;
    pub fn fr_panel_rows_solve(ctx: *mut fr_ctx, s: *mut f64, lds: i64, rows: i64, l: *const f64, ldl: i64, kb: i64,
                               dinv: *const f64) -> c_int;
    // ---- src/parameters/kernel.rs heuristics
    pub fn fr_mean_pairwise_distance(ctx: *mut fr_ctx, x: *const f64, n: i64, ldx: i64, d: i64, out: *mut f64) -> c_int;
    // ---- src/parameters/prior.rs
    pub fn fr_linear_prior_fit(ctx: *mut fr_ctx, x: *const f64, n: i64, ldx: i64, d: i64, y: *const f64,
                               out_weights: *mut f64, out_intercept: *mut f64) -> c_int;
    // ---- src/gaussian_process/optimizer.rs
    pub fn fr_grad_terms(chol: *mut fr_chol, kernel: *const fr_kprog, y: *const f64, noise: f64, scaled: c_int,
                         out_grad: *mut f64, out_scale: *mut f64) -> c_int;
}

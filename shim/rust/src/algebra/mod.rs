//! Replacement for friedrich's `src/algebra/mod.rs` (lines 41-155 of 0.5.1) with the `friedrich_mi355x` feature.
//!
//! Same four functions, same signatures apart from the factor type (`CholeskyHandle` instead of
//! `Cholesky<f64, Dynamic>`), same panic texts.  A kernel without a device program (user-defined `Kernel`), a machine
//! without a gfx950 GPU, or a model below `DEVICE_MIN_ROWS` rows takes the original nalgebra bodies: friedrich 0.5.1's own
//! `src/algebra/mod.rs` is kept, UNCHANGED, as `src/algebra/host.rs` (`git mv`; only its two `mod` / `pub use` lines for
//! `extendable_matrix` move here).

mod extendable_matrix;
pub use extendable_matrix::{EMatrix, EVector};
pub mod device;
pub mod ffi;
mod host;

use self::device::{check, context, program, raw, CholeskyHandle, DEVICE_MIN_ROWS};
use self::ffi::*;
use crate::parameters::kernel::Kernel;
use nalgebra::{storage::Storage, DMatrix, Dynamic, Matrix, SliceStorage, U1};
use std::os::raw::c_int;
use std::ptr;

pub type SMatrix<S> = Matrix<f64, Dynamic, Dynamic, S>;
pub type SRowVector<S> = Matrix<f64, U1, Dynamic, S>;
pub type SVector<S> = Matrix<f64, Dynamic, U1, S>;
pub type MatrixSlice<'a> = Matrix<f64, Dynamic, Dynamic, SliceStorage<'a, f64, Dynamic, Dynamic, U1, Dynamic>>;
pub type VectorSlice<'a> = Matrix<f64, Dynamic, U1, SliceStorage<'a, f64, Dynamic, U1, U1, Dynamic>>;

//-----------------------------------------------------------------------------
// COVARIANCE MATRIX

/// make_covariance_matrix (algebra/mod.rs:41-54) -> fr_gram
pub fn make_covariance_matrix<S1: Storage<f64, Dynamic, Dynamic>, S2: Storage<f64, Dynamic, Dynamic>, K: Kernel>(
    m1: &SMatrix<S1>,
    m2: &SMatrix<S2>,
    kernel: &K)
    -> DMatrix<f64>
{
    if let (Some(ctx), Some(prog)) = (context(), program(kernel))
    {
        if m1.nrows() * m2.nrows() >= DEVICE_MIN_ROWS * DEVICE_MIN_ROWS
        {
            let mut out = DMatrix::<f64>::zeros(m1.nrows(), m2.nrows());
            let ((a, lda), (b, ldb)) = (raw(m1), raw(m2));
            let st = unsafe {
                fr_gram(ctx, &prog, a, m1.nrows() as i64, lda, b, m2.nrows() as i64, ldb, m1.ncols() as i64,
                        out.as_mut_ptr(), m1.nrows() as i64)
            };
            check(ctx, st);
            return out;
        }
    }
    host::make_covariance_matrix(m1, m2, kernel)
}

/// make_cholesky_cov_matrix (algebra/mod.rs:59-92) -> fr_chol_from_inputs.  `capacity` = row capacity of the `EMatrix` the inputs
/// live in, so that later `add_samples` grow in place.
pub fn make_cholesky_cov_matrix<S: Storage<f64, Dynamic, Dynamic>, K: Kernel>(inputs: &SMatrix<S>,
                                                                              kernel: &K,
                                                                              diagonal_noise: f64,
                                                                              cholesky_epsilon: Option<f64>)
                                                                              -> CholeskyHandle
{
    if let (Some(ctx), Some(prog)) = (context(), program(kernel))
    {
        if inputs.nrows() >= DEVICE_MIN_ROWS
        {
            let (x, ldx) = raw(inputs);
            let mut h = ptr::null_mut();
            let st = unsafe {
                fr_chol_from_inputs(ctx, &prog, x, inputs.nrows() as i64, ldx, inputs.ncols() as i64, diagonal_noise,
                                    cholesky_epsilon.is_some() as c_int, cholesky_epsilon.unwrap_or(0.), ldx, &mut h)
            };
            return match (st, cholesky_epsilon)
            {
                (FR_OK, _) => CholeskyHandle::Device(h),
                (FR_NOT_POSITIVE_DEFINITE, Some(cholesky_epsilon)) =>
                {
                    unsafe { fr_chol_free(h) };
                    panic!("Cholesky decomposition failed even though we used `cholesky_epsilon` value of {cholesky_epsilon}")
                }
                (FR_NOT_POSITIVE_DEFINITE, None) =>
                {
                    unsafe { fr_chol_free(h) };
                    panic!("Cholesky decomposition failed, consider setting `cholesky_epsilon` via `GaussianProcessBuilder`")
                }
                _ => panic!("friedrich_amd (status {}): {}", st, device::last_error(ctx))
            };
        }
    }
    CholeskyHandle::Host(host::make_cholesky_cov_matrix(inputs, kernel, diagonal_noise, cholesky_epsilon))
}

/// add_rows_cholesky_cov_matrix (algebra/mod.rs:97-126) -> fr_chol_add_rows: one blocked bordered update instead of
/// `nb_new_inputs` calls of `insert_column`; no epsilon and no failure check, exactly like the reference (plain sqrt => NaN).
pub fn add_rows_cholesky_cov_matrix<S: Storage<f64, Dynamic, Dynamic>, K: Kernel>(covmat_cholesky: &mut CholeskyHandle,
                                                                                  all_inputs: &SMatrix<S>,
                                                                                  nb_new_inputs: usize,
                                                                                  kernel: &K,
                                                                                  diagonal_noise: f64)
{
    // A host model that has grown past the cross-over moves to the device once (no refactorisation).
    if let CholeskyHandle::Host(host) = covmat_cholesky
    {
        let nb_old_inputs = all_inputs.nrows() - nb_new_inputs;
        if all_inputs.nrows() >= DEVICE_MIN_ROWS && program(kernel).is_some()
        {
            if let Some(dev) = CholeskyHandle::to_device(host, &all_inputs.rows(0, nb_old_inputs))
            {
                *covmat_cholesky = dev;
            }
        }
    }
    match covmat_cholesky
    {
        CholeskyHandle::Device(h) =>
        {
            let ctx = context().expect("friedrich_amd: device factor without a device");
            let prog = program(kernel).expect("friedrich_amd: the kernel of a device model lost its device program");
            let (x, ldx) = raw(all_inputs);
            let st = unsafe {
                fr_chol_add_rows(*h, &prog, x, all_inputs.nrows() as i64, ldx, all_inputs.ncols() as i64, nb_new_inputs as i64,
                                 diagonal_noise)
            };
            check(ctx, st);
        }
        CholeskyHandle::Host(host) => host::add_rows_cholesky_cov_matrix(host, all_inputs, nb_new_inputs, kernel, diagonal_noise)
    }
}

/// make_gradient_covariance_matrices (algebra/mod.rs:129-155): only the host optimizer path still materialises the p n x n
/// matrices; the device path reduces them on the fly (`fr_grad_terms`, see gaussian_process/optimizer.rs).
pub use self::host::make_gradient_covariance_matrices;

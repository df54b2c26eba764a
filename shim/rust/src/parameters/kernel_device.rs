//! Additions to friedrich's `src/parameters/kernel.rs` with the `friedrich_mi355x` feature.
//!
//! 1. ONE new defaulted method inside `pub trait Kernel` (kernel.rs:23-82) -- every existing implementation, including user
//!    kernels outside the crate, keeps compiling and keeps the nalgebra path:
//!
//! ```ignore
//!     /// Appends this kernel's reverse-polish device program (friedrich_amd.h `fr_kprog`: leaves push k(x, y) with their
//!     /// parameters in `get_parameters()` order, `KernelSum` / `KernelProd` pop two); `false` = not expressible on the device.
//!     fn device_program(&self, _prog: &mut crate::algebra::ffi::fr_kprog) -> bool
//!     {
//!         false
//!     }
//! ```
//!
//! 2. The overrides below, one per built-in kernel, pasted into the matching `impl Kernel for ...` block (file:line of the block in
//!    0.5.1 given with each).  Parameter order = `get_parameters()` of that kernel (kernel.rs:393, 474, 583, 688, 795, 907, 991,
//!    1061, 1147), which is also the order `fr_grad_terms` returns the gradient in.
//!
//! 3. `fit_bandwidth_mean` (kernel.rs:94-113) on the device, at the bottom.
use crate::algebra::device::{check, context, push_leaf, push_op, raw, DEVICE_MIN_ROWS};
use crate::algebra::ffi::*;
use crate::algebra::SMatrix;
use nalgebra::{storage::Storage, Dynamic};

// impl<T: Kernel, U: Kernel> Kernel for KernelSum<T, U>        (kernel.rs:141-211)
//     fn device_program(&self, p: &mut fr_kprog) -> bool { self.k1.device_program(p) && self.k2.device_program(p) && push_op(p, FR_K_SUM) }
// impl<T: Kernel, U: Kernel> Kernel for KernelProd<T, U>       (kernel.rs:230-307)
//     fn device_program(&self, p: &mut fr_kprog) -> bool { self.k1.device_program(p) && self.k2.device_program(p) && push_op(p, FR_K_PROD) }
// impl<K: Kernel> Kernel for KernelArith<K>                    (kernel.rs:312-338, forwards everything to self.0)
//     fn device_program(&self, p: &mut fr_kprog) -> bool { self.0.device_program(p) }

/// The bodies of the nine leaf overrides, written as free functions over the kernels' public fields so that this file
/// type-checks on its own; inside `kernel.rs` each becomes `fn device_program(&self, p: &mut fr_kprog) -> bool { <body> }`.
pub mod leaves
{
    use super::*;
    use crate::parameters::kernel::*;

    /// impl Kernel for Linear (kernel.rs:366-402): [c]
    pub fn linear(k: &Linear, p: &mut fr_kprog) -> bool
    {
        push_leaf(p, FR_K_LINEAR, &[k.c])
    }
    /// impl Kernel for Polynomial (kernel.rs:441-485): [alpha, c, d]
    pub fn polynomial(k: &Polynomial, p: &mut fr_kprog) -> bool
    {
        push_leaf(p, FR_K_POLYNOMIAL, &[k.alpha, k.c, k.d])
    }
    /// impl Kernel for SquaredExp (kernel.rs:534-601; `Gaussian` is an alias): [ls, ampl]
    pub fn squared_exp(k: &SquaredExp, p: &mut fr_kprog) -> bool
    {
        push_leaf(p, FR_K_SQUAREDEXP, &[k.ls, k.ampl])
    }
    /// impl Kernel for Exponential (kernel.rs:639-706): [ls, ampl]
    pub fn exponential(k: &Exponential, p: &mut fr_kprog) -> bool
    {
        push_leaf(p, FR_K_EXPONENTIAL, &[k.ls, k.ampl])
    }
    /// impl Kernel for Matern1 (kernel.rs:744-813): [ls, ampl]
    pub fn matern1(k: &Matern1, p: &mut fr_kprog) -> bool
    {
        push_leaf(p, FR_K_MATERN1, &[k.ls, k.ampl])
    }
    /// impl Kernel for Matern2 (kernel.rs:851-925): [ls, ampl]
    pub fn matern2(k: &Matern2, p: &mut fr_kprog) -> bool
    {
        push_leaf(p, FR_K_MATERN2, &[k.ls, k.ampl])
    }
    /// impl Kernel for HyperTan (kernel.rs:961-1001): [alpha, c]
    pub fn hyper_tan(k: &HyperTan, p: &mut fr_kprog) -> bool
    {
        push_leaf(p, FR_K_HYPERTAN, &[k.alpha, k.c])
    }
    /// impl Kernel for Multiquadric (kernel.rs:1034-1070): [c]
    pub fn multiquadric(k: &Multiquadric, p: &mut fr_kprog) -> bool
    {
        push_leaf(p, FR_K_MULTIQUADRIC, &[k.c])
    }
    /// impl Kernel for RationalQuadratic (kernel.rs:1106-1157): [alpha, ls]
    pub fn rational_quadratic(k: &RationalQuadratic, p: &mut fr_kprog) -> bool
    {
        push_leaf(p, FR_K_RATIONALQUADRATIC, &[k.alpha, k.ls])
    }
    /// KernelSum / KernelProd: k1, k2, then the combinator
    pub fn combine(left: bool, right: bool, kind: i32, p: &mut fr_kprog) -> bool
    {
        left && right && push_op(p, kind)
    }
}

/// fit_bandwidth_mean (kernel.rs:94-113): mean Euclidean distance over the n (n - 1) / 2 row pairs -> fr_mean_pairwise_distance.
/// Replaces the body; the host loop stays for small inputs / no device.
pub fn fit_bandwidth_mean<S: Storage<f64, Dynamic, Dynamic>>(training_inputs: &SMatrix<S>) -> f64
{
    if let Some(ctx) = context()
    {
        if training_inputs.nrows() >= DEVICE_MIN_ROWS
        {
            let (x, ldx) = raw(training_inputs);
            let mut mean = 0f64;
            let st = unsafe {
                fr_mean_pairwise_distance(ctx, x, training_inputs.nrows() as i64, ldx, training_inputs.ncols() as i64, &mut mean)
            };
            check(ctx, st);
            return mean;
        }
    }
    // kernel.rs:97-112
    let mut sum_distances = 0.;
    for (sample_index, sample) in training_inputs.row_iter().enumerate()
    {
        for sample2 in training_inputs.row_iter().skip(sample_index + 1)
        {
            sum_distances += (sample - sample2).norm();
        }
    }
    let nb_samples = training_inputs.nrows();
    sum_distances / (((nb_samples * nb_samples - nb_samples) / 2) as f64)
}

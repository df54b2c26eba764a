//! Replacement body for `LinearPrior::fit` (src/parameters/prior.rs:139-159) with the `friedrich_mi355x` feature.
//!
//! The reference clones the n x d inputs, inserts a column of ones and runs nalgebra's SVD solve on the host.  The device path
//! factors [1 | X | y] with a tall-skinny Householder QR where the data are and takes the SVD of the (d + 1) x (d + 1) triangle --
//! the same singular-value solve (eps = 0) -- so a large-N model never bounces to a host SVD.  d <= 62 on the device.
use crate::algebra::device::{check, context, raw, DEVICE_MIN_ROWS};
use crate::algebra::ffi::*;
use crate::algebra::{SMatrix, SVector};
use nalgebra::{storage::Storage, DVector, Dynamic, U1};

/// Returns (intercept, weights); inside `impl Prior for LinearPrior` the caller stores them:
/// `let (intercept, weights) = fit_linear(training_inputs, training_outputs); self.intercept = intercept; self.weights = weights;`
pub fn fit_linear<SM: Storage<f64, Dynamic, Dynamic> + Clone, SV: Storage<f64, Dynamic, U1>>(training_inputs: &SMatrix<SM>,
                                                                                             training_outputs: &SVector<SV>)
                                                                                             -> (f64, DVector<f64>)
{
    let (n, d) = (training_inputs.nrows(), training_inputs.ncols());
    if let Some(ctx) = context()
    {
        if n >= DEVICE_MIN_ROWS && d <= 62
        {
            let (x, ldx) = raw(training_inputs);
            let (y, _) = raw(training_outputs);
            let mut weights = DVector::<f64>::zeros(d);
            let mut intercept = 0f64;
            let st = unsafe { fr_linear_prior_fit(ctx, x, n as i64, ldx, d as i64, y, weights.as_mut_ptr(), &mut intercept) };
            if st != FR_UNSUPPORTED_KERNEL
            {
                check(ctx, st);
                return (intercept, weights);
            }
        }
    }
    // prior.rs:144-158
    let weights = training_inputs.clone()
                                 .insert_column(0, 1.)
                                 .svd(true, true)
                                 .solve(training_outputs, 0.)
                                 .expect("Linear prior fit : solve failed.");
    (weights[0], weights.remove_row(0))
}

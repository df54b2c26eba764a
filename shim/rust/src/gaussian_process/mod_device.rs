//! Replacement bodies for `impl GaussianProcess` (src/gaussian_process/mod.rs:142-445 of friedrich 0.5.1) with the
//! `friedrich_mi355x` feature.  The struct changes in ONE place:
//!
//! ```ignore
//!     covmat_cholesky: crate::algebra::device::CholeskyHandle     // was: Cholesky<f64, Dynamic>   (mod.rs:78)
//! ```
//!
//! Every method below has the reference's signature, asserts and panic texts; a model whose factor is `CholeskyHandle::Host`
//! (user-defined kernel, no GPU, fewer than `DEVICE_MIN_ROWS` rows) runs the original body, which stays in the file under the
//! name given in each `Host(..) =>` arm (`host_predict`, ...: the 0.5.1 bodies, renamed, untouched).
use super::multivariate_normal::MultivariateNormal;
use super::GaussianProcess;
use crate::algebra::device::{check, context, program, raw, CholeskyHandle};
use crate::algebra::ffi::*;
use crate::algebra::{add_rows_cholesky_cov_matrix, make_cholesky_cov_matrix, EMatrix, EVector};
use crate::conversion::Input;
use crate::parameters::{kernel::Kernel, prior::Prior};
use chrono::Duration;
use nalgebra::{DMatrix, DVector};

impl<KernelType: Kernel, PriorType: Prior> GaussianProcess<KernelType, PriorType>
{
    /// The device factor, its context and the kernel's program, or `None` for a host model.
    fn device(&self) -> Option<(*mut fr_ctx, *mut fr_chol, fr_kprog)>
    {
        let h = self.covmat_cholesky.device()?;
        Some((context()?, h, program(&self.kernel)?))
    }

    /// Hands the residual training outputs to the factor so that `predict` can use the cached alpha = K^-1 y (the reference's own
    /// todo.md:10).  Called wherever the outputs or the factor change: `new`, `add_samples`, `fit_parameters`.
    fn refresh_targets(&self)
    {
        if let Some((ctx, h, _)) = self.device()
        {
            let (y, _) = raw(&self.training_outputs.as_vector());
            check(ctx, unsafe { fr_chol_set_targets(h, y) });
        }
    }

    // mod.rs:142-167
    pub fn new<T: Input>(prior: PriorType,
                         kernel: KernelType,
                         noise: f64,
                         cholesky_epsilon: Option<f64>,
                         training_inputs: T,
                         training_outputs: T::InVector)
                         -> Self
    {
        assert!(noise >= 0., "The noise parameter should non-negative but we tried to set it to {}", noise);
        let training_inputs = T::into_dmatrix(training_inputs);
        let training_outputs = T::into_dvector(training_outputs);
        assert_eq!(training_inputs.nrows(), training_outputs.nrows());
        let training_inputs = EMatrix::new(training_inputs);
        let training_outputs = EVector::new(training_outputs - prior.prior(&training_inputs.as_matrix()));
        // (the device keeps its own copy of the inputs inside the factor handle: predict* never sends them again)
        let covmat_cholesky = make_cholesky_cov_matrix(&training_inputs.as_matrix(), &kernel, noise, cholesky_epsilon);
        let gp = GaussianProcess { prior, kernel, noise, cholesky_epsilon, training_inputs, training_outputs, covmat_cholesky };
        gp.refresh_targets();
        gp
    }

    // mod.rs:173-190
    pub fn add_samples<T: Input>(&mut self, inputs: &T, outputs: &T::InVector)
    {
        let inputs = T::to_dmatrix(inputs);
        let outputs = T::to_dvector(outputs);
        assert_eq!(inputs.nrows(), outputs.nrows());
        assert_eq!(inputs.ncols(), self.training_inputs.as_matrix().ncols());
        let outputs = outputs - self.prior.prior(&inputs);
        self.training_inputs.add_rows(&inputs);
        self.training_outputs.add_rows(&outputs);
        let nb_new_inputs = inputs.nrows();
        add_rows_cholesky_cov_matrix(&mut self.covmat_cholesky,
                                     &self.training_inputs.as_matrix(),
                                     nb_new_inputs,
                                     &self.kernel,
                                     self.noise);
        self.refresh_targets();
    }

    // mod.rs:196-220
    pub fn likelihood(&self) -> f64
    {
        match self.device()
        {
            Some((ctx, h, prog)) =>
            {
                let (y, _) = raw(&self.training_outputs.as_vector());
                let mut out = 0f64;
                match unsafe { fr_likelihood(h, &prog, y, self.noise, &mut out) }
                {
                    FR_SINGULAR_SOLVE => panic!("likelihood : solve failed"),
                    st => check(ctx, st)
                }
                out
            }
            None => self.host_likelihood()
        }
    }

    // mod.rs:226-244
    pub fn predict<T: Input>(&self, inputs: &T) -> T::OutVector
    {
        let inputs = T::to_dmatrix(inputs);
        assert_eq!(inputs.ncols(), self.training_inputs.as_matrix().ncols());
        match self.device()
        {
            Some((ctx, h, prog)) =>
            {
                let prior = self.prior.prior(&inputs);
                let mut mean = DVector::<f64>::zeros(inputs.nrows());
                let (xq, ldq) = raw(&inputs);
                // y = null: prior + K*^T alpha with the cached alpha (values agree with (K^-1 K*)^T y to rounding)
                let st = unsafe {
                    fr_predict_mean(h, &prog, std::ptr::null(), xq, inputs.nrows() as i64, ldq, prior.as_ptr(), mean.as_mut_ptr())
                };
                check(ctx, st);
                T::from_dvector(&mean)
            }
            None => T::from_dvector(&self.host_predict(&inputs))
        }
    }

    // mod.rs:248-273
    pub fn predict_variance<T: Input>(&self, inputs: &T) -> T::OutVector
    {
        let inputs = T::to_dmatrix(inputs);
        assert_eq!(inputs.ncols(), self.training_inputs.as_matrix().ncols());
        match self.device()
        {
            Some((ctx, h, prog)) =>
            {
                let mut variances = DVector::<f64>::zeros(inputs.nrows());
                let (xq, ldq) = raw(&inputs);
                match unsafe { fr_predict_variance(h, &prog, xq, inputs.nrows() as i64, ldq, variances.as_mut_ptr()) }
                {
                    FR_SINGULAR_SOLVE => panic!("predict_covariance : solve failed"),
                    st => check(ctx, st)
                }
                T::from_dvector(&variances)
            }
            None => T::from_dvector(&self.host_predict_variance(&inputs))
        }
    }

    // mod.rs:290-326
    pub fn predict_mean_variance<T: Input>(&self, inputs: &T) -> (T::OutVector, T::OutVector)
    {
        let inputs = T::to_dmatrix(inputs);
        assert_eq!(inputs.ncols(), self.training_inputs.as_matrix().ncols());
        match self.device()
        {
            Some((ctx, h, prog)) =>
            {
                let prior = self.prior.prior(&inputs);
                let (y, _) = raw(&self.training_outputs.as_vector());
                let mut mean = DVector::<f64>::zeros(inputs.nrows());
                let mut variances = DVector::<f64>::zeros(inputs.nrows());
                let (xq, ldq) = raw(&inputs);
                let st = unsafe {
                    fr_predict_mean_variance(h, &prog, y, xq, inputs.nrows() as i64, ldq, prior.as_ptr(), mean.as_mut_ptr(),
                                             variances.as_mut_ptr())
                };
                check(ctx, st);
                (T::from_dvector(&mean), T::from_dvector(&variances))
            }
            None =>
            {
                let (mean, variances) = self.host_predict_mean_variance(&inputs);
                (T::from_dvector(&mean), T::from_dvector(&variances))
            }
        }
    }

    // mod.rs:329-350
    pub fn predict_covariance<T: Input>(&self, inputs: &T) -> DMatrix<f64>
    {
        let inputs = T::to_dmatrix(inputs);
        assert_eq!(inputs.ncols(), self.training_inputs.as_matrix().ncols());
        match self.device()
        {
            Some((ctx, h, prog)) =>
            {
                let m = inputs.nrows();
                let mut cov = DMatrix::<f64>::zeros(m, m);
                let (xq, ldq) = raw(&inputs);
                match unsafe { fr_predict_covariance(h, &prog, xq, m as i64, ldq, cov.as_mut_ptr(), m as i64) }
                {
                    FR_SINGULAR_SOLVE => panic!("predict_covariance : solve failed"),
                    st => check(ctx, st)
                }
                cov
            }
            None => self.host_predict_covariance(&inputs)
        }
    }

    // mod.rs:371-392 + MultivariateNormal::new (multivariate_normal.rs:54-59)
    pub fn sample_at<T: Input>(&self, inputs: &T) -> MultivariateNormal<T>
    {
        let inputs = T::to_dmatrix(inputs);
        assert_eq!(inputs.ncols(), self.training_inputs.as_matrix().ncols());
        match self.device()
        {
            Some((ctx, h, prog)) =>
            {
                let m = inputs.nrows();
                let prior = self.prior.prior(&inputs);
                let (y, _) = raw(&self.training_outputs.as_vector());
                let mut mean = DVector::<f64>::zeros(m);
                let mut cholesky_covariance = DMatrix::<f64>::zeros(m, m);
                let (xq, ldq) = raw(&inputs);
                // posterior mean and cholesky(cov).unpack() in one call; the covariance itself is not needed (NULL)
                let st = unsafe {
                    fr_posterior(h, &prog, y, xq, m as i64, ldq, prior.as_ptr(), mean.as_mut_ptr(), std::ptr::null_mut(), m as i64,
                                 cholesky_covariance.as_mut_ptr(), m as i64)
                };
                match st
                {
                    FR_NOT_POSITIVE_DEFINITE => panic!("MultivariateNormal: Cholesky decomposition failed!"),
                    st => check(ctx, st)
                }
                // (a crate-private constructor next to `MultivariateNormal::new`: the factor is already there)
                MultivariateNormal::from_cholesky(mean, cholesky_covariance)
            }
            None => self.host_sample_at(&inputs)
        }
    }

    // mod.rs:405-445
    pub fn fit_parameters(&mut self,
                          fit_prior: bool,
                          fit_kernel: bool,
                          max_iter: usize,
                          convergence_fraction: f64,
                          max_time: Duration)
    {
        if fit_prior
        {
            let training_outputs = self.training_outputs.as_vector() + self.prior.prior(&self.training_inputs.as_matrix());
            self.prior.fit(&self.training_inputs.as_matrix(), &training_outputs);
            let training_outputs = training_outputs - self.prior.prior(&self.training_inputs.as_matrix());
            self.training_outputs.assign(&training_outputs);
            if !fit_kernel
            {
                self.refit();
            }
        }
        if fit_kernel
        {
            if self.kernel.is_scalable()
            {
                self.scaled_optimize_parameters(max_iter, convergence_fraction, max_time);
            }
            else
            {
                self.optimize_parameters(max_iter, convergence_fraction, max_time);
            }
        }
        self.refresh_targets();
    }

    /// "Retrains model from scratch" (mod.rs:426-429, optimizer.rs:133-136, :267-270): on the device the training inputs are
    /// already resident in the handle, so only the kernel program and the noise travel -> fr_chol_refactor.
    pub(super) fn refit(&mut self)
    {
        if let Some((_ctx, h, prog)) = self.device()
        {
            let st = unsafe {
                fr_chol_refactor(h, &prog, self.noise, self.cholesky_epsilon.is_some() as i32, self.cholesky_epsilon.unwrap_or(0.))
            };
            match (st, self.cholesky_epsilon)
            {
                (FR_OK, _) => return,
                (FR_NOT_POSITIVE_DEFINITE, Some(cholesky_epsilon)) =>
                {
                    panic!("Cholesky decomposition failed even though we used `cholesky_epsilon` value of {cholesky_epsilon}")
                }
                (FR_NOT_POSITIVE_DEFINITE, None) =>
                {
                    panic!("Cholesky decomposition failed, consider setting `cholesky_epsilon` via `GaussianProcessBuilder`")
                }
                // a kernel whose parameters left the device's range: fall through to a fresh factor (which may choose the host)
                (FR_UNSUPPORTED_KERNEL, _) => {}
                _ => panic!("friedrich_amd (status {}): {}", st, crate::algebra::device::last_error(_ctx))
            }
        }
        self.covmat_cholesky =
            make_cholesky_cov_matrix(&self.training_inputs.as_matrix(), &self.kernel, self.noise, self.cholesky_epsilon);
    }
}

// In multivariate_normal.rs, next to `new` (:54-59):
//
//     /// The factor of the covariance is already known (computed on the device by `fr_posterior`).
//     pub(crate) fn from_cholesky(mean: DVector<f64>, cholesky_covariance: DMatrix<f64>) -> Self
//     {
//         MultivariateNormal { mean, cholesky_covariance, input_type: PhantomData }
//     }

//! Replacement bodies for `src/gaussian_process/optimizer.rs` (friedrich 0.5.1, lines 24-283) with the `friedrich_mi355x` feature.
//!
//! The two gradient functions lose their O(n^2) temporaries: `Cholesky::inverse()` (:32, :169, 2 n^3 flops on one core), the p
//! gradient matrices of `make_gradient_covariance_matrices` and the products with them become ONE call, `fr_grad_terms`, which
//! returns the p (+ 1) scalars.  The ADAM loops are the reference's, statement for statement -- constants, the `p == 0 -> epsilon`
//! guard, the multiplicative update, `noise *= scale` (:263), the chrono budget -- except that "Fits model" is `self.refit()`
//! (`fr_chol_refactor`: the inputs never leave the device).  A host model (`CholeskyHandle::Host`) keeps the original functions,
//! renamed `host_gradient_marginal_likelihood` / `host_scaled_gradient_marginal_likelihood`.
use super::GaussianProcess;
use crate::algebra::device::{check, raw};
use crate::algebra::ffi::*;
use crate::parameters::{kernel::Kernel, prior::Prior};
use chrono::{Duration, Utc};

impl<KernelType: Kernel, PriorType: Prior> GaussianProcess<KernelType, PriorType>
{
    // optimizer.rs:24-60: gradient per kernel parameter followed by the gradient for the noise parameter
    fn gradient_marginal_likelihood(&self) -> Vec<f64>
    {
        match self.device()
        {
            Some((ctx, h, prog)) =>
            {
                let (y, _) = raw(&self.training_outputs.as_vector());
                let mut results = vec![0f64; self.kernel.nb_parameters() + 1];
                let mut unused_scale = 0f64;
                check(ctx, unsafe { fr_grad_terms(h, &prog, y, self.noise, 0, results.as_mut_ptr(), &mut unused_scale) });
                results
            }
            None => self.host_gradient_marginal_likelihood()
        }
    }

    // optimizer.rs:159-203: (optimal scale for kernel + noise, gradient per kernel parameter -- NOT the noise)
    fn scaled_gradient_marginal_likelihood(&self) -> (f64, Vec<f64>)
    {
        match self.device()
        {
            Some((ctx, h, prog)) =>
            {
                let (y, _) = raw(&self.training_outputs.as_vector());
                let mut results = vec![0f64; self.kernel.nb_parameters()];
                let mut scale = 0f64;
                check(ctx, unsafe { fr_grad_terms(h, &prog, y, self.noise, 1, results.as_mut_ptr(), &mut scale) });
                (scale, results)
            }
            None => self.host_scaled_gradient_marginal_likelihood()
        }
    }

    // optimizer.rs:69-149
    pub(super) fn optimize_parameters(&mut self, max_iter: usize, convergence_fraction: f64, max_time: Duration)
    {
        let (beta1, beta2, epsilon, learning_rate) = (0.9f64, 0.999f64, 1e-8f64, 0.1f64);
        let mut parameters: Vec<f64> = self.kernel.get_parameters().iter().map(|&p| if p == 0. { epsilon } else { p }).collect();
        parameters.push(self.noise.ln()); // noise in log-space
        let mut mean_grad = vec![0.; parameters.len()];
        let mut var_grad = vec![0.; parameters.len()];
        let time_start = Utc::now();
        for i in 1..=max_iter
        {
            let mut gradients = self.gradient_marginal_likelihood();
            if let Some(noise_grad) = gradients.last_mut()
            {
                *noise_grad *= self.noise // corrects the noise gradient for log-space
            }
            let had_significant_progress =
                adam_step(&mut parameters, &gradients, &mut mean_grad, &mut var_grad, i, beta1, beta2, epsilon, learning_rate,
                          convergence_fraction);
            self.kernel.set_parameters(&parameters);
            if let Some(noise) = parameters.last()
            {
                self.noise = noise.exp()
            }
            self.refit();
            if (!had_significant_progress) || (Utc::now().signed_duration_since(time_start) > max_time)
            {
                break;
            };
        }
    }

    // optimizer.rs:211-283
    pub(super) fn scaled_optimize_parameters(&mut self, max_iter: usize, convergence_fraction: f64, max_time: Duration)
    {
        let (beta1, beta2, epsilon, learning_rate) = (0.9f64, 0.999f64, 1e-8f64, 0.1f64);
        let mut parameters: Vec<f64> = self.kernel.get_parameters().iter().map(|&p| if p == 0. { epsilon } else { p }).collect();
        let mut mean_grad = vec![0.; parameters.len()];
        let mut var_grad = vec![0.; parameters.len()];
        let time_start = Utc::now();
        for i in 1..=max_iter
        {
            let (scale, gradients) = self.scaled_gradient_marginal_likelihood();
            let had_significant_progress =
                adam_step(&mut parameters, &gradients, &mut mean_grad, &mut var_grad, i, beta1, beta2, epsilon, learning_rate,
                          convergence_fraction);
            self.kernel.set_parameters(&parameters);
            self.kernel.rescale(scale);
            self.noise *= scale;
            parameters = self.kernel.get_parameters(); // they have been rescaled
            self.refit();
            if (!had_significant_progress) || (Utc::now().signed_duration_since(time_start) > max_time)
            {
                break;
            };
        }
    }
}

/// One ADAM update of every parameter, multiplicative as in the reference (optimizer.rs:111-120, :245-254); returns whether any
/// relative step exceeded `convergence_fraction`.
#[allow(clippy::too_many_arguments)]
fn adam_step(parameters: &mut [f64], gradients: &[f64], mean_grad: &mut [f64], var_grad: &mut [f64], i: usize, beta1: f64, beta2: f64,
             epsilon: f64, learning_rate: f64, convergence_fraction: f64)
             -> bool
{
    let mut had_significant_progress = false;
    for p in 0..parameters.len()
    {
        mean_grad[p] = beta1 * mean_grad[p] + (1. - beta1) * gradients[p];
        var_grad[p] = beta2 * var_grad[p] + (1. - beta2) * gradients[p].powi(2);
        let bias_corrected_mean = mean_grad[p] / (1. - beta1.powi(i as i32));
        let bias_corrected_variance = var_grad[p] / (1. - beta2.powi(i as i32));
        let delta = learning_rate * bias_corrected_mean / (bias_corrected_variance.sqrt() + epsilon);
        had_significant_progress |= delta.abs() > convergence_fraction;
        parameters[p] *= 1. + delta;
    }
    had_significant_progress
}

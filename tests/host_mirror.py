"""Test support: the reference's HOST logic around the device boundary, restated in Python over the ctypes binding.

In a real integration these lines stay friedrich's unchanged Rust (builder defaults, priors, the ADAM scalar loop:
SURVEY.md section 2, out of scope); they are mirrored here only so that the tests can drive the C ABI the way
`GaussianProcess::default` / `fit_parameters` do and compare with the oracle's golden values.  Everything O(n^2) and up
goes through libfriedrich_amd.so (Gram + Cholesky, K^-1 reductions of the gradient, predict); nothing here touches
oracle/.

    GaussianProcess::new            src/gaussian_process/mod.rs:142-167
    GaussianProcess::default        mod.rs:96-102 -> builder.rs:66-95, 189-214
    fit_parameters                  mod.rs:406-445
    optimize_parameters             optimizer.rs:69-149
    scaled_optimize_parameters      optimizer.rs:211-283
"""
import numpy as np

SCALABLE = ("squared_exp", "gaussian", "exponential", "matern1", "matern2")


def nalgebra_variance(y):  # [nalgebra 0.31] Matrix::variance(): E[x^2] - E[x]^2, sequential folds
    s2 = s1 = 0.0
    for v in np.asarray(y, dtype=np.float64):
        s2 = s2 + v * v
        s1 = s1 + v
    d = 1.0 / len(y)
    return s2 * d - (s1 * d) * (s1 * d)


def nalgebra_mean(y):
    s = 0.0
    for v in np.asarray(y, dtype=np.float64):
        s += v
    return s / len(y)


# ---- kernel specs as parameter lists (Kernel::get_parameters / set_parameters / rescale / is_scalable) --------------
def get_parameters(spec):
    if spec[0] in ("sum", "prod"):
        return get_parameters(spec[1]) + get_parameters(spec[2])
    return [float(v) for v in spec[1:]]


def set_parameters(spec, params):
    """-> (new spec, number of parameters consumed)"""
    if spec[0] in ("sum", "prod"):
        a, na = set_parameters(spec[1], params)
        b, nb = set_parameters(spec[2], params[na:])
        return (spec[0], a, b), na + nb
    k = len(spec) - 1
    return (spec[0],) + tuple(float(v) for v in params[:k]), k


def is_scalable(spec):
    if spec[0] == "sum":  # kernel.rs:152
        return is_scalable(spec[1]) and is_scalable(spec[2])
    if spec[0] == "prod":  # :241
        return is_scalable(spec[1]) or is_scalable(spec[2])
    return spec[0] in SCALABLE


def rescale(spec, scale):
    if spec[0] == "sum":  # kernel.rs:174-178
        return ("sum", rescale(spec[1], scale), rescale(spec[2], scale))
    if spec[0] == "prod":  # :264-274: the first scalable factor takes the scale
        if is_scalable(spec[1]):
            return ("prod", rescale(spec[1], scale), spec[2])
        return ("prod", spec[1], rescale(spec[2], scale))
    assert spec[0] in SCALABLE
    return (spec[0], spec[1], spec[2] * scale)  # ampl *= scale (kernel.rs:578-581, 683-686, ...)


def heuristic_fit(spec, ls, var_y):
    if spec[0] in ("sum", "prod"):
        return (spec[0], heuristic_fit(spec[1], ls, var_y), heuristic_fit(spec[2], ls, var_y))
    if spec[0] in SCALABLE:  # kernel.rs:594-600 ...
        return (spec[0], ls, var_y)
    return spec


class DeviceGP:
    """GaussianProcess<Kernel, ConstantPrior> with its covmat_cholesky on the device (friedrich_amd.device.Cholesky)"""

    def __init__(self, ctx, prior_c, kernel, noise, cholesky_epsilon, X, y):  # mod.rs:142-167
        self.ctx = ctx
        self.prior_c = float(prior_c)
        self.kernel = kernel
        self.noise = float(noise)
        self.eps = cholesky_epsilon
        self.X = np.asfortranarray(np.asarray(X, dtype=np.float64))
        self.y = np.asarray(y, dtype=np.float64) - self.prior_c  # :156
        self.chol = ctx.cholesky_from_inputs(kernel, self.X, self.noise, eps=cholesky_epsilon)
        self.iterations = 0

    @classmethod
    def default(cls, ctx, X, y, max_iter=100, convergence_fraction=0.05):  # mod.rs:96-102, builder.rs:66-95, 189-214
        X = np.asfortranarray(np.asarray(X, dtype=np.float64))
        y = np.asarray(y, dtype=np.float64)
        noise = 0.1 * np.sqrt(nalgebra_variance(y))  # builder.rs:73
        ls = ctx.mean_pairwise_distance(X)  # kernel.rs:94-113 on the device (K3)
        kernel = heuristic_fit(("squared_exp", 1.0, 1.0), ls, nalgebra_variance(y))  # builder.rs:195
        gp = cls(ctx, 0.0, kernel, noise, None, X, y)
        gp.fit_parameters(True, True, max_iter, convergence_fraction)
        return gp

    def close(self):
        self.chol.free()

    def fit_parameters(self, fit_prior, fit_kernel, max_iter=100, convergence_fraction=0.05):  # mod.rs:406-445
        if fit_prior:
            y_full = self.y + self.prior_c  # :416-417
            self.prior_c = nalgebra_mean(y_full)  # ConstantPrior::fit prior.rs:97
            self.y = y_full - self.prior_c
            if not fit_kernel:  # :423-430
                self.chol.refactor(self.kernel, self.noise, eps=self.eps)
        if fit_kernel:  # :434-444
            if is_scalable(self.kernel):
                self._scaled_optimize(max_iter, convergence_fraction)
            else:
                self._optimize(max_iter, convergence_fraction)

    def _adam(self, i, gradients, parameters, mean_grad, var_grad, convergence_fraction):
        beta1, beta2, epsilon, learning_rate = 0.9, 0.999, 1e-8, 0.1  # optimizer.rs:79-82 / :221-224
        progress = False
        for q in range(len(parameters)):  # :113-122 / :249-258
            mean_grad[q] = beta1 * mean_grad[q] + (1.0 - beta1) * gradients[q]
            var_grad[q] = beta2 * var_grad[q] + (1.0 - beta2) * gradients[q] ** 2
            bias_corrected_mean = mean_grad[q] / (1.0 - beta1 ** i)
            bias_corrected_variance = var_grad[q] / (1.0 - beta2 ** i)
            delta = learning_rate * bias_corrected_mean / (np.sqrt(bias_corrected_variance) + epsilon)
            progress |= abs(delta) > convergence_fraction
            parameters[q] *= 1.0 + delta
        return progress

    def _scaled_optimize(self, max_iter, convergence_fraction):  # optimizer.rs:211-283
        parameters = [p if p != 0.0 else 1e-8 for p in get_parameters(self.kernel)]  # :226-239
        mean_grad, var_grad = [0.0] * len(parameters), [0.0] * len(parameters)
        for i in range(1, max_iter + 1):
            self.iterations = i
            g, scale = self.chol.grad_terms(self.kernel, self.y, self.noise, scaled=True,
                                            nb_parameters=len(parameters))  # :246 on the device
            progress = self._adam(i, g, parameters, mean_grad, var_grad, convergence_fraction)
            self.kernel, _ = set_parameters(self.kernel, parameters)  # :261
            self.kernel = rescale(self.kernel, scale)  # :262
            self.noise *= scale  # :263 (noise, not noise^2)
            parameters = get_parameters(self.kernel)  # :264
            self.chol.refactor(self.kernel, self.noise, eps=self.eps)  # :267-270
            if not progress:  # :272 (time budget not modelled)
                break

    def _optimize(self, max_iter, convergence_fraction):  # optimizer.rs:69-149
        parameters = [p if p != 0.0 else 1e-8 for p in get_parameters(self.kernel)]
        npar = len(parameters)
        parameters.append(np.log(self.noise))  # :98
        mean_grad, var_grad = [0.0] * (npar + 1), [0.0] * (npar + 1)
        for i in range(1, max_iter + 1):
            self.iterations = i
            g, _ = self.chol.grad_terms(self.kernel, self.y, self.noise, scaled=False, nb_parameters=npar)
            g = list(g)
            g[-1] *= self.noise  # :106-110
            progress = self._adam(i, g, parameters, mean_grad, var_grad, convergence_fraction)
            self.kernel, _ = set_parameters(self.kernel, parameters[:npar])  # :125
            self.noise = float(np.exp(parameters[npar]))  # :126-130
            self.chol.refactor(self.kernel, self.noise, eps=self.eps)  # :133-136
            if not progress:  # :138
                break

    # the predict family (prior evaluated on the host, as prior.rs does)
    def predict(self, Xq):
        Xq = np.asfortranarray(np.asarray(Xq, dtype=np.float64))
        return self.chol.predict_mean(self.kernel, self.y, Xq, np.full(Xq.shape[0], self.prior_c))

    def predict_variance(self, Xq):
        return self.chol.predict_variance(self.kernel, np.asfortranarray(np.asarray(Xq, dtype=np.float64)))

    def likelihood(self):
        return self.chol.likelihood(self.kernel, self.y, self.noise)

"""Thin object wrappers over the C ABI (include/friedrich_amd.h): Context and Cholesky handles.

Matrices may be
  * numpy arrays (host; copied to column-major float64 if they are not already), or
  * torch tensors that live on the GPU and are column-major (stride(0) == 1) -- passed zero-copy by pointer.
All numerical work happens inside libfriedrich_amd.so; this file only marshals pointers, sizes and statuses.
"""
import ctypes
import weakref

import numpy as np

from . import _capi as C
from . import _capi as C_


class FriedrichError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{C.STATUS_NAMES.get(status, status)}: {message}")
        self.status = status
        self.message = message


def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "stride")


class DeviceMatrix:
    """a column-major n x d matrix in HBM produced by fr_inputs_to_device (the `Input` staging of conversion/mod.rs);
    accepted wherever a data matrix is"""

    def __init__(self, ctx, ptr, rows, cols, ld):
        self.ctx, self.ptr, self.rows, self.cols, self.ld = ctx, ptr, rows, cols, ld

    def free(self):
        if self.ptr and getattr(self.ctx, "h", None):
            self.ctx.lib.fr_device_free(self.ctx.h, ctypes.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _Mat:
    """pointer + shape + ld view of a host/device matrix; keeps the backing object alive"""

    def __init__(self, obj, writable=False, vector=False):
        if isinstance(obj, DeviceMatrix):
            self.rows, self.cols, self.ld = obj.rows, obj.cols, obj.ld
            self.keep, self.ptr, self.host = obj, obj.ptr, None
        elif _is_torch(obj):
            t = obj
            if t.dim() == 1:
                if t.numel() > 1 and t.stride(0) != 1:
                    raise ValueError("device vectors must be contiguous")
                self.rows, self.cols, self.ld = t.shape[0], 1, max(t.shape[0], 1)
            else:
                if t.shape[0] > 1 and t.stride(0) != 1:
                    raise ValueError("device matrices must be column-major (stride(0) == 1)")
                self.rows, self.cols = t.shape
                self.ld = t.stride(1) if t.shape[1] > 1 else max(t.shape[0], 1)
                if self.ld < self.rows:
                    raise ValueError("bad leading dimension")
            if str(t.dtype) != "torch.float64":
                raise ValueError("float64 required")
            self.keep = t
            self.ptr = t.data_ptr()
            self.host = None
        else:
            if writable:
                if not (isinstance(obj, np.ndarray) and obj.dtype == np.float64):
                    raise ValueError("writable host matrices must be float64 numpy arrays")
                arr = obj.reshape(-1, 1) if obj.ndim == 1 else obj
                if not (arr.flags.f_contiguous or (arr.shape[1] == 1 and arr.flags.c_contiguous)):
                    raise ValueError("writable host matrices must be column-major")
            else:
                a = np.asarray(obj, dtype=np.float64)
                if a.ndim == 1:
                    a = a.reshape(-1, 1)
                # the leading rows of a column-major matrix are a column-major matrix with a leading dimension -- what the ABI
                # takes (EMatrix::as_matrix has ld = capacity too): no copy
                if a.ndim == 2 and a.shape[0] > 0 and a.shape[1] > 1 and a.strides[0] == 8 and a.strides[1] % 8 == 0 and a.strides[1] // 8 >= a.shape[0]:
                    self.rows, self.cols, self.ld = a.shape[0], a.shape[1], a.strides[1] // 8
                    self.keep, self.ptr, self.host = a, a.ctypes.data, a
                    return
                arr = np.asfortranarray(a)
            self.rows, self.cols = arr.shape
            self.ld = max(self.rows, 1)
            self.keep = arr
            self.ptr = arr.ctypes.data
            self.host = arr


def _vecptr(v):
    """-> (pointer or None, length, keepalive)"""
    if v is None:
        return None, 0, None
    if _is_torch(v):
        return v.data_ptr(), v.numel(), v
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
    return a.ctypes.data, a.shape[0], a


class Context:
    """fr_ctx: one per (process, GPU).  Fails loudly when no gfx950 device / library is present."""

    def __init__(self, device=-1):
        self.lib = C.load()
        h = ctypes.c_void_p()
        st = self.lib.fr_ctx_create(ctypes.byref(h), int(device))
        if st != C.FR_OK:
            raise FriedrichError(st, "fr_ctx_create failed (is a gfx950 GPU visible?)")
        self.h = h
        self._children = weakref.WeakSet()  # live Cholesky handles: freed before the context goes away

    def close(self):
        if getattr(self, "h", None):
            for child in list(self._children):
                child.free()
            self.lib.fr_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, st, allow=()):
        if st != C.FR_OK and st not in allow:
            raise FriedrichError(st, self.lib.fr_last_error(self.h).decode())
        return st

    def set_stream(self, hip_stream_ptr):
        self.check(self.lib.fr_ctx_set_stream(self.h, ctypes.c_void_p(hip_stream_ptr)))

    def set_option(self, name, value):
        self.check(self.lib.fr_ctx_set_option(self.h, name.encode(), int(value)))

    def counter(self, name):
        v = ctypes.c_int64(0)
        self.check(self.lib.fr_ctx_get_counter(self.h, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def synchronize(self):
        self.check(self.lib.fr_ctx_synchronize(self.h))

    # profiling ------------------------------------------------------------------------------------
    def profile_enable(self, on=True, classes=None):
        """classes: iterable of class names (see _capi.PROF_NAMES) to time; None = all"""
        flag = 1 if on else 0
        if on and classes is not None:
            flag = 0
            for name in classes:
                flag |= 1 << (C.PROF_NAMES.index(name) + 1)
        self.check(self.lib.fr_ctx_profile_enable(self.h, flag))

    def profile_reset(self):
        self.check(self.lib.fr_ctx_profile_reset(self.h))

    def profile(self):
        out = {}
        for cls, name in enumerate(C.PROF_NAMES):
            ms, n, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            self.check(self.lib.fr_ctx_profile_get(self.h, cls, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl),
                                                   ctypes.byref(by)))
            out[name] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
        return out

    # communicator ---------------------------------------------------------------------------------
    def comm_unique_id(self):
        buf = ctypes.create_string_buffer(C.FR_COMM_ID_BYTES)
        st = self.lib.fr_comm_unique_id(buf)
        if st != C.FR_OK:
            raise FriedrichError(st, "fr_comm_unique_id failed")
        return buf.raw

    def comm_init(self, rank, world_size, unique_id=None):
        buf = ctypes.create_string_buffer(unique_id, C.FR_COMM_ID_BYTES) if unique_id is not None else None
        self.check(self.lib.fr_ctx_comm_init(self.h, int(rank), int(world_size), buf))

    def comm_selftest(self):
        """one broadcast + one all-gather through the attached communicator, verified (collective)"""
        self.check(self.lib.fr_ctx_comm_selftest(self.h))

    def comm_finalize(self, abort=False):
        """detach the communicator (after a time-out: tear it down without waiting for the peers); a new one can be attached"""
        self.check(self.lib.fr_ctx_comm_finalize(self.h, 1 if abort else 0))

    def comm_init_local(self, group_id, rank, world_size):
        """in-process transport (ranks = threads sharing one GPU); see fr_ctx_comm_init_local"""
        self.check(self.lib.fr_ctx_comm_init_local(self.h, int(group_id), int(rank), int(world_size)))

    # src/conversion/mod.rs ------------------------------------------------------------------------
    def inputs_to_device(self, data, layout="rowmajor"):
        """Input::to_dmatrix + upload in one pass (pinned bounce buffer, device-side transpose of row-major samples).
        data: 2-D numpy array in the named layout ("colmajor" | "rowmajor"), or a list of 1-D arrays ("rowptrs" = Vec<Vec<f64>>)"""
        dev, ld = ctypes.c_void_p(), ctypes.c_int64()
        if layout == "rowptrs":
            rows = [np.ascontiguousarray(np.asarray(r, dtype=np.float64)) for r in data]
            n, d = len(rows), (rows[0].shape[0] if rows else 0)
            ptrs = (ctypes.c_void_p * max(n, 1))(*[r.ctypes.data for r in rows])
            st = self.lib.fr_inputs_to_device(self.h, 2, ctypes.cast(ptrs, ctypes.c_void_p), n, d, 0, ctypes.byref(dev),
                                              ctypes.byref(ld))
        else:
            a = np.asarray(data, dtype=np.float64)
            n, d = a.shape
            if layout == "rowmajor":
                a = np.ascontiguousarray(a)
                st = self.lib.fr_inputs_to_device(self.h, 1, ctypes.c_void_p(a.ctypes.data), n, d, max(d, 1), ctypes.byref(dev),
                                                  ctypes.byref(ld))
            else:
                a = np.asfortranarray(a)
                st = self.lib.fr_inputs_to_device(self.h, 0, ctypes.c_void_p(a.ctypes.data), n, d, max(n, 1), ctypes.byref(dev),
                                                  ctypes.byref(ld))
        self.check(st)
        return DeviceMatrix(self, dev.value, n, d, ld.value)

    # src/parameters/prior.rs ----------------------------------------------------------------------
    def linear_prior_fit(self, X, y):
        """LinearPrior::fit (prior.rs:139-159) -> (weights [d], intercept)"""
        x = _Mat(X)
        yp, _, keep = _vecptr(y)
        w = (ctypes.c_double * max(x.cols, 1))()
        b = ctypes.c_double()
        self.check(self.lib.fr_linear_prior_fit(self.h, x.ptr, x.rows, x.ld, x.cols, yp, w, ctypes.byref(b)))
        return np.array(w[:x.cols]), b.value

    # src/algebra/mod.rs ---------------------------------------------------------------------------
    def gram(self, kernel, A, B, out=None):
        """make_covariance_matrix (algebra/mod.rs:41-54)"""
        p = C.kprog(kernel)
        a, b = _Mat(A), _Mat(B)
        if a.cols != b.cols:
            raise FriedrichError(C.FR_SHAPE, "feature counts differ")
        if out is None:
            out = np.empty((a.rows, b.rows), order="F")
        o = _Mat(out, writable=True)
        self.check(self.lib.fr_gram(self.h, ctypes.byref(p), a.ptr, a.rows, a.ld, b.ptr, b.rows, b.ld, a.cols, o.ptr,
                                    o.ld))
        return out

    def gemm(self, A, B, C=None, trans_a=False, trans_b=False, alpha=1.0, beta=0.0):
        """DMatrix::gemm / gemm_tr"""
        a, b = _Mat(A), _Mat(B)
        M, K = (a.cols, a.rows) if trans_a else (a.rows, a.cols)
        K2, N = (b.cols, b.rows) if trans_b else (b.rows, b.cols)
        if K != K2:
            raise FriedrichError(C_.FR_SHAPE, "inner dimensions differ")
        if C is None:
            C = np.zeros((M, N), order="F")
        c = _Mat(C, writable=True)
        self.check(self.lib.fr_gemm(self.h, int(trans_a), int(trans_b), M, N, K, float(alpha), a.ptr, a.ld, b.ptr,
                                    b.ld, float(beta), c.ptr, c.ld))
        return C

    def mean_pairwise_distance(self, X):
        """fit_bandwidth_mean (kernel.rs:94-113)"""
        x = _Mat(X)
        out = ctypes.c_double()
        self.check(self.lib.fr_mean_pairwise_distance(self.h, x.ptr, x.rows, x.ld, x.cols, ctypes.byref(out)))
        return out.value

    def cholesky_from_inputs(self, kernel, X, noise, eps=None, capacity_hint=0, allow_failure=False):
        """make_cholesky_cov_matrix (algebra/mod.rs:59-92)"""
        p = C.kprog(kernel)
        x = _Mat(X)
        h = ctypes.c_void_p()
        st = self.lib.fr_chol_from_inputs(self.h, ctypes.byref(p), x.ptr, x.rows, x.ld, x.cols, float(noise),
                                          0 if eps is None else 1, 0.0 if eps is None else float(eps),
                                          int(capacity_hint), ctypes.byref(h))
        chol = Cholesky(self, h) if h else None
        if st == C.FR_NOT_POSITIVE_DEFINITE and allow_failure:
            return chol
        if st != C.FR_OK:
            msg = self.lib.fr_last_error(self.h).decode()
            if chol is not None:
                chol.free()
            raise FriedrichError(st, msg)
        return chol

    def cholesky_from_matrix(self, A, eps=None, allow_failure=False):
        """DMatrix::cholesky() / new_with_substitute (multivariate_normal.rs:57)"""
        a = _Mat(A)
        h = ctypes.c_void_p()
        st = self.lib.fr_chol_from_matrix(self.h, a.ptr, a.rows, a.ld, 0 if eps is None else 1,
                                          0.0 if eps is None else float(eps), ctypes.byref(h))
        chol = Cholesky(self, h) if h else None
        if st == C.FR_NOT_POSITIVE_DEFINITE and allow_failure:
            return chol
        if st != C.FR_OK:
            msg = self.lib.fr_last_error(self.h).decode()
            if chol is not None:
                chol.free()
            raise FriedrichError(st, msg)
        return chol

    def cholesky_upload(self, L, X, capacity_hint=0):
        l, x = _Mat(L), _Mat(X)
        h = ctypes.c_void_p()
        self.check(self.lib.fr_chol_upload_l(self.h, l.ptr, l.rows, l.ld, x.ptr, x.ld, x.cols, int(capacity_hint),
                                             ctypes.byref(h)))
        return Cholesky(self, h)


class Cholesky:
    """fr_chol: the device-resident stand-in for GaussianProcess::covmat_cholesky (mod.rs:78)"""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.lib = ctx.lib
        self.h = handle
        ctx._children.add(self)

    def free(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):  # the factor's stream lives in the context
                self.lib.fr_chol_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def info(self):
        v = [ctypes.c_int64() for _ in range(5)]
        self.ctx.check(self.lib.fr_chol_info(self.h, *[ctypes.byref(x) for x in v]))
        return dict(zip(("n", "capacity", "d", "n_subst", "fail_col"), (x.value for x in v)))

    @property
    def n(self):
        return self.info()["n"]

    def conditioning(self):
        """-> (largest conditioning estimate of the 128 x 128 diagonal blocks, whether the handle refines)"""
        est, ref = ctypes.c_double(), ctypes.c_int()
        self.ctx.check(self.lib.fr_chol_conditioning(self.h, ctypes.byref(est), ctypes.byref(ref)))
        return est.value, bool(ref.value)

    def substitutions(self):
        ns = self.info()["n_subst"]
        idx = (ctypes.c_int64 * max(ns, 1))()
        self.ctx.check(self.lib.fr_chol_substitutions(self.h, idx, ns))
        return np.array(idx[:ns], dtype=np.int64)

    def refactor(self, kernel, noise, eps=None, allow_failure=False):
        p = C.kprog(kernel)
        st = self.lib.fr_chol_refactor(self.h, ctypes.byref(p), float(noise), 0 if eps is None else 1,
                                       0.0 if eps is None else float(eps))
        if st == C.FR_NOT_POSITIVE_DEFINITE and allow_failure:
            return st
        return self.ctx.check(st)

    def add_rows(self, kernel, X_all, nb_new, noise):
        """add_rows_cholesky_cov_matrix (algebra/mod.rs:97-126)"""
        p = C.kprog(kernel)
        x = _Mat(X_all)
        self.ctx.check(self.lib.fr_chol_add_rows(self.h, ctypes.byref(p), x.ptr, x.rows, x.ld, x.cols, int(nb_new),
                                                 float(noise)))

    def solve(self, B):
        """Cholesky::solve (clone + solve_mut)"""
        B = _copy_in(B)
        b = _Mat(B, writable=True)
        self.ctx.check(self.lib.fr_chol_solve(self.h, b.ptr, b.cols, b.ld))
        return B

    def solve_lower(self, B):
        """l().solve_lower_triangular"""
        B = _copy_in(B)
        b = _Mat(B, writable=True)
        self.ctx.check(self.lib.fr_chol_solve_lower(self.h, b.ptr, b.cols, b.ld))
        return B

    def inverse(self):
        n = self.n
        out = np.empty((n, n), order="F")
        self.ctx.check(self.lib.fr_chol_inverse(self.h, out.ctypes.data, max(n, 1)))
        return out

    def l(self, nan_upper=False):
        n = self.n
        out = np.empty((n, n), order="F")
        self.ctx.check(self.lib.fr_chol_download_l(self.h, out.ctypes.data, max(n, 1), 1 if nan_upper else 0))
        return out

    # src/gaussian_process/mod.rs ------------------------------------------------------------------
    def set_targets(self, y):
        """cache the residual training outputs: predict_mean(kernel, None, ...) then uses alpha = K^-1 y, solved once per
        change of the factor (todo.md:10)"""
        yp, _, keep = _vecptr(y)
        self.ctx.check(self.lib.fr_chol_set_targets(self.h, yp))

    def likelihood(self, kernel, y, noise):
        p = C.kprog(kernel)
        yp, _, keep = _vecptr(y)
        out = ctypes.c_double()
        self.ctx.check(self.lib.fr_likelihood(self.h, ctypes.byref(p), yp, float(noise), ctypes.byref(out)))
        return out.value

    def predict_mean(self, kernel, y, Xq, prior_q=None, out=None):
        p = C.kprog(kernel)
        q = _Mat(Xq)
        yp, _, k1 = _vecptr(y)
        pp, _, k2 = _vecptr(prior_q)
        if out is None:
            out = np.empty(q.rows)
        op, _, k3 = _vecptr_out(out)
        self.ctx.check(self.lib.fr_predict_mean(self.h, ctypes.byref(p), yp, q.ptr, q.rows, q.ld, pp, op))
        return out

    def predict_variance(self, kernel, Xq, out=None):
        p = C.kprog(kernel)
        q = _Mat(Xq)
        if out is None:
            out = np.empty(q.rows)
        op, _, k3 = _vecptr_out(out)
        self.ctx.check(self.lib.fr_predict_variance(self.h, ctypes.byref(p), q.ptr, q.rows, q.ld, op))
        return out

    def predict_mean_variance(self, kernel, y, Xq, prior_q=None):
        p = C.kprog(kernel)
        q = _Mat(Xq)
        yp, _, k1 = _vecptr(y)
        pp, _, k2 = _vecptr(prior_q)
        mean, var = np.empty(q.rows), np.empty(q.rows)
        self.ctx.check(self.lib.fr_predict_mean_variance(self.h, ctypes.byref(p), yp, q.ptr, q.rows, q.ld, pp,
                                                         mean.ctypes.data, var.ctypes.data))
        return mean, var

    def predict_covariance(self, kernel, Xq):
        p = C.kprog(kernel)
        q = _Mat(Xq)
        cov = np.empty((q.rows, q.rows), order="F")
        self.ctx.check(self.lib.fr_predict_covariance(self.h, ctypes.byref(p), q.ptr, q.rows, q.ld, cov.ctypes.data,
                                                      max(q.rows, 1)))
        return cov

    def grad_terms(self, kernel, y, noise, scaled, nb_parameters):
        """gradient_marginal_likelihood (optimizer.rs:24-60) / scaled_gradient_marginal_likelihood (:159-203)
        -> (gradients [nb_parameters (+1 noise when not scaled)], scale)"""
        p = C.kprog(kernel)
        yp, _, k1 = _vecptr(y)
        out = (ctypes.c_double * (nb_parameters + 1))()
        scale = ctypes.c_double()
        self.ctx.check(self.lib.fr_grad_terms(self.h, ctypes.byref(p), yp, float(noise), 1 if scaled else 0, out,
                                              ctypes.byref(scale)))
        n = nb_parameters + (0 if scaled else 1)
        return np.array(out[:n]), scale.value

    def posterior(self, kernel, y, Xq, prior_q=None):
        """sample_at: -> (mean, cov, cholesky(cov).unpack())"""
        p = C.kprog(kernel)
        q = _Mat(Xq)
        yp, _, k1 = _vecptr(y)
        pp, _, k2 = _vecptr(prior_q)
        m = q.rows
        mean = np.empty(m)
        cov = np.empty((m, m), order="F")
        cov_l = np.empty((m, m), order="F")
        self.ctx.check(self.lib.fr_posterior(self.h, ctypes.byref(p), yp, q.ptr, m, q.ld, pp, mean.ctypes.data,
                                             cov.ctypes.data, max(m, 1), cov_l.ctypes.data, max(m, 1)))
        return mean, cov, cov_l


def _copy_in(B):
    if _is_torch(B):
        return B
    a = np.array(B, dtype=np.float64, order="F", copy=True)
    if a.ndim == 1:
        a = a.reshape(-1, 1, order="F")
    return a


def _vecptr_out(v):
    if _is_torch(v):
        return v.data_ptr(), v.numel(), v
    if not (isinstance(v, np.ndarray) and v.dtype == np.float64 and v.flags.c_contiguous):
        raise ValueError("output vectors must be contiguous float64 numpy arrays or device tensors")
    return v.ctypes.data, v.size, v

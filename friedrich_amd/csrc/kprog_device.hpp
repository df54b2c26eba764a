// kprog_device.hpp -- device-side evaluation of a friedrich kernel program for one (x, y) pair, given
// s = ||x - y||^2 and u = x . y.  Formulas follow src/parameters/kernel.rs statement by statement
// (file:line at each case); mul/add are kept unfused so the values match the reference's rounding up to
// the last-ulp differences between the device and host libm (exp/pow/tanh/hypot).
#pragma once

#include <hip/hip_runtime.h>

#include "friedrich_amd.h"

namespace fr {

enum { NEED_S = 1, NEED_U = 2 };

inline bool kind_is_leaf(int k) { return k >= FR_K_LINEAR && k <= FR_K_RATIONALQUADRATIC; }

inline int leaf_nvalues(int kind)
{
    switch (kind) {
    case FR_K_LINEAR: return 1;
    case FR_K_POLYNOMIAL: return 3;
    case FR_K_MULTIQUADRIC: return 1;
    default: return 2;
    }
}

// which pair statistics a program needs
inline int kprog_needs(const fr_kprog& p)
{
    int needs = 0;
    for (int i = 0; i < p.nops; ++i) {
        switch (p.ops[i].kind) {
        case FR_K_LINEAR:
        case FR_K_POLYNOMIAL:
        case FR_K_HYPERTAN: needs |= NEED_U; break;
        case FR_K_SUM:
        case FR_K_PROD: break;
        default: needs |= NEED_S; break;
        }
    }
    return needs;
}

// ---- the two operations that dominated the instruction count of a pair (rocprofv3 SQ_INSTS_VALU, round 2: ~137 VALU
// instructions per pair at d = 16, 48 of them the distance) ------------------------------------------------------------------
// a / c for a divisor that is uniform over the launch (2 ls^2, |ls|, ...): the reciprocal is loop-invariant (hoisted by the
// compiler), the quotient is q0 = a rc corrected by one fused residual step, q0 + rc (a - q0 c) -- three instructions instead of
// the ~11 of the IEEE division sequence.  The corrected quotient is the correctly rounded one except for rare last-bit cases
// (Markstein's theorem needs an exactly rounded reciprocal, which rc is: it comes from a true division): |error| <= 1 ulp.
// Non-finite quotients (c = 0, infinite operands) take the true division, so the reference's inf / NaN pattern is kept.
__device__ __forceinline__ double div_uniform(double a, double c)
{
    const double rc = 1.0 / c;
    const double q0 = a * rc;
    const double q = __builtin_fma(__builtin_fma(-q0, c, a), rc, q0);
    if (__builtin_expect(!(__builtin_fabs(q) < __builtin_inf()), 0)) return a / c;
    return q;
}

// exp(x), |error| < 1 ulp (checked against 50-digit mpmath in tests/test_gpu_parity.py): n = rint(x log2 e); r = x - n ln 2 in
// two fused steps (Cody-Waite, |r| <= 0.3466); exp(r) by its Taylor polynomial of degree 13 (remainder 0.3466^14 / 14! = 4e-18
// relative) in Horner form; scaled by 2^n with v_ldexp_f64 (one rounding: gradual underflow comes out right).  ~21 instructions
// against ~35 of the library routine, which carries the full special-case handling of an arbitrary argument.  Arguments below
// -746 give 0; NaN propagates; overflow gives +inf (through ldexp up to the first arguments whose n no longer fits, and by the
// explicit select beyond -- x = +inf would otherwise reduce to inf - inf = NaN.  Positive arguments are reachable: the Matern-5/2
// GRADIENT keeps the reference's signed length scale, kernel.rs:881-900, so a negative ls gives exp(+x)).
__device__ __forceinline__ double exp_fast(double x)
{
    const double n = __builtin_rint(x * 1.44269504088896338700e+00);
    double r = __builtin_fma(n, -6.93147180369123816490e-01, x);
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);
    double p = 1.60590438368216145994e-10;           // 1 / 13!
    p = __builtin_fma(p, r, 2.08767569878680989792e-09);  // 1 / 12!
    p = __builtin_fma(p, r, 2.50521083854417187751e-08);  // 1 / 11!
    p = __builtin_fma(p, r, 2.75573192239858906526e-07);  // 1 / 10!
    p = __builtin_fma(p, r, 2.75573192239858906526e-06);  // 1 / 9!
    p = __builtin_fma(p, r, 2.48015873015873015873e-05);  // 1 / 8!
    p = __builtin_fma(p, r, 1.98412698412698412698e-04);  // 1 / 7!
    p = __builtin_fma(p, r, 1.38888888888888888889e-03);  // 1 / 6!
    p = __builtin_fma(p, r, 8.33333333333333333333e-03);  // 1 / 5!
    p = __builtin_fma(p, r, 4.16666666666666666667e-02);  // 1 / 4!
    p = __builtin_fma(p, r, 1.66666666666666666667e-01);  // 1 / 3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double e = __builtin_ldexp(p, (int)n);
    return x < -746.0 ? 0.0 : (x > 710.0 ? __builtin_inf() : e);
}

// KIND >= 0: the leaf kind is a compile-time constant (single-leaf programs get a kernel of their own: no dispatch, the
// other eight formulas are not even compiled in); KIND < 0: taken from the op at run time.
template <int KIND>
__device__ __forceinline__ double leaf_eval_k(const fr_kernel_op& op, double s, double u)
{
#pragma clang fp contract(off)
    const double p0 = op.params[0], p1 = op.params[1], p2 = op.params[2];
    const int kind = KIND >= 0 ? KIND : op.kind;
    switch (kind) {
    case FR_K_LINEAR:  // kernel.rs:381
        return u + p0;
    case FR_K_POLYNOMIAL:  // :456
        return pow(p0 * u + p1, p2);
    case FR_K_SQUAREDEXP: {  // :556-560
        const double x = -div_uniform(s, 2.0 * p0 * p0);
        return fabs(p1) * exp_fast(x);
    }
    case FR_K_EXPONENTIAL: {  // :661-665
        const double x = -div_uniform(sqrt(s), 2.0 * p0 * p0);
        return fabs(p1) * exp_fast(x);
    }
    case FR_K_MATERN1: {  // :766-771
        const double x = div_uniform(sqrt(3.0) * sqrt(s), fabs(p0));
        return fabs(p1) * (1.0 + x) * exp_fast(-x);
    }
    case FR_K_MATERN2: {  // :873-878
        const double l = fabs(p0);
        const double dist = sqrt(s);
        const double x = div_uniform(sqrt(5.0) * dist, l);
        return fabs(p1) * (1.0 + x + div_uniform(5.0 * dist * dist, 3.0 * l * l)) * exp_fast(-x);
    }
    case FR_K_HYPERTAN:  // :976
        return tanh(p0 * u + p1);
    case FR_K_MULTIQUADRIC:  // :1049 (hypot(||x-y||^2, c), as written)
        return hypot(s, p0);
    case FR_K_RATIONALQUADRATIC:  // :1121-1122
        return pow(1.0 + div_uniform(s, 2.0 * p0 * p1 * p1), -p0);
    default: return __builtin_nan("");
    }
}

__device__ __forceinline__ double leaf_eval(const fr_kernel_op& op, double s, double u) { return leaf_eval_k<-1>(op, s, u); }

__device__ __forceinline__ double kprog_eval(const fr_kprog& p, double s, double u)
{
    if (p.nops == 1) return leaf_eval(p.ops[0], s, u);
    double st[8];
    int sp = 0;
    for (int i = 0; i < p.nops; ++i) {
        const int k = p.ops[i].kind;
        if (k == FR_K_SUM) {  // kernel.rs:160
            st[sp - 2] = st[sp - 2] + st[sp - 1];
            --sp;
        } else if (k == FR_K_PROD) {  // :249
            st[sp - 2] = st[sp - 2] * st[sp - 1];
            --sp;
        } else {
            st[sp++] = leaf_eval(p.ops[i], s, u);
        }
    }
    return st[0];
}

// LEAF >= 0: the program is that single leaf
template <int LEAF>
__device__ __forceinline__ double kprog_eval_k(const fr_kprog& p, double s, double u)
{
    if constexpr (LEAF >= 0)
        return leaf_eval_k<LEAF>(p.ops[0], s, u);
    else
        return kprog_eval(p, s, u);
}

}  // namespace fr

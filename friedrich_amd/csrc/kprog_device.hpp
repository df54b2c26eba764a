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

__device__ __forceinline__ double leaf_eval(const fr_kernel_op& op, double s, double u)
{
#pragma clang fp contract(off)
    const double p0 = op.params[0], p1 = op.params[1], p2 = op.params[2];
    switch (op.kind) {
    case FR_K_LINEAR:  // kernel.rs:381
        return u + p0;
    case FR_K_POLYNOMIAL:  // :456
        return pow(p0 * u + p1, p2);
    case FR_K_SQUAREDEXP: {  // :556-560
        const double x = -s / (2.0 * p0 * p0);
        return fabs(p1) * exp(x);
    }
    case FR_K_EXPONENTIAL: {  // :661-665
        const double x = -sqrt(s) / (2.0 * p0 * p0);
        return fabs(p1) * exp(x);
    }
    case FR_K_MATERN1: {  // :766-771
        const double x = sqrt(3.0) * sqrt(s) / fabs(p0);
        return fabs(p1) * (1.0 + x) * exp(-x);
    }
    case FR_K_MATERN2: {  // :873-878
        const double l = fabs(p0);
        const double dist = sqrt(s);
        const double x = sqrt(5.0) * dist / l;
        return fabs(p1) * (1.0 + x + (5.0 * dist * dist) / (3.0 * l * l)) * exp(-x);
    }
    case FR_K_HYPERTAN:  // :976
        return tanh(p0 * u + p1);
    case FR_K_MULTIQUADRIC:  // :1049 (hypot(||x-y||^2, c), as written)
        return hypot(s, p0);
    case FR_K_RATIONALQUADRATIC:  // :1121-1122
        return pow(1.0 + s / (2.0 * p0 * p1 * p1), -p0);
    default: return __builtin_nan("");
    }
}

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

}  // namespace fr

/*
 * friedrich_oracle.c -- CPU restatement of friedrich 0.5.1's hot path (see friedrich_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into, imported by or called from the product path.
 * PARITY UNPINNED (no reference build, no reference golden vectors; see header).
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 * "[nalgebra]" marks arithmetic that lives in the third-party crate nalgebra 0.31.4 (Cargo.toml:23),
 * restated from its published algorithm -- loop order, non-fused mul/add and
 * true division are kept so the rounding behaviour is the reference's.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 */
#include "friedrich_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define AT(M, ld, r, c) ((M)[(int64_t)(r) + (int64_t)(c) * (int64_t)(ld)])

/* ------------------------------------------------------------------------------------------ */
/* host threads (NOT part of the reference, which is single-threaded)                          */
/* Loops whose iterations are independent in the reference -- the right-hand-side columns of a solve, the entries of a
 * Gram matrix, the rows of an axpy -- may be dealt to host threads so that full-size BASELINE configurations can be
 * checked in seconds.  Every element still sees exactly the reference's sequence of operations, so results are
 * bit-identical to the single-thread restatement (tests/test_oracle.py asserts equality, not closeness). */
static int g_threads = 1;

void fro_set_threads(int n)
{
    if (n <= 0) {
        long c = sysconf(_SC_NPROCESSORS_ONLN);
        n = (int)(c > 0 ? c : 1);
    }
    if (n > 64) n = 64;
    g_threads = n;
}

int fro_get_threads(void) { return g_threads; }

typedef void (*par_body)(int64_t lo, int64_t hi, void* arg);
typedef struct {
    par_body fn;
    void* arg;
    int64_t lo, hi;
} par_task;

static void* par_tramp(void* q)
{
    par_task* t = (par_task*)q;
    t->fn(t->lo, t->hi, t->arg);
    return NULL;
}

/* fn over [0, n) cut into contiguous ranges, one per thread (the calling thread takes the first) */
static void par_for(int64_t n, par_body fn, void* arg)
{
    int T = g_threads;
    if (T > n) T = (int)(n > 0 ? n : 1);
    if (T <= 1) {
        fn(0, n, arg);
        return;
    }
    pthread_t th[64];
    par_task task[64];
    for (int t = 0; t < T; ++t) {
        task[t].fn = fn;
        task[t].arg = arg;
        task[t].lo = n * t / T;
        task[t].hi = n * (t + 1) / T;
    }
    int started[64] = {0};
    for (int t = 1; t < T; ++t) started[t] = pthread_create(&th[t], NULL, par_tramp, &task[t]) == 0;
    par_tramp(&task[0]);
    for (int t = 1; t < T; ++t) {
        if (started[t])
            pthread_join(th[t], NULL);
        else
            par_tramp(&task[t]);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* small nalgebra primitives                                                                   */

/* f64::powi -> compiler-rt __powidf2 (square-and-multiply) */
static double powi_(double a, int b)
{
    const int recip = b < 0;
    double r = 1.0;
    while (1) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0 / r : r;
}

static double signum_(double x)
{ /* f64::signum: 1.0 for +0.0 and positives, -1.0 for -0.0 and negatives, NaN for NaN */
    if (isnan(x)) return x;
    return signbit(x) ? -1.0 : 1.0;
}

/* [nalgebra] Matrix::dot on column vectors (base/blas.rs dotx): 8 accumulators over blocks of 8 rows,
 * combined as (0+4)+(1+5)+(2+6)+(3+7), then the sequential tail. */
static double dot8(const double* a, const double* b, int64_t n)
{
    double res = 0.0;
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, acc4 = 0, acc5 = 0, acc6 = 0, acc7 = 0;
    int64_t i = 0;
    while (n - i >= 8) {
        acc0 += a[i + 0] * b[i + 0];
        acc1 += a[i + 1] * b[i + 1];
        acc2 += a[i + 2] * b[i + 2];
        acc3 += a[i + 3] * b[i + 3];
        acc4 += a[i + 4] * b[i + 4];
        acc5 += a[i + 5] * b[i + 5];
        acc6 += a[i + 6] * b[i + 6];
        acc7 += a[i + 7] * b[i + 7];
        i += 8;
    }
    res += acc0 + acc4;
    res += acc1 + acc5;
    res += acc2 + acc6;
    res += acc3 + acc7;
    for (; i < n; ++i) res += a[i] * b[i];
    return res;
}

/* [nalgebra] axpy with b = 1: y[i] = a*x[i] + y[i], mul then add (not fused). */
static void axpy1(double a, const double* x, double* y, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) y[i] = a * x[i] + y[i];
}

/* [nalgebra] norm_squared of a contiguous column: sequential sum of squares. */
static double norm_squared_col(const double* x, int64_t n)
{
    double res = 0.0;
    for (int64_t i = 0; i < n; ++i) res += x[i] * x[i];
    return res;
}

/* (x1 - x2).norm_squared() on 1 x d row views: the difference is materialised, then a sequential sum of
 * squares in storage order (kernel.rs:558, 571, 1049, 1121). */
static double row_dist2(const double* x1, int64_t inc1, const double* x2, int64_t inc2, int64_t d)
{
    double res = 0.0;
    for (int64_t c = 0; c < d; ++c) {
        const double diff = x1[c * inc1] - x2[c * inc2];
        res += diff * diff;
    }
    return res;
}

/* x1.dot(x2) on 1 x d row views: nalgebra's dotx unrolls over rows within a column; a row vector has one
 * row per column, so the sum is sequential over columns (kernel.rs:381, 456, 976). */
static double row_dot(const double* x1, int64_t inc1, const double* x2, int64_t inc2, int64_t d)
{
    double res = 0.0;
    for (int64_t c = 0; c < d; ++c) res += x1[c * inc1] * x2[c * inc2];
    return res;
}

/* ------------------------------------------------------------------------------------------ */
/* kernel programs                                                                             */

static int is_leaf(int kind) { return kind >= FRO_K_LINEAR && kind <= FRO_K_RATIONALQUADRATIC; }

/* declared Kernel::nb_parameters per leaf (Multiquadric returns 2, kernel.rs:1039-1042) */
static int leaf_nb_parameters(int kind)
{
    switch (kind) {
    case FRO_K_LINEAR: return 1;            /* :371-374 */
    case FRO_K_POLYNOMIAL: return 3;        /* :446-449 */
    case FRO_K_SQUAREDEXP: return 2;        /* :539-542 */
    case FRO_K_EXPONENTIAL: return 2;       /* :644-647 */
    case FRO_K_MATERN1: return 2;           /* :749-752 */
    case FRO_K_MATERN2: return 2;           /* :856-859 */
    case FRO_K_HYPERTAN: return 2;          /* :966-969 */
    case FRO_K_MULTIQUADRIC: return 2;      /* :1039-1042 (sic) */
    case FRO_K_RATIONALQUADRATIC: return 2; /* :1111-1114 */
    default: return -1;
    }
}

/* length of get_parameters()/gradient() per leaf (Multiquadric: 1, kernel.rs:1057-1064) */
static int leaf_nb_values(int kind)
{
    if (kind == FRO_K_MULTIQUADRIC) return 1;
    return leaf_nb_parameters(kind);
}

int fro_kprog_validate(const fro_kprog* p)
{
    if (!p || p->nops < 1 || p->nops > FRO_KPROG_MAX_OPS) return -1;
    int depth = 0;
    for (int i = 0; i < p->nops; ++i) {
        const int k = p->ops[i].kind;
        if (is_leaf(k)) {
            if (p->ops[i].nparams != leaf_nb_values(k)) return -1;
            depth += 1;
        } else if (k == FRO_K_SUM || k == FRO_K_PROD) {
            if (depth < 2) return -1;
            depth -= 1;
        } else {
            return -1;
        }
    }
    return depth == 1 ? 0 : -1;
}

int fro_kprog_nb_parameters(const fro_kprog* p)
{ /* Sum/Prod: k1.nb_parameters() + k2.nb_parameters() (kernel.rs:145-148, 234-237) */
    int n = 0;
    for (int i = 0; i < p->nops; ++i)
        if (is_leaf(p->ops[i].kind)) n += leaf_nb_parameters(p->ops[i].kind);
    return n;
}

int fro_kprog_nb_gradients(const fro_kprog* p)
{
    int n = 0;
    for (int i = 0; i < p->nops; ++i)
        if (is_leaf(p->ops[i].kind)) n += leaf_nb_values(p->ops[i].kind);
    return n;
}

/* first op index of the subtree whose root is op i */
static int subtree_start(const fro_kprog* p, int i)
{
    if (is_leaf(p->ops[i].kind)) return i;
    const int r_start = subtree_start(p, i - 1);
    return subtree_start(p, r_start - 1);
}

static int scalable_at(const fro_kprog* p, int i)
{
    const int k = p->ops[i].kind;
    if (is_leaf(k))
        return k == FRO_K_SQUAREDEXP || k == FRO_K_EXPONENTIAL || k == FRO_K_MATERN1 ||
               k == FRO_K_MATERN2; /* :544-547, 649-652, 754-757, 861-864; default false :33-36 */
    const int r = i - 1;
    const int l = subtree_start(p, r) - 1;
    if (k == FRO_K_SUM) return scalable_at(p, l) && scalable_at(p, r); /* :150-153 */
    return scalable_at(p, l) || scalable_at(p, r);                     /* :239-242 */
}

int fro_kprog_is_scalable(const fro_kprog* p) { return scalable_at(p, p->nops - 1); }

static int rescale_at(fro_kprog* p, int i, double scale)
{
    const int k = p->ops[i].kind;
    if (is_leaf(k)) {
        if (!scalable_at(p, i)) return -1; /* "You tried to rescale a Kernel that is not Scalable!" :52 */
        p->ops[i].params[1] *= scale;      /* ampl *= scale :578-581, 683-686, 790-793, 902-905 */
        return 0;
    }
    const int r = i - 1;
    const int l = subtree_start(p, r) - 1;
    if (k == FRO_K_SUM) { /* :174-178 */
        if (rescale_at(p, l, scale)) return -1;
        return rescale_at(p, r, scale);
    }
    /* Prod :264-274 */
    if (scalable_at(p, l)) return rescale_at(p, l, scale);
    return rescale_at(p, r, scale);
}

int fro_kprog_rescale(fro_kprog* p, double scale) { return rescale_at(p, p->nops - 1, scale); }

int fro_kprog_get_parameters(const fro_kprog* p, double* out)
{ /* leaves in k1-then-k2 order (:180-186, 276-282) */
    int n = 0;
    for (int i = 0; i < p->nops; ++i)
        if (is_leaf(p->ops[i].kind))
            for (int q = 0; q < p->ops[i].nparams; ++q) out[n++] = p->ops[i].params[q];
    return n;
}

int fro_kprog_set_parameters(fro_kprog* p, const double* params, int n)
{
    /* Sum/Prod slice by nb_parameters() (:188-192, 284-288); each leaf reads parameters[0..] of its
     * slice, except Multiquadric which reads parameters[1] (:1066-1069) -- out of bounds (a panic in the
     * reference, -1 here) when its slice holds a single value. */
    int off = 0;
    for (int i = 0; i < p->nops; ++i) {
        const int k = p->ops[i].kind;
        if (!is_leaf(k)) continue;
        const int width = leaf_nb_parameters(k);
        int avail = n - off;
        if (avail > width) avail = width;
        /* the last leaf of a slice chain receives "the rest" */
        int is_last_leaf = 1;
        for (int j = i + 1; j < p->nops; ++j)
            if (is_leaf(p->ops[j].kind)) is_last_leaf = 0;
        if (is_last_leaf) avail = n - off;
        if (k == FRO_K_MULTIQUADRIC) {
            if (avail < 2) return -1;
            p->ops[i].params[0] = params[off + 1];
        } else {
            if (avail < width) return -1;
            for (int q = 0; q < width; ++q) p->ops[i].params[q] = params[off + q];
        }
        off += width;
    }
    return 0;
}

/* one leaf's kernel value.  s = ||x1-x2||^2 and u = x1.x2 are recomputed per leaf exactly as each
 * reference body does (no sharing between leaves). */
static double leaf_kernel(const fro_kernel_op* op, const double* x1, int64_t inc1, const double* x2, int64_t inc2,
                          int64_t d)
{
    const double* P = op->params;
    switch (op->kind) {
    case FRO_K_LINEAR: /* kernel.rs:376-382 */
        return row_dot(x1, inc1, x2, inc2, d) + P[0];
    case FRO_K_POLYNOMIAL: /* :451-457 */
        return pow(P[0] * row_dot(x1, inc1, x2, inc2, d) + P[1], P[2]);
    case FRO_K_SQUAREDEXP: { /* :550-561 */
        const double ampl = fabs(P[1]);
        const double distance_squared = row_dist2(x1, inc1, x2, inc2, d);
        const double x = -distance_squared / (2.0 * P[0] * P[0]);
        return ampl * exp(x);
    }
    case FRO_K_EXPONENTIAL: { /* :655-666 */
        const double ampl = fabs(P[1]);
        const double distance = sqrt(row_dist2(x1, inc1, x2, inc2, d));
        const double x = -distance / (2.0 * P[0] * P[0]);
        return ampl * exp(x);
    }
    case FRO_K_MATERN1: { /* :760-772 */
        const double ampl = fabs(P[1]);
        const double l = fabs(P[0]);
        const double distance = sqrt(row_dist2(x1, inc1, x2, inc2, d));
        const double x = sqrt(3.0) * distance / l;
        return ampl * (1.0 + x) * exp(-x);
    }
    case FRO_K_MATERN2: { /* :867-879 */
        const double ampl = fabs(P[1]);
        const double l = fabs(P[0]);
        const double distance = sqrt(row_dist2(x1, inc1, x2, inc2, d));
        const double x = sqrt(5.0) * distance / l;
        return ampl * (1.0 + x + (5.0 * distance * distance) / (3.0 * l * l)) * exp(-x);
    }
    case FRO_K_HYPERTAN: /* :971-977 */
        return tanh(P[0] * row_dot(x1, inc1, x2, inc2, d) + P[1]);
    case FRO_K_MULTIQUADRIC: /* :1044-1050 : hypot(||x-y||^2, c) as written */
        return hypot(row_dist2(x1, inc1, x2, inc2, d), P[0]);
    case FRO_K_RATIONALQUADRATIC: { /* :1116-1123 */
        const double distance_squared = row_dist2(x1, inc1, x2, inc2, d);
        return pow(1.0 + distance_squared / (2.0 * P[0] * P[1] * P[1]), -P[0]);
    }
    default: return NAN;
    }
}

/* one leaf's gradient; returns the number of values written */
static int leaf_gradient(const fro_kernel_op* op, const double* x1, int64_t inc1, const double* x2, int64_t inc2,
                         int64_t d, double* g)
{
    const double* P = op->params;
    switch (op->kind) {
    case FRO_K_LINEAR: /* :384-391 */
        g[0] = 1.0;
        return 1;
    case FRO_K_POLYNOMIAL: { /* :459-472 */
        const double x = row_dot(x1, inc1, x2, inc2, d);
        const double inner_term = P[0] * x + P[1];
        const double grad_c = P[2] * pow(inner_term, P[2] - 1.0);
        const double grad_alpha = x * grad_c;
        const double grad_d = log(inner_term) * pow(inner_term, P[2]);
        g[0] = grad_alpha;
        g[1] = grad_c;
        g[2] = grad_d;
        return 3;
    }
    case FRO_K_SQUAREDEXP: { /* :563-576 */
        const double ampl = fabs(P[1]);
        const double distance_squared = row_dist2(x1, inc1, x2, inc2, d);
        const double exponential = exp(-distance_squared / (2.0 * P[0] * P[0]));
        g[0] = (distance_squared * ampl * exponential) / powi_(P[0], 3);
        g[1] = signum_(P[1]) * exponential;
        return 2;
    }
    case FRO_K_EXPONENTIAL: { /* :668-681 */
        const double ampl = fabs(P[1]);
        const double distance = sqrt(row_dist2(x1, inc1, x2, inc2, d));
        const double exponential = exp(-distance / (2.0 * P[0] * P[0]));
        g[0] = (distance * ampl * exponential) / powi_(P[0], 3);
        g[1] = signum_(P[1]) * exponential;
        return 2;
    }
    case FRO_K_MATERN1: { /* :774-788 */
        const double ampl = fabs(P[1]);
        const double l = fabs(P[0]);
        const double distance = sqrt(row_dist2(x1, inc1, x2, inc2, d));
        const double x = sqrt(3.0) * distance / l;
        g[0] = (3.0 * ampl * powi_(distance, 2) * exp(-x)) / (powi_(P[0], 3));
        g[1] = signum_(P[1]) * (1.0 + x) * exp(-x);
        return 2;
    }
    case FRO_K_MATERN2: { /* :881-900 (x uses the signed ls, :891; grad_ls as written) */
        const double ampl = fabs(P[1]);
        const double l = fabs(P[0]);
        const double distance = sqrt(row_dist2(x1, inc1, x2, inc2, d));
        const double x = sqrt(5.0) * distance / P[0];
        g[0] = signum_(P[0]) * ampl *
               ((2.0 * l / 3.0 + 1.0) + distance * sqrt(5.0) * ((powi_(l, 2) / 3.0 + l + 1.0) / powi_(l, 2))) *
               exp(-x);
        g[1] = signum_(P[1]) * (1.0 + x + (5.0 * distance * distance) / (3.0 * l * l)) * exp(-x);
        return 2;
    }
    case FRO_K_HYPERTAN: { /* :979-989 */
        const double x = row_dot(x1, inc1, x2, inc2, d);
        const double grad_c = 1.0 / powi_(cosh(P[0] * x + P[1]), 2);
        g[0] = x * grad_c;
        g[1] = grad_c;
        return 2;
    }
    case FRO_K_MULTIQUADRIC: /* :1052-1059 */
        g[0] = P[0] / hypot(sqrt(row_dist2(x1, inc1, x2, inc2, d)), P[0]);
        return 1;
    case FRO_K_RATIONALQUADRATIC: { /* :1125-1145 */
        const double alpha = P[0];
        const double l = fabs(P[1]);
        const double ds = row_dist2(x1, inc1, x2, inc2, d);
        const double grad_alpha =
            pow((ds + 2.0 * powi_(l, 2) * alpha) / (powi_(l, 2) * alpha), -alpha) *
            (pow(2.0, alpha) * (1.0 - log((ds + 2.0 * powi_(l, 2) * alpha) / (2.0 * powi_(l, 2) * alpha))) -
             (powi_(l, 2) * pow(2.0, alpha + 1.0) * alpha) / (ds + 2.0 * powi_(l, 2) * alpha));
        const double grad_ls = ds * pow(ds / (2.0 * alpha * l * l) + 1.0, -alpha - 1.0) / powi_(P[1], 3);
        g[0] = grad_alpha;
        g[1] = grad_ls;
        return 2;
    }
    default: return 0;
    }
}

double fro_kernel(const fro_kprog* p, const double* x1, int64_t inc1, const double* x2, int64_t inc2, int64_t d)
{
    double stack[FRO_KPROG_MAX_OPS];
    int sp = 0;
    for (int i = 0; i < p->nops; ++i) {
        const int k = p->ops[i].kind;
        if (is_leaf(k)) {
            stack[sp++] = leaf_kernel(&p->ops[i], x1, inc1, x2, inc2, d);
        } else {
            const double b = stack[--sp];
            const double a = stack[--sp];
            stack[sp++] = (k == FRO_K_SUM) ? a + b : a * b; /* :160 / :249 */
        }
    }
    return stack[0];
}

int fro_kernel_gradient(const fro_kprog* p, const double* x1, int64_t inc1, const double* x2, int64_t inc2,
                        int64_t d, double* out)
{
    /* stack of (value, gradient slice); leaves write their gradients in program order, which is the
     * k1-then-k2 concatenation order of :168-171 and :261. */
    double val[FRO_KPROG_MAX_OPS];
    int gstart[FRO_KPROG_MAX_OPS], glen[FRO_KPROG_MAX_OPS];
    int sp = 0, ng = 0;
    for (int i = 0; i < p->nops; ++i) {
        const int k = p->ops[i].kind;
        if (is_leaf(k)) {
            val[sp] = leaf_kernel(&p->ops[i], x1, inc1, x2, inc2, d);
            gstart[sp] = ng;
            glen[sp] = leaf_gradient(&p->ops[i], x1, inc1, x2, inc2, d, out + ng);
            ng += glen[sp];
            ++sp;
        } else {
            const int b = sp - 1, a = sp - 2;
            if (k == FRO_K_PROD) { /* :252-262 : g1*k2 then g2*k1 */
                for (int q = 0; q < glen[a]; ++q) out[gstart[a] + q] = out[gstart[a] + q] * val[b];
                for (int q = 0; q < glen[b]; ++q) out[gstart[b] + q] = out[gstart[b] + q] * val[a];
                val[a] = val[a] * val[b];
            } else {
                val[a] = val[a] + val[b];
            }
            glen[a] += glen[b];
            sp -= 1;
        }
    }
    return ng;
}

/* ------------------------------------------------------------------------------------------ */
/* heuristics                                                                                  */

/* kernel.rs:94-113 */
double fro_fit_bandwidth_mean(const double* X, int64_t n, int64_t ldx, int64_t d)
{
    double sum_distances = 0.0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i + 1; j < n; ++j) {
            const double distance = sqrt(row_dist2(X + i, ldx, X + j, ldx, d));
            sum_distances += distance;
        }
    const double nb_distances = (double)((n * n - n) / 2);
    return sum_distances / nb_distances;
}

/* [nalgebra 0.31] Matrix::variance(): fold (sum x^2, sum x); E[x^2] - E[x]^2 (kernel.rs:116-119, builder.rs:73) */
double fro_variance(const double* y, int64_t n)
{
    if (n == 0) return 0.0;
    double s2 = 0.0, s1 = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        s2 = s2 + y[i] * y[i];
        s1 = s1 + y[i];
    }
    const double denom = 1.0 / (double)n;
    const double vd = s1 * denom;
    return s2 * denom - vd * vd;
}

/* [nalgebra] Matrix::mean(): sum / n (prior.rs:97) */
double fro_mean(const double* y, int64_t n)
{
    if (n == 0) return 0.0;
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += y[i];
    return s / (double)n;
}

/* heuristic_fit: kernel.rs:594-600, 699-705, 806-812, 918-924; Sum/Prod :194-200, 290-296 */
void fro_heuristic_fit(fro_kprog* p, const double* X, int64_t n, int64_t ldx, int64_t d, const double* y)
{
    for (int i = 0; i < p->nops; ++i) {
        const int k = p->ops[i].kind;
        if (k == FRO_K_SQUAREDEXP || k == FRO_K_EXPONENTIAL || k == FRO_K_MATERN1 || k == FRO_K_MATERN2) {
            p->ops[i].params[0] = fro_fit_bandwidth_mean(X, n, ldx, d);
            p->ops[i].params[1] = fro_variance(y, n);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* src/algebra/mod.rs                                                                          */

/* algebra/mod.rs:41-54 : DMatrix::from_fn fills column by column */
typedef struct {
    const fro_kprog* p;
    const double *A, *B;
    int64_t n1, lda, ldb, d, ldo;
    double* out;
} gram_arg;

static void gram_cols(int64_t c0, int64_t c1, void* q)
{
    const gram_arg* g = (const gram_arg*)q;
    for (int64_t c = c0; c < c1; ++c)
        for (int64_t r = 0; r < g->n1; ++r)
            AT(g->out, g->ldo, r, c) = fro_kernel(g->p, g->A + r, g->lda, g->B + c, g->ldb, g->d);
}

void fro_make_covariance_matrix(const fro_kprog* p, const double* A, int64_t n1, int64_t lda, const double* B,
                                int64_t n2, int64_t ldb, int64_t d, double* out, int64_t ldo)
{
    gram_arg g = {p, A, B, n1, lda, ldb, d, ldo, out};
    par_for(n2, gram_cols, &g); /* entries are independent (algebra/mod.rs:46-51) */
}

/* [nalgebra] Cholesky::new_internal (linalg/cholesky.rs) */
int fro_cholesky(double* A, int64_t n, int64_t lda, int has_sub, double sub, int64_t* n_subst, int64_t* subst_idx)
{
    int64_t ns = 0;
    for (int64_t j = 0; j < n; ++j) {
        for (int64_t k = 0; k < j; ++k) {
            const double factor = -AT(A, lda, j, k);
            axpy1(factor, &AT(A, lda, j, k), &AT(A, lda, j, j), n - j);
        }
        const double diag = AT(A, lda, j, j);
        double denom;
        int ok = 0;
        /* sqrt_denom(v): None if v == 0, else try_sqrt (Some iff v >= 0; NaN -> None) */
        if (diag != 0.0 && diag >= 0.0) {
            denom = sqrt(diag);
            ok = 1;
        } else if (has_sub && sub != 0.0 && sub >= 0.0) {
            denom = sqrt(sub);
            ok = 1;
            if (subst_idx) subst_idx[ns] = j;
            ++ns;
        }
        if (!ok) {
            if (n_subst) *n_subst = ns;
            return (int)(1 + j);
        }
        AT(A, lda, j, j) = denom;
        for (int64_t i = j + 1; i < n; ++i) AT(A, lda, i, j) /= denom;
    }
    if (n_subst) *n_subst = ns;
    return 0;
}

/* ---- the same factorisation, scheduled for several host threads and for the leading `ncols` columns only ------------
 * Element (i, j) of the reference's factor is
 *     ((..((a_ij - l_j0 l_i0) - l_j1 l_i1) ..) - l_j,j-1 l_i,j-1) / l_jj        (fro_cholesky above: k ascending, mul then add)
 * and depends only on columns < j.  Any schedule that applies the updates to an element in ascending k with the same
 * unfused multiply-add therefore reproduces fro_cholesky BIT FOR BIT; this one is right-looking in panels of CB columns
 * (the panel's own columns left-looking, then the panel applied to the later columns < ncols), rows dealt to threads in
 * blocks.  Columns >= ncols are never touched: their leading-column dependencies are one-way.  tests/test_oracle.py
 * asserts array_equal against fro_cholesky (substitution list and failure column included). */
#define CB 64   /* panel width */
#define RB 256  /* row block */

typedef struct {
    double* A;
    int64_t n, lda, ncols;
    int has_sub;
    double sub;
    int64_t* subst_idx;
    /* shared state, written by thread 0 between barriers */
    int64_t ns;
    int fail; /* 1 + failing column, 0 = none */
    double denom;
    pthread_barrier_t bar;
    int T;
} cholmt;

typedef struct {
    cholmt* c;
    int t;
} cholmt_thread;

static void cholmt_update_block(double* A, int64_t lda, int64_t i0, int64_t i1, int64_t j, int64_t k0, int64_t k1,
                                const double* f)
{
    /* y[i] = f[k] * x_k[i] + y[i] for k = k0..k1-1 in order, rows i0..i1-1 of column j */
    double* y = &AT(A, lda, i0, j);
    const int64_t len = i1 - i0;
    int64_t k = k0;
    for (; k + 4 <= k1; k += 4) {
        const double f0 = f[k - k0], f1 = f[k + 1 - k0], f2 = f[k + 2 - k0], f3 = f[k + 3 - k0];
        const double *x0 = &AT(A, lda, i0, k), *x1 = &AT(A, lda, i0, k + 1), *x2 = &AT(A, lda, i0, k + 2),
                     *x3 = &AT(A, lda, i0, k + 3);
        for (int64_t i = 0; i < len; ++i) {
            double t = y[i];
            t = f0 * x0[i] + t;
            t = f1 * x1[i] + t;
            t = f2 * x2[i] + t;
            t = f3 * x3[i] + t;
            y[i] = t;
        }
    }
    for (; k < k1; ++k) axpy1(f[k - k0], &AT(A, lda, i0, k), y, len);
}

static void* cholmt_worker(void* q)
{
    cholmt_thread* me = (cholmt_thread*)q;
    cholmt* c = me->c;
    double* A = c->A;
    const int64_t n = c->n, lda = c->lda, ncols = c->ncols;
    const int T = c->T, t = me->t;
    const int64_t nrb = (n + RB - 1) / RB;
    double f[CB];
    for (int64_t k0 = 0; k0 < ncols; k0 += CB) {
        const int64_t k1 = k0 + CB < ncols ? k0 + CB : ncols;
        /* the panel's own columns */
        for (int64_t j = k0; j < k1; ++j) {
            for (int64_t k = k0; k < j; ++k) f[k - k0] = -AT(A, lda, j, k);
            for (int64_t rb = j / RB + ((t - (j / RB) % T + T) % T); rb < nrb; rb += T) { /* blocks rb == t (mod T) */
                const int64_t i0 = rb * RB > j ? rb * RB : j, i1 = (rb + 1) * RB < n ? (rb + 1) * RB : n;
                if (i0 < i1 && j > k0) cholmt_update_block(A, lda, i0, i1, j, k0, j, f);
            }
            pthread_barrier_wait(&c->bar);
            if (t == 0 && !c->fail) {
                const double diag = AT(A, lda, j, j);
                if (diag != 0.0 && diag >= 0.0) {
                    c->denom = sqrt(diag);
                } else if (c->has_sub && c->sub != 0.0 && c->sub >= 0.0) {
                    c->denom = sqrt(c->sub);
                    if (c->subst_idx) c->subst_idx[c->ns] = j;
                    ++c->ns;
                } else {
                    c->fail = (int)(1 + j);
                }
                if (!c->fail) AT(A, lda, j, j) = c->denom;
            }
            pthread_barrier_wait(&c->bar);
            if (c->fail) return NULL;
            const double denom = c->denom;
            for (int64_t rb = j / RB + ((t - (j / RB) % T + T) % T); rb < nrb; rb += T) {
                const int64_t i0 = rb * RB > j + 1 ? rb * RB : j + 1, i1 = (rb + 1) * RB < n ? (rb + 1) * RB : n;
                for (int64_t i = i0; i < i1; ++i) AT(A, lda, i, j) /= denom;
            }
            pthread_barrier_wait(&c->bar);
        }
        /* the panel applied to the later leading columns */
        if (k1 < ncols) {
            for (int64_t rb = k1 / RB + ((t - (k1 / RB) % T + T) % T); rb < nrb; rb += T) {
                const int64_t r0 = rb * RB, r1 = (rb + 1) * RB < n ? (rb + 1) * RB : n;
                const int64_t jmax = ncols < r1 ? ncols : r1;
                for (int64_t j = k1; j < jmax; ++j) {
                    const int64_t i0 = r0 > j ? r0 : j;
                    for (int64_t k = k0; k < k1; ++k) f[k - k0] = -AT(A, lda, j, k);
                    cholmt_update_block(A, lda, i0, r1, j, k0, k1, f);
                }
            }
            pthread_barrier_wait(&c->bar);
        }
    }
    return NULL;
}

int fro_cholesky_cols_mt(double* A, int64_t n, int64_t lda, int64_t ncols, int has_sub, double sub, int64_t* n_subst,
                         int64_t* subst_idx)
{
    if (ncols > n) ncols = n;
    cholmt c;
    memset(&c, 0, sizeof(c));
    c.A = A;
    c.n = n;
    c.lda = lda;
    c.ncols = ncols;
    c.has_sub = has_sub;
    c.sub = sub;
    c.subst_idx = subst_idx;
    int T = g_threads;
    if (T > 64) T = 64;
    if ((int64_t)T > (n + RB - 1) / RB) T = (int)((n + RB - 1) / RB);
    if (T < 1) T = 1;
    c.T = T;
    if (n_subst) *n_subst = 0;
    if (ncols <= 0) return 0;
    pthread_barrier_init(&c.bar, NULL, (unsigned)T);
    pthread_t th[64];
    cholmt_thread arg[64];
    for (int t = 0; t < T; ++t) {
        arg[t].c = &c;
        arg[t].t = t;
    }
    for (int t = 1; t < T; ++t)
        if (pthread_create(&th[t], NULL, cholmt_worker, &arg[t]) != 0) abort(); /* the barrier counts T participants */
    cholmt_worker(&arg[0]);
    for (int t = 1; t < T; ++t) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&c.bar);
    if (n_subst) *n_subst = c.ns;
    return c.fail;
}

/* algebra/mod.rs:59-92 restricted to the leading ncols columns of the factor (out: n x ncols), threaded as above */
typedef struct {
    const fro_kprog* p;
    const double* X;
    int64_t n, ldx, d, ldo;
    double noise;
    double* out;
} gramsym_arg;

static void gramsym_cols(int64_t c0, int64_t c1, void* q)
{
    const gramsym_arg* g = (const gramsym_arg*)q;
    for (int64_t col = c0; col < c1; ++col) { /* :70-79 */
        for (int64_t row = 0; row < col; ++row) AT(g->out, g->ldo, row, col) = NAN; /* :67 */
        for (int64_t row = col; row < g->n; ++row)
            AT(g->out, g->ldo, row, col) = fro_kernel(g->p, g->X + col, g->ldx, g->X + row, g->ldx, g->d);
        AT(g->out, g->ldo, col, col) += g->noise * g->noise;
    }
}

int fro_make_cholesky_cov_matrix_cols_mt(const fro_kprog* p, const double* X, int64_t n, int64_t ldx, int64_t d,
                                         double noise, int has_eps, double eps, int64_t ncols, double* out, int64_t ldo,
                                         int64_t* n_subst, int64_t* subst_idx)
{
    if (ncols > n) ncols = n;
    gramsym_arg g = {p, X, n, ldx, d, ldo, noise, out};
    par_for(ncols, gramsym_cols, &g);
    return fro_cholesky_cols_mt(out, n, ldo, ncols, has_eps, eps, n_subst, subst_idx);
}

/* algebra/mod.rs:59-92 */
int fro_make_cholesky_cov_matrix(const fro_kprog* p, const double* X, int64_t n, int64_t ldx, int64_t d,
                                 double noise, int has_eps, double eps, double* out, int64_t ldo,
                                 int64_t* n_subst, int64_t* subst_idx)
{
    for (int64_t c = 0; c < n; ++c) /* :67 */
        for (int64_t r = 0; r < n; ++r) AT(out, ldo, r, c) = NAN;
    for (int64_t col = 0; col < n; ++col) { /* :70-79 */
        for (int64_t row = col; row < n; ++row) AT(out, ldo, row, col) = fro_kernel(p, X + col, ldx, X + row, ldx, d);
        AT(out, ldo, col, col) += noise * noise;
    }
    return fro_cholesky(out, n, ldo, has_eps, eps, n_subst, subst_idx); /* :81-91 */
}

typedef struct {
    const double* L;
    int64_t n, ldl, ldb;
    double* B;
} solve_arg;

/* [nalgebra] solve_lower_triangular_mut: column-oriented forward substitution, one right-hand side at a time
 * (the right-hand sides are independent: par_for over them) */
static void solve_lower_cols(int64_t c0, int64_t c1, void* q)
{
    const solve_arg* a = (const solve_arg*)q;
    const double* L = a->L;
    const int64_t n = a->n, ldl = a->ldl;
    for (int64_t c = c0; c < c1; ++c) {
        double* b = a->B + c * a->ldb;
        for (int64_t i = 0; i < n; ++i) {
            const double coeff = b[i] / AT(L, ldl, i, i);
            b[i] = coeff;
            axpy1(-coeff, &AT(L, ldl, i + 1 < n ? i + 1 : i, i), b + i + 1, n - i - 1);
        }
    }
}

/* unchecked variant used by Cholesky::solve_mut */
static void solve_lower_unchecked(const double* L, int64_t n, int64_t ldl, double* B, int64_t m, int64_t ldb)
{
    solve_arg a = {L, n, ldl, ldb, B};
    par_for(m, solve_lower_cols, &a);
}

/* checked: None (here -1) iff a diagonal entry is exactly zero -- the check only depends on L, so it is hoisted */
int fro_solve_lower(const double* L, int64_t n, int64_t ldl, double* B, int64_t m, int64_t ldb)
{
    if (m > 0)
        for (int64_t i = 0; i < n; ++i)
            if (AT(L, ldl, i, i) == 0.0) return -1;
    solve_lower_unchecked(L, n, ldl, B, m, ldb);
    return 0;
}

static void ad_solve_cols(int64_t c0, int64_t c1, void* q)
{
    const solve_arg* a = (const solve_arg*)q;
    const double* L = a->L;
    const int64_t n = a->n, ldl = a->ldl;
    for (int64_t c = c0; c < c1; ++c) {
        double* b = a->B + c * a->ldb;
        for (int64_t i = n - 1; i >= 0; --i) {
            const double dot = (i + 1 < n) ? dot8(&AT(L, ldl, i + 1, i), b + i + 1, n - i - 1) : dot8(b, b, 0);
            b[i] = (b[i] - dot) / AT(L, ldl, i, i);
        }
    }
}

/* [nalgebra] ad_solve_lower_triangular_unchecked_mut: dot-oriented backward substitution with L^T */
void fro_ad_solve_lower(const double* L, int64_t n, int64_t ldl, double* B, int64_t m, int64_t ldb)
{
    solve_arg a = {L, n, ldl, ldb, B};
    par_for(m, ad_solve_cols, &a);
}

/* [nalgebra] Cholesky::solve_mut */
void fro_chol_solve(const double* L, int64_t n, int64_t ldl, double* B, int64_t m, int64_t ldb)
{
    solve_lower_unchecked(L, n, ldl, B, m, ldb);
    fro_ad_solve_lower(L, n, ldl, B, m, ldb);
}

/* [nalgebra] Cholesky::inverse: identity then solve_mut (optimizer.rs:32, 169) */
void fro_chol_inverse(const double* L, int64_t n, int64_t ldl, double* out, int64_t ldo)
{
    for (int64_t c = 0; c < n; ++c)
        for (int64_t r = 0; r < n; ++r) AT(out, ldo, r, c) = (r == c) ? 1.0 : 0.0;
    fro_chol_solve(L, n, ldl, out, n, ldo);
}

/* algebra/mod.rs:97-126 with [nalgebra] Cholesky::insert_column(j = n) (Appendix A.3) */
void fro_add_rows_cholesky_cov_matrix(const fro_kprog* p, double* L, int64_t ldl, const double* Xall, int64_t n_all,
                                      int64_t ldx, int64_t d, int64_t nb_new, double noise)
{
    const int64_t nb_old = n_all - nb_new; /* :102 */
    double* col = (double*)malloc(sizeof(double) * (size_t)(n_all > 0 ? n_all : 1));
    for (int64_t row_index = 0; row_index < nb_new; ++row_index) { /* :108 */
        const int64_t col_index = nb_old + row_index;              /* :111 */
        const double* row = Xall + col_index;
        for (int64_t t = 0; t <= col_index; ++t) col[t] = fro_kernel(p, Xall + t, ldx, row, ldx, d); /* :115-118 */
        col[col_index] += noise * noise;                                                               /* :121 */
        /* insert_column(col_index, col): the old factor occupies L[0..col_index, 0..col_index] */
        const double col_j = col[col_index];
        /* new_rowj_adjoint = top_left^-1 * col[..j] (checked forward solve; zero diagonal asserts in the
         * reference -- left as a division by zero here) */
        for (int64_t i = 0; i < col_index; ++i) {
            const double coeff = col[i] / AT(L, ldl, i, i);
            col[i] = coeff;
            axpy1(-coeff, &AT(L, ldl, i + 1 < col_index ? i + 1 : i, i), col + i + 1, col_index - i - 1);
        }
        for (int64_t c = 0; c < col_index; ++c) AT(L, ldl, col_index, c) = col[c]; /* adjoint_to row j */
        AT(L, ldl, col_index, col_index) = sqrt(col_j - norm_squared_col(col, col_index));
        for (int64_t r = 0; r < col_index; ++r) AT(L, ldl, r, col_index) = NAN; /* upper stays unspecified */
    }
    free(col);
}

/* algebra/mod.rs:129-155 */
void fro_make_gradient_covariance_matrices(const fro_kprog* p, const double* X, int64_t n, int64_t ldx, int64_t d,
                                           double* out)
{
    const int np = fro_kprog_nb_parameters(p);
    const int64_t nn = n * n;
    for (int64_t e = 0; e < (int64_t)np * nn; ++e) out[e] = NAN; /* :135-139 */
    double g[3 * FRO_KPROG_MAX_OPS];
    for (int64_t col = 0; col < n; ++col)
        for (int64_t row = col; row < n; ++row) {
            const int ng = fro_kernel_gradient(p, X + col, ldx, X + row, ldx, d, g);
            for (int q = 0; q < ng && q < np; ++q) { /* zip stops at the shorter (:146) */
                out[q * nn + row + col * n] = g[q];
                out[q * nn + col + row * n] = g[q];
            }
        }
}

/* ------------------------------------------------------------------------------------------ */
/* src/gaussian_process/mod.rs                                                                 */

/* [nalgebra] y.gemv_tr(alpha, A, x, beta) per output: alpha * dot(A[:,j], x) + beta * y[j] (A.4) */
static void gemv_tr(double alpha, const double* A, int64_t n, int64_t lda, int64_t m, const double* x, double beta,
                    double* y)
{
    for (int64_t j = 0; j < m; ++j) y[j] = alpha * dot8(A + j * lda, x, n) + beta * y[j];
}

/* mod.rs:196-220 */
double fro_likelihood(const fro_kprog* p, const double* L, int64_t n, int64_t ldl, const double* X, int64_t ldx,
                      int64_t d, const double* y, double noise)
{
    double* ol = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    memcpy(ol, y, sizeof(double) * (size_t)n);
    if (fro_solve_lower(L, n, ldl, ol, 1, n)) { /* "likelihood : solve failed" :203 */
        free(ol);
        return NAN;
    }
    const double data_fit = norm_squared_col(ol, n);
    free(ol);
    double complexity_penalty = 0.0;
    for (int64_t r = 0; r < n; ++r) /* :208-213 */
        complexity_penalty += log(fabs(fro_kernel(p, X + r, ldx, X + r, ldx, d) + noise * noise));
    const double normalization_constant = (double)n * log(2.0 * M_PI);
    return -(data_fit + complexity_penalty + normalization_constant) / 2.0;
}

/* mod.rs:226-244 */
void fro_predict(const fro_kprog* p, const double* L, int64_t n, int64_t ldl, const double* X, int64_t ldx, int64_t d,
                 const double* y, const double* Xq, int64_t m, int64_t ldq, const double* prior_q, double* mean)
{
    double* weights = (double*)malloc(sizeof(double) * (size_t)(n * m > 0 ? n * m : 1));
    fro_make_covariance_matrix(p, X, n, ldx, Xq, m, ldq, d, weights, n); /* :234 */
    fro_chol_solve(L, n, ldl, weights, m, n);                            /* :235 */
    for (int64_t i = 0; i < m; ++i) mean[i] = prior_q ? prior_q[i] : 0.0; /* :238 */
    gemv_tr(1.0, weights, n, n, m, y, 1.0, mean);                        /* :241 */
    free(weights);
}

/* mod.rs:248-273 */
int fro_predict_variance(const fro_kprog* p, const double* L, int64_t n, int64_t ldl, const double* X, int64_t ldx,
                         int64_t d, const double* Xq, int64_t m, int64_t ldq, double* var)
{
    double* kl = (double*)malloc(sizeof(double) * (size_t)(n * m > 0 ? n * m : 1));
    fro_make_covariance_matrix(p, X, n, ldx, Xq, m, ldq, d, kl, n); /* :256-257 */
    if (fro_solve_lower(L, n, ldl, kl, m, n)) {                      /* :260-263 */
        free(kl);
        return -1;
    }
    for (int64_t i = 0; i < m; ++i) { /* :266-270 */
        const double base_cov = fro_kernel(p, Xq + i, ldq, Xq + i, ldq, d);
        const double predicted_cov = norm_squared_col(kl + i * n, n);
        var[i] = base_cov - predicted_cov;
    }
    free(kl);
    return 0;
}

/* mod.rs:290-326 */
void fro_predict_mean_variance(const fro_kprog* p, const double* L, int64_t n, int64_t ldl, const double* X,
                               int64_t ldx, int64_t d, const double* y, const double* Xq, int64_t m, int64_t ldq,
                               const double* prior_q, double* mean, double* var)
{
    const size_t sz = sizeof(double) * (size_t)(n * m > 0 ? n * m : 1);
    double* cov_train_inputs = (double*)malloc(sz);
    double* weights = (double*)malloc(sz);
    fro_make_covariance_matrix(p, X, n, ldx, Xq, m, ldq, d, cov_train_inputs, n); /* :296-297 */
    memcpy(weights, cov_train_inputs, sz);
    fro_chol_solve(L, n, ldl, weights, m, n); /* :298 */
    for (int64_t i = 0; i < m; ++i) mean[i] = prior_q ? prior_q[i] : 0.0;
    gemv_tr(1.0, weights, n, n, m, y, 1.0, mean); /* :306 */
    for (int64_t i = 0; i < m; ++i) {            /* :313-319 */
        const double base_cov = fro_kernel(p, Xq + i, ldq, Xq + i, ldq, d);
        const double predicted_cov = dot8(cov_train_inputs + i * n, weights + i * n, n);
        var[i] = base_cov - predicted_cov;
    }
    free(cov_train_inputs);
    free(weights);
}

/* mod.rs:329-350 */
int fro_predict_covariance(const fro_kprog* p, const double* L, int64_t n, int64_t ldl, const double* X, int64_t ldx,
                           int64_t d, const double* Xq, int64_t m, int64_t ldq, double* cov, int64_t ldc)
{
    double* kl = (double*)malloc(sizeof(double) * (size_t)(n * m > 0 ? n * m : 1));
    fro_make_covariance_matrix(p, X, n, ldx, Xq, m, ldq, d, kl, n);     /* :337-338 */
    fro_make_covariance_matrix(p, Xq, m, ldq, Xq, m, ldq, d, cov, ldc); /* :339 */
    if (fro_solve_lower(L, n, ldl, kl, m, n)) {                          /* :342-345 */
        free(kl);
        return -1;
    }
    for (int64_t c = 0; c < m; ++c) /* :348 gemm_tr(-1, kl, kl, 1): per column a gemv_tr */
        gemv_tr(-1.0, kl, n, n, m, kl + c * n, 1.0, cov + c * ldc);
    free(kl);
    return 0;
}

/* mod.rs:371-392 + multivariate_normal.rs:54-59 */
int fro_sample_at(const fro_kprog* p, const double* L, int64_t n, int64_t ldl, const double* X, int64_t ldx, int64_t d,
                  const double* y, const double* Xq, int64_t m, int64_t ldq, const double* prior_q, double* mean,
                  double* cov, double* cov_l)
{
    const size_t sz = sizeof(double) * (size_t)(n * m > 0 ? n * m : 1);
    double* cov_train_inputs = (double*)malloc(sz);
    double* weights = (double*)malloc(sz);
    fro_make_covariance_matrix(p, X, n, ldx, Xq, m, ldq, d, cov_train_inputs, n); /* :377-378 */
    memcpy(weights, cov_train_inputs, sz);
    fro_chol_solve(L, n, ldl, weights, m, n);                         /* :379 */
    fro_make_covariance_matrix(p, Xq, m, ldq, Xq, m, ldq, d, cov, m); /* :382 */
    for (int64_t c = 0; c < m; ++c)                                   /* :383 gemm_tr(-1, K*, W, 1) */
        gemv_tr(-1.0, cov_train_inputs, n, n, m, weights + c * n, 1.0, cov + c * m);
    for (int64_t i = 0; i < m; ++i) mean[i] = prior_q ? prior_q[i] : 0.0; /* :387 */
    gemv_tr(1.0, weights, n, n, m, y, 1.0, mean);                     /* :388 */
    free(cov_train_inputs);
    free(weights);
    /* MultivariateNormal::new: covariance.cholesky().expect(..).unpack() */
    memcpy(cov_l, cov, sizeof(double) * (size_t)(m * m));
    const int st = fro_cholesky(cov_l, m, m, 0, 0.0, NULL, NULL);
    if (st) return st;
    for (int64_t c = 0; c < m; ++c) /* unpack(): strict upper triangle zeroed */
        for (int64_t r = 0; r < c; ++r) AT(cov_l, m, r, c) = 0.0;
    return 0;
}

/* multivariate_normal.rs:68-73 : mean + L * z ([nalgebra] gemv: column-oriented axpy accumulation) */
void fro_mvn_sample(const double* mean, const double* cov_l, int64_t m, const double* z, double* out)
{
    for (int64_t i = 0; i < m; ++i) out[i] = 0.0;
    if (m > 0) {
        for (int64_t i = 0; i < m; ++i) out[i] = z[0] * cov_l[i];
        for (int64_t j = 1; j < m; ++j) axpy1(z[j], cov_l + j * m, out, m);
    }
    for (int64_t i = 0; i < m; ++i) out[i] = mean[i] + out[i];
}

/* ------------------------------------------------------------------------------------------ */
/* src/gaussian_process/optimizer.rs                                                           */

/* [nalgebra] &A * x (gemv, beta = 0): y = x0*col0, then y += x_j * col_j */
static void gemv_n(const double* A, int64_t n, int64_t lda, const double* x, double* y)
{
    if (n == 0) return;
    for (int64_t i = 0; i < n; ++i) y[i] = x[0] * A[i];
    for (int64_t j = 1; j < n; ++j) axpy1(x[j], A + j * lda, y, n);
}

/* shared body of optimizer.rs:24-52 and :159-192.  scale <= 0 means "unscaled". */
static void gradient_terms(const fro_kprog* p, const double* cov_inv, const double* alpha, int64_t n, const double* X,
                           int64_t ldx, int64_t d, int use_scale, double scale, double* out_grad)
{
    const int np = fro_kprog_nb_parameters(p);
    double* G = (double*)malloc(sizeof(double) * (size_t)((int64_t)np * n * n > 0 ? (int64_t)np * n * n : 1));
    fro_make_gradient_covariance_matrices(p, X, n, ldx, d, G);
    for (int q = 0; q < np; ++q) {
        const double* cg = G + (int64_t)q * n * n;
        double data_fit = 0.0; /* :40-43 / :181-186 */
        for (int64_t c = 0; c < n; ++c) data_fit += dot8(alpha, cg + c * n, n) * alpha[c];
        if (use_scale) data_fit = data_fit / scale;
        double complexity_penalty = 0.0; /* :46-47 / :189-190 : row_i(cov_inv).tr_dot(col_i(G)) */
        for (int64_t i = 0; i < n; ++i) {
            double res = 0.0;
            for (int64_t k = 0; k < n; ++k) res += AT(cov_inv, n, i, k) * cg[k + i * n];
            complexity_penalty += res;
        }
        out_grad[q] = (data_fit - complexity_penalty) / 2.0;
    }
    free(G);
}

/* optimizer.rs:24-60 */
void fro_gradient_marginal_likelihood(const fro_kprog* p, const double* L, int64_t n, int64_t ldl, const double* X,
                                      int64_t ldx, int64_t d, const double* y, double noise, double* out_grad)
{
    const int np = fro_kprog_nb_parameters(p);
    double* cov_inv = (double*)malloc(sizeof(double) * (size_t)(n * n > 0 ? n * n : 1));
    double* alpha = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    fro_chol_inverse(L, n, ldl, cov_inv, n); /* :32 */
    gemv_n(cov_inv, n, n, y, alpha);         /* :33 */
    gradient_terms(p, cov_inv, alpha, n, X, ldx, d, 0, 0.0, out_grad);
    const double data_fit = dot8(alpha, alpha, n); /* :54 */
    double trace = 0.0;                            /* :55 */
    for (int64_t i = 0; i < n; ++i) trace += AT(cov_inv, n, i, i);
    out_grad[np] = noise * (data_fit - trace); /* :56 */
    free(cov_inv);
    free(alpha);
}

/* optimizer.rs:159-203 */
void fro_scaled_gradient_marginal_likelihood(const fro_kprog* p, const double* L, int64_t n, int64_t ldl,
                                             const double* X, int64_t ldx, int64_t d, const double* y,
                                             double* out_scale, double* out_grad)
{
    double* cov_inv = (double*)malloc(sizeof(double) * (size_t)(n * n > 0 ? n * n : 1));
    double* alpha = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    fro_chol_inverse(L, n, ldl, cov_inv, n);            /* :169 */
    gemv_n(cov_inv, n, n, y, alpha);                    /* :171 */
    const double scale = dot8(y, alpha, n) / (double)n; /* :174 */
    gradient_terms(p, cov_inv, alpha, n, X, ldx, d, 1, scale, out_grad);
    *out_scale = scale;
    free(cov_inv);
    free(alpha);
}

#define FRO_MAX_PARAMS (3 * FRO_KPROG_MAX_OPS + 1)

/* optimizer.rs:69-149 */
int fro_optimize_parameters(fro_kprog* p, double* noise, double* L, int64_t n, int64_t ldl, const double* X,
                            int64_t ldx, int64_t d, const double* y, int has_eps, double eps, int max_iter,
                            double convergence_fraction)
{
    const double beta1 = 0.9, beta2 = 0.999, epsilon = 1e-8, learning_rate = 0.1; /* :79-82 */
    double parameters[FRO_MAX_PARAMS], mean_grad[FRO_MAX_PARAMS], var_grad[FRO_MAX_PARAMS], gradients[FRO_MAX_PARAMS];
    int np = fro_kprog_get_parameters(p, parameters);
    for (int q = 0; q < np; ++q)
        if (parameters[q] == 0.0) parameters[q] = epsilon; /* :84-97 */
    parameters[np] = log(*noise);                          /* :98 */
    const int nparam = np + 1;
    for (int q = 0; q < nparam; ++q) mean_grad[q] = var_grad[q] = 0.0;
    int it = 0;
    for (int i = 1; i <= max_iter; ++i) {
        it = i;
        /* gradient_marginal_likelihood returns nb_parameters()+1 values; the ADAM loop zips by index */
        double gfull[FRO_MAX_PARAMS + 1];
        fro_gradient_marginal_likelihood(p, L, n, ldl, X, ldx, d, y, *noise, gfull);
        const int ngrad = fro_kprog_nb_parameters(p) + 1;
        gfull[ngrad - 1] *= *noise; /* :106-110 */
        for (int q = 0; q < nparam; ++q) gradients[q] = gfull[q];
        int had_significant_progress = 0;
        for (int q = 0; q < nparam; ++q) { /* :113-122 */
            mean_grad[q] = beta1 * mean_grad[q] + (1.0 - beta1) * gradients[q];
            var_grad[q] = beta2 * var_grad[q] + (1.0 - beta2) * powi_(gradients[q], 2);
            const double bias_corrected_mean = mean_grad[q] / (1.0 - powi_(beta1, i));
            const double bias_corrected_variance = var_grad[q] / (1.0 - powi_(beta2, i));
            const double delta = learning_rate * bias_corrected_mean / (sqrt(bias_corrected_variance) + epsilon);
            had_significant_progress |= fabs(delta) > convergence_fraction;
            parameters[q] *= 1.0 + delta;
        }
        if (fro_kprog_set_parameters(p, parameters, nparam)) return -1000000; /* :125 (reads [..]) */
        *noise = exp(parameters[np]);                                         /* :126-130 */
        const int st = fro_make_cholesky_cov_matrix(p, X, n, ldx, d, *noise, has_eps, eps, L, ldl, NULL, NULL);
        if (st) return -st;
        if (!had_significant_progress) break; /* :138 (time budget not modelled) */
    }
    return it;
}

/* optimizer.rs:211-283 */
int fro_scaled_optimize_parameters(fro_kprog* p, double* noise, double* L, int64_t n, int64_t ldl, const double* X,
                                   int64_t ldx, int64_t d, const double* y, int has_eps, double eps, int max_iter,
                                   double convergence_fraction)
{
    const double beta1 = 0.9, beta2 = 0.999, epsilon = 1e-8, learning_rate = 0.1; /* :221-224 */
    double parameters[FRO_MAX_PARAMS], mean_grad[FRO_MAX_PARAMS], var_grad[FRO_MAX_PARAMS], gradients[FRO_MAX_PARAMS];
    int np = fro_kprog_get_parameters(p, parameters);
    for (int q = 0; q < np; ++q)
        if (parameters[q] == 0.0) parameters[q] = epsilon; /* :226-239 */
    for (int q = 0; q < np; ++q) mean_grad[q] = var_grad[q] = 0.0;
    int it = 0;
    for (int i = 1; i <= max_iter; ++i) {
        it = i;
        double scale;
        fro_scaled_gradient_marginal_likelihood(p, L, n, ldl, X, ldx, d, y, &scale, gradients); /* :246 */
        int had_significant_progress = 0;
        for (int q = 0; q < np; ++q) { /* :249-258 */
            mean_grad[q] = beta1 * mean_grad[q] + (1.0 - beta1) * gradients[q];
            var_grad[q] = beta2 * var_grad[q] + (1.0 - beta2) * powi_(gradients[q], 2);
            const double bias_corrected_mean = mean_grad[q] / (1.0 - powi_(beta1, i));
            const double bias_corrected_variance = var_grad[q] / (1.0 - powi_(beta2, i));
            const double delta = learning_rate * bias_corrected_mean / (sqrt(bias_corrected_variance) + epsilon);
            had_significant_progress |= fabs(delta) > convergence_fraction;
            parameters[q] *= 1.0 + delta;
        }
        if (fro_kprog_set_parameters(p, parameters, np)) return -1000000; /* :261 */
        if (fro_kprog_rescale(p, scale)) return -1000001;                 /* :262 */
        *noise *= scale;                                                  /* :263 (noise, not noise^2) */
        np = fro_kprog_get_parameters(p, parameters);                     /* :264 */
        const int st = fro_make_cholesky_cov_matrix(p, X, n, ldx, d, *noise, has_eps, eps, L, ldl, NULL, NULL);
        if (st) return -st;
        if (!had_significant_progress) break; /* :272 */
    }
    return it;
}

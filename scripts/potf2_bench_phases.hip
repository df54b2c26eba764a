#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
namespace fr {
constexpr int PB = 128;
constexpr int PT = 1024;  // threads
constexpr int PE = 16;    // elements per thread

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for the
// column store of every step to be acknowledged by memory (~2 us per step, 6x the rest of the step).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// pivot rule for one diagonal value (executed by ONE thread per step); returns the pivot, logs substitutions/failures
__device__ __forceinline__ double pivot_of(double d, int mode, double sub, int64_t col, int64_t* __restrict__ info)
{
    if (mode == 3) return d;        // already a factor
    if (mode == 2) return sqrt(d);  // insert_column: plain sqrt
    if (d > 0.0) return sqrt(d);
    if (mode == 1 && sub > 0.0) {
        const int64_t q = info[1];
        info[3 + q] = col;
        info[1] = q + 1;
        return sqrt(sub);
    }
    if (info[0] == 0) info[0] = 1 + col;
    return __builtin_nan("");
}

__global__ __launch_bounds__(PT) void potf2_kernel(double* __restrict__ A, int64_t lda, int n, int64_t col0, int mode,
                                                   double sub, double* __restrict__ inv, int64_t ldinv,
                                                   int64_t* __restrict__ info)
{
    // Lc[x]      : L(x, j), the scaled column j (x > j)                          -- the row multiplier of the step
    // Vc[c]      : L(c, j) for c > j (0 in mode 3: no factor update), Vc[PB + c] : X(j, c) for c < j (0 for c >= j)
    // so that the update of element (i, c) is  a -= Lc[i] * Vc[c > j ? c : PB + c]  with no per-element predicate.
    __shared__ double Lc[PB];
    __shared__ double Vc[2 * PB];
    __shared__ __attribute__((aligned(16))) double piv[2];  // {pivot, 1/pivot} of the current step
    const int t = threadIdx.x;
    const int i = t & (PB - 1);
    const int cg = t >> 7;  // 8 column groups of 128 threads (two waves): cg is wave-uniform
    const bool row_ok = i < n;
    const int wave_row0 = i & 64;  // first row held by this wave

    double a[PE];  // working element (i, cg + 8k): A, then (once column c is done) the inverse X
    double l[PE];  // finished factor entries L(i, cg + 8k), stored after the loop (no global traffic inside it)
#pragma unroll
    for (int k = 0; k < PE; ++k) {
        const int c = cg + 8 * k;
        a[k] = (row_ok && c < n && i >= c) ? A[i + (int64_t)c * lda] : 0.0;
        l[k] = 0.0;
    }
    if (t == 0) {
        const double p0 = pivot_of(a[0], mode, sub, col0, info);
        piv[0] = p0;
        piv[1] = 1.0 / p0;
    }
    lds_barrier();

    long long s1=0,s2=0,s3=0,s4=0;
    for (int j = 0; j < n; ++j) {
        const long long q0 = __builtin_amdgcn_s_memtime();
        const int jcg = j & 7, jk = j >> 3;
        // ---- phase 1: owners of column j / row j scale and publish (sqrt and reciprocal were computed by ONE thread
        //      at the end of the previous step)
        const double p = piv[0], ip = piv[1];
        if (cg == jcg && i >= j && row_ok) {  // the threads holding column j
#pragma unroll
            for (int k = 0; k < PE; ++k) {
                if (k == jk) {  // uniform: exactly one of the 16 statically indexed bodies runs
                    const double v = a[k];
                    const double q = (mode == 3) ? v : v / p;  // true division, as `col /= denom`
                    const bool diag = (i == j);
                    const double lv = diag ? p : q;  // L(i, j)
                    Lc[i] = lv;
                    Vc[i] = (mode == 3) ? 0.0 : lv;
                    l[k] = lv;
                    a[k] = diag ? ip : -q * ip;  // X(i, j): 1/p on the diagonal, else 0 - L(i,j) X(j,j)
                }
            }
        }
        if (i == j) {  // the 8 threads holding row j: scale and publish X(j, c), c < j
#pragma unroll
            for (int k = 0; k < PE; ++k) {
                const int c = cg + 8 * k;
                const double sc = a[k] * ip;
                const bool lt = c < j;
                a[k] = lt ? sc : a[k];
                Vc[PB + c] = lt ? sc : 0.0;  // zero for c >= j: the update of column j itself must be a no-op
            }
        }
        const long long q1 = __builtin_amdgcn_s_memtime();
        lds_barrier();
        const long long q2 = __builtin_amdgcn_s_memtime();
        // ---- phase 2: a(i, c) -= L(i, j) * (c > j ? L(c, j) : X(j, c)).  One LDS read + one FMA per element; waves whose
        //      rows are all finished skip it (the block is VALU-throughput bound: 1024 threads x 16 elements per step)
        if (wave_row0 + 63 > j) {
            const bool act = (i > j) && row_ok;
            const double lraw = Lc[i];
            const double lij = act ? lraw : 0.0;
            if (act && i == j + 1 && cg == ((j + 1) & 7)) {
                // owner of the next diagonal element: take the next pivot now; the other waves overlap it with their
                // 16 updates
                const int nk = (j + 1) >> 3;
                double nd = 0.0;
#pragma unroll
                for (int k = 0; k < PE; ++k)
                    if (k == nk) nd = a[k];
                if (mode != 3) nd = nd - lij * lij;
                const double pn = pivot_of(nd, mode, sub, col0 + j + 1, info);
                piv[0] = pn;
                piv[1] = 1.0 / pn;
            }
            double vc[PE];
#pragma unroll
            for (int k = 0; k < PE; ++k) {
                const int c = cg + 8 * k;
                vc[k] = Vc[c + ((c > j) ? 0 : PB)];
            }
#pragma unroll
            for (int k = 0; k < PE; ++k) a[k] = __builtin_fma(-lij, vc[k], a[k]);
        }
        const long long q3 = __builtin_amdgcn_s_memtime();
        lds_barrier();
        const long long q4 = __builtin_amdgcn_s_memtime();
        s1+=q1-q0; s2+=q2-q1; s3+=q3-q2; s4+=q4-q3;
    }

    if ((t & 63) == 0 && info) { int w = t >> 6; info[8+4*w]=s1; info[9+4*w]=s2; info[10+4*w]=s3; info[11+4*w]=s4; }
#pragma unroll
    for (int k = 0; k < PE; ++k) {
        const int c = cg + 8 * k;
        if (row_ok && c < n && i >= c) {
            if (mode != 3) A[i + (int64_t)c * lda] = l[k];
            if (inv) inv[i + (int64_t)c * ldinv] = a[k];
        } else if (row_ok && c < n && inv) {
            inv[i + (int64_t)c * ldinv] = 0.0;
        }
    }
}

}
int main(){
  const int n=128; std::vector<double> h(n*n);
  for(int c=0;c<n;++c) for(int r=0;r<n;++r) h[r+c*n]= (r==c? n+1.0 : 1.0/(1.0+abs(r-c)));
  double *A,*inv; int64_t* info; (void)hipMalloc(&A,n*n*8); (void)hipMalloc(&inv,n*n*8); (void)hipMalloc(&info,8*(3+n));
  for(int rep=0;rep<2;++rep){
    (void)hipMemcpy(A,h.data(),n*n*8,hipMemcpyHostToDevice); (void)hipMemset(info,0,8*(3+n));
    hipLaunchKernelGGL(fr::potf2_kernel,dim3(1),dim3(1024),0,0,A,(int64_t)n,n,(int64_t)0,0,0.0,inv,(int64_t)n,info); (void)hipDeviceSynchronize();
    int64_t hi[80]; (void)hipMemcpy(hi,info,8*80,hipMemcpyDeviceToHost);
    if (rep) for (int w=0; w<16; w+=3) printf("wave %2d: per-step ticks P1 %5.0f  bar1 %5.0f  P2 %5.0f  bar2 %5.0f\n",w,hi[8+4*w]/128.0,hi[9+4*w]/128.0,hi[10+4*w]/128.0,hi[11+4*w]/128.0);
  }
  return 0; }
